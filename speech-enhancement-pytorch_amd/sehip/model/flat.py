"""Base class of the HIP models: a module tree with the reference's parameter / buffer NAMES whose tensors are views into
ONE flat fp32 parameter buffer, one flat buffer of running statistics and one int64 vector of BatchNorm counters.

Why flat: the data-parallel exchange is a single RCCL all-reduce of `flat_grads`, clip + Adam is one fused launch over
(`flat_params`, `flat_grads`, m, v) (sehip/optim.py), and the table-driven weight packing of the GEMM engine indexes
`flat_params` directly.  A subclass provides `self.static.layout` (sehip.plan.ParamLayout) before calling `_build_flat()`.
"""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .._lib import SehipError


class _Node(nn.Module):
    """Plain container; the tree of these reproduces the reference's module/parameter names."""

    def __getitem__(self, idx):  # encoder[i][0] style access like the reference's nn.Sequential
        return getattr(self, str(idx))


def _clone_state_hook(module, state_dict, prefix, local_metadata):
    # parameters are views of one flat buffer; give every checkpoint entry its own storage like the reference's
    for k in list(state_dict.keys()):
        state_dict[k] = state_dict[k].detach().clone()
    return state_dict


class FlatModule(nn.Module):
    def _build_flat(self, list_roots=()):
        """Creates the flat storage and registers every tensor of `self.static.layout.specs` under its dotted name.
        `list_roots`: top-level names that are nn.ModuleList in the reference (integer children)."""
        L = self.static.layout
        self._flat = torch.zeros(L.n_params)
        self._gflat = None
        self._bflat = torch.zeros(max(L.n_buffers, 1))
        self._nbt = torch.zeros(len(L.nbt_names), dtype=torch.int64)
        self._ws = OrderedDict()
        self.storage_epoch = 0        # bumped whenever the flat buffers are re-created (captured hipGraphs go stale)
        self._anchor = None
        self._grads_live = False
        # single-replica tail of the train step (sehip_unpack_grad_sums / sehip_opt_step_m): FlatOptimizer.zero_grad() leaves the
        # addresses of its accumulators here, a plan that supports it un-packs the gradients with the sums and sets _tail_done
        self._tail_sink, self._tail_done = None, False
        self._tail_counted = self._tail_dirty = False      # the un-pack already advanced the step counter / left sums in the current set
        for name in list_roots:
            setattr(self, name, nn.ModuleList())
        self._params, self._buffers_named, self._nbt_named = [], [], []
        for name, shape, kind in L.specs:
            parts = name.split(".")
            node = self._descend(parts[:-1])
            if kind == "param":
                off, _ = L.param_off[name]
                p = nn.Parameter(self._flat[off:off + int(np.prod(shape))].view(shape))
                node.register_parameter(parts[-1], p)
                self._params.append((name, p))
            elif kind == "buffer":
                off, _ = L.buffer_off[name]
                node.register_buffer(parts[-1], self._bflat[off:off + int(np.prod(shape))].view(shape))
                self._buffers_named.append((name, node, parts[-1]))
            else:
                node.register_buffer(parts[-1], self._nbt[L.nbt_idx[name]])
                self._nbt_named.append((name, node, parts[-1]))
        self._register_state_dict_hook(_clone_state_hook)

    def _tail_for_backward(self):
        """FlatOptimizer's accumulators for this backward pass (sehip_unpack_grad_sums), or None: not when the pass accumulates into
        live gradients or a data-parallel hook takes finished ranges (the sums must then follow the final buffer)."""
        accumulating = self._grads_live and self._params[0][1].grad is not None
        if accumulating or getattr(self, "grad_range_hook", None) is not None:
            return None
        return self._tail_sink

    def _tail_mark(self, tail):
        if tail is not None:
            self._tail_done = self._tail_counted = self._tail_dirty = True
        else:
            self._tail_done = False

    def _descend(self, parts):
        node = self
        for key in parts:
            if isinstance(node, nn.ModuleList):
                while len(node) <= int(key):
                    node.append(_Node())
                node = node[int(key)]
            else:
                if not hasattr(node, key):
                    node.add_module(key, _Node())
                node = getattr(node, key)
        return node

    # ---- flat storage follows the module across devices ---------------------------------------------
    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        L = self.static.layout
        dev = self._params[0][1].device
        flat = torch.zeros(L.n_params, device=dev)
        for name, p in self._params:
            off, shape = L.param_off[name]
            v = flat[off:off + p.numel()].view(shape)
            v.copy_(p.data)
            p.data = v
            p.grad = None
        bflat = torch.zeros(max(L.n_buffers, 1), device=dev)
        for name, node, leaf in self._buffers_named:
            off, shape = L.buffer_off[name]
            v = bflat[off:off + int(np.prod(shape))].view(shape)
            v.copy_(getattr(node, leaf))
            node._buffers[leaf] = v
        nbt = torch.zeros(len(L.nbt_names), dtype=torch.int64, device=dev)
        for name, node, leaf in self._nbt_named:
            i = L.nbt_idx[name]
            nbt[i] = getattr(node, leaf).to(torch.int64)
            node._buffers[leaf] = nbt[i]
        self._flat, self._bflat, self._nbt = flat, bflat, nbt
        for ws in self._ws.values():
            ws.close()
        self._gflat, self._tables, self._ws, self._anchor, self._grads_live = None, None, OrderedDict(), None, False
        self.storage_epoch += 1
        return self

    @property
    def flat_params(self):
        return self._flat

    @property
    def flat_grads(self):
        if self._gflat is None or self._gflat.device != self._flat.device:
            self._gflat = torch.zeros_like(self._flat)
        return self._gflat

    def bind_grads(self):
        """Make every p.grad a view into flat_grads (what the fused optimizer and the all-reduce operate on)."""
        L = self.static.layout
        g = self.flat_grads
        for name, p in self._params:
            off, shape = L.param_off[name]
            p.grad = g[off:off + p.numel()].view(shape)

    def _require_gpu(self, what):
        dev = self._flat.device
        if dev.type != "cuda":
            raise SehipError(f"{what} parameters are on {dev}: the HIP path needs a gfx950 GPU (no CPU fallback); "
                             "call model.to('cuda') first")
        return dev

    def _lru_get(self, key, cap, make):
        """Workspace cache: least recently used shapes are evicted first (never one a captured hipGraph points into)."""
        ws = self._ws.get(key)
        if ws is None:
            while len(self._ws) >= cap:
                victim = next((k for k, w in self._ws.items() if not w.pinned), None)
                if victim is None:
                    break
                self._ws.pop(victim).close()
            ws = self._ws[key] = make()
        else:
            self._ws.move_to_end(key)
        return ws

    def _note_graph_epoch(self, ws):
        """A workspace bumps `graph_epoch` when launches captured from it have gone stale although the flat buffers are unchanged (its
        persistent LSTM kernels fell back after a hand-off time-out): captured hipGraphs are keyed on storage_epoch, so move that."""
        ge = getattr(ws, "graph_epoch", 0)
        if ws is not None and ge != getattr(ws, "_graph_epoch_seen", 0):
            ws._graph_epoch_seen = ge
            self.storage_epoch += 1

    def _backward_into_flat(self, run):
        """`run(dst)` writes the flat parameter gradients of one backward pass into dst; accumulates like autograd when the
        previous gradients are still live (no zero_grad in between)."""
        accumulate = self._grads_live and self._params[0][1].grad is not None
        if accumulate:
            tmp = torch.empty_like(self.flat_grads)
            run(tmp)
            self.flat_grads.add_(tmp)
        else:
            run(self.flat_grads)
        self.bind_grads()
        self._grads_live = True
