"""Drop-in DCCRN module on libsehip (reference: src/model/dccrn.py:10-246).

Same constructor arguments, same `forward(x[B,1,N]) -> [B,1,length]`, same state_dict keys/shapes (204 entries incl.
the persistent stft/istft buffers at the defaults; rnn_layers 1 .. 8, rnn_units 64 ... 256, any win_type, use_cbn / use_clstm either way), so checkpoints of the reference load here and vice versa.  Differences by design:
  * all parameters are views into ONE flat fp32 buffer (`flat_params`), gradients into one flat buffer
    (`flat_grads`) -- one RCCL all-reduce, one fused clip+Adam launch;
  * forward/backward run the hand-written HIP kernels through the C ABI; there is no PyTorch/CPU fallback:
    calling forward on a CPU tensor raises SehipError.
"""
import os
import numpy as np
import torch
from torch import nn

from .. import plan, ops
from .._lib import SehipError
from .flat import FlatModule, _Node

_STATIC_CACHE = {}


def _static_for(cfg, deterministic=False):
    # (the experiment switches that DCCRNStatic reads when it is built are part of the key)
    switches = tuple(os.environ.get(k) for k in ("SEHIP_NO_FUSE_STATS", "SEHIP_NO_FUSE_STATS64", "SEHIP_NO_FUSE_STATS32", "SEHIP_DEC_SPLIT"))
    key = (tuple(cfg.kernel_num), cfg.rnn_layers, cfg.rnn_units, cfg.win_len, cfg.win_inc, cfg.fft_len, cfg.length, cfg.masking_mode, str(cfg.win_type), bool(cfg.use_cbn), bool(cfg.use_clstm), switches,
           bool(deterministic))
    if key not in _STATIC_CACHE:
        _STATIC_CACHE[key] = plan.DCCRNStatic(cfg, deterministic=deterministic)
    return _STATIC_CACHE[key]


class _DCCRNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, wav, anchor):
        ctx.model = model
        ctx.ws = model._run_forward(wav)
        ctx.generation = ctx.ws.generation
        return ctx.ws.wav.view(wav.shape[0], 1, -1).clone()

    @staticmethod
    def backward(ctx, grad_out):
        # the activations live in the workspace shared by every call of this (batch, samples) shape, not in autograd's
        # saved tensors: a second forward of the same shape before this backward has overwritten them
        if ctx.generation != ctx.ws.generation or ctx.ws.closed:
            raise SehipError("DCCRN.backward: the workspace of this forward was overwritten by a later forward of the same "
                             "shape (or evicted); run backward before the next forward of that shape")
        # (autograd runs this in its device thread: without a scope of its own every library call of the backward pass looks
        #  torch's current stream up again -- ~60 look-ups of ~7 us per step: round 5, tools/host_profile2.py)
        from .._lib import stream_scope
        with stream_scope():
            ctx.model._run_backward(ctx.ws, grad_out)
        return None, None, None


class DCCRN(FlatModule):
    def __init__(self, rnn_layers=2, rnn_units=128, win_len=400, win_inc=100, fft_len=512, length=16384, win_type="hann",
                 masking_mode="E", use_clstm=True, use_cbn=True, kernel_size=5, kernel_num=[16, 32, 64, 128, 256, 256],
                 *args, **kwargs):
        super().__init__()
        self.cfg = cfg = plan.DCCRNConfig(rnn_layers=rnn_layers, rnn_units=rnn_units, win_len=win_len, win_inc=win_inc,
                                          fft_len=fft_len, length=length, win_type=win_type, masking_mode=masking_mode,
                                          use_clstm=use_clstm, use_cbn=use_cbn, kernel_size=kernel_size,
                                          kernel_num=list(kernel_num))
        self.win_len, self.win_inc, self.fft_len, self.rnn_units = win_len, win_inc, fft_len, rnn_units
        self.masking_mode, self.kernel_num = masking_mode, cfg.kernel_num
        self.static = _static_for(cfg)
        self._tables = None
        self._ws_cap = max(1, int(os.environ.get("SEHIP_WS_CACHE", "4")))
        self.grad_range_hook = None   # data-parallel: called with (lo, hi, stream) when flat_grads[lo:hi] is final (see plan.backward)

        # persistent STFT buffers (checkpoint compatibility; the FFT kernels do not read them)
        an, sy, win = _stft_bases(win_len, fft_len, win_type)
        self.stft = _Node()
        self.stft.register_buffer("weight", torch.from_numpy(an[:, None, :]))
        self.istft = _Node()
        self.istft.register_buffer("weight", torch.from_numpy(sy[:, None, :]))
        self.istft.register_buffer("window", torch.from_numpy(win[None, :, None]))
        self.istft.register_buffer("enframe", torch.eye(win_len)[:, None, :])
        self._build_flat(list_roots=("encoder", "decoder"))   # registration order encoder, decoder, enhance = the reference's
                                                               # parameters() order (optimizer state indices interchange)
        self.reset_parameters()

    def reset_parameters(self):
        """Same distributions as the reference constructors: conv N(0,0.05)/bias 0 (src/model/dccrn.py:352-355,
        :417-420), CBN Wrr=Wii=1, Wri~U(-.9,.9), B=0, running stats (0,0,1,0,1) (:497-514), PReLU 0.25,
        LSTM/Linear U(-1/sqrt(fan),1/sqrt(fan))."""
        h = self.cfg.hid
        with torch.no_grad():
            for name, p in self._params:
                leaf = name.split(".")[-1]
                if "conv.weight" in name:
                    p.normal_(0.0, 0.05)
                elif "conv.bias" in name or leaf in ("Br", "Bi") or name.endswith(".1.bias"):
                    p.zero_()
                elif leaf in ("Wrr", "Wii") or name.endswith(".1.weight"):        # (".1.weight": nn.BatchNorm2d of use_cbn=False)
                    p.fill_(1.0)
                elif leaf == "Wri":
                    p.uniform_(-0.9, 0.9)
                elif name.endswith("2.weight"):
                    p.fill_(0.25)
                elif "lstm" in name or name.startswith("enhance."):     # (use_clstm=False: `enhance` is the nn.LSTM itself)
                    p.uniform_(-1.0 / h ** 0.5, 1.0 / h ** 0.5)
                elif "trans" in name or name.startswith("tranform."):   # (sic: the reference's attribute name, src/model/dccrn.py:106)
                    p.uniform_(-1.0 / h ** 0.5, 1.0 / h ** 0.5)
                else:
                    raise KeyError(name)
            for name, node, leaf in self._buffers_named:
                getattr(node, leaf).fill_(1.0 if leaf in ("RVrr", "RVii", "running_var") else 0.0)
            self._nbt.zero_()

    # ---- HIP path -------------------------------------------------------------------------------------
    def workspace(self, batch, nsample):
        dev = self._require_gpu("DCCRN")
        if self._tables is None:
            self._tables = plan.DeviceTables(self.static, dev)
        return self._lru_get((batch, nsample), self._ws_cap,
                             lambda: plan.DCCRNWorkspace(self.static, self._tables, batch, nsample, dev))

    def set_deterministic(self, on=True):
        """The reference's `solver.cudnn_deterministic` switch (src/conf/config.yaml:130, src/utils.py:108-111) for this model: the plan
        is rebuilt without the BatchNorm sums of the convolution epilogues (fp32 atomics) and without grouped weight-gradient launches;
        the library side (per-split partial arrays for every weight gradient, one-workgroup norms) is the process-wide
        sehip_set_deterministic, which sehip.utils.prepare_device / the Solver switch on.  Same parameters, same checkpoints; every
        workspace is rebuilt.  Two runs of the same steps are then bit-identical (tests/test_gpu_deterministic.py)."""
        on = bool(on)
        if on == self.static.deterministic:
            return self
        self.static = _static_for(self.cfg, deterministic=on)
        self._tables = None
        for ws in list(self._ws.values()):
            ws.close()
        self._ws.clear()
        self._last_ws = None
        return self

    step_guard_early = True      # the guard word is final once the fused LSTM backward launch is enqueued: before the first gradient range is handed over

    def step_guard(self):
        """Device word the fused optimizer checks (sehip_opt_begin_g / sehip_opt_step_g): the sticky hand-off time-out word of the fused
        two-layer LSTM kernels of the workspace the last forward ran in (None when that workspace runs the unfused launches)."""
        ws = getattr(self, "_last_ws", None)
        # (returned while the word EXISTS, not while the fused launches are the current path: graphs captured before a fall-back keep
        #  it as their guard, and under data parallelism every rank must issue the same MAX all-reduce whatever its own path -- ADVICE r4)
        return ws.l2_sync[0:1] if ws is not None and hasattr(ws, "l2_sync") else None

    def check_health(self):
        """Called by the Solver wherever it synchronises anyway: True if steps were lost to a hand-off time-out (the workspace has
        returned to one launch per LSTM layer; captured hipGraphs are invalidated through storage_epoch).  Collective under data
        parallelism: the decision is the OR over the ranks."""
        from ..distrib import global_flag
        ws = getattr(self, "_last_ws", None)
        if ws is None:
            return global_flag(False, self._flat.device if self._flat is not None else None)
        lost = bool(ws.check_lstm_handoffs(recover=True, global_flag=global_flag))
        self._note_graph_epoch(ws)       # graphs captured on that workspace hold the fused launches: Solver.train_step_graphed re-captures
        return lost

    def _run_forward(self, wav):
        ws = self.workspace(wav.shape[0], wav.shape[-1])
        self._last_ws = ws
        ws.generation += 1
        x = wav.reshape(wav.shape[0], wav.shape[-1]).contiguous().float()
        ws.forward(x, self._flat, self._bflat, self._nbt, training=self.training)
        return ws

    def _run_backward(self, ws, grad_out):
        if not self.training:
            raise SehipError("DCCRN.backward in eval mode (running-statistics BatchNorm) is not built")
        g = grad_out.reshape(ws.B, ws.length).contiguous().float()
        accumulating = self._grads_live and self._params[0][1].grad is not None
        hook = self.grad_range_hook if not accumulating else None   # not when accumulating
        tail = self._tail_sink if (hook is None and not accumulating) else None      # FlatOptimizer's accumulators (single replica)
        self._backward_into_flat(lambda dst: ws.backward(g, self._flat, dst, range_ready=hook, tail=tail))
        if tail is not None:
            self._tail_done = self._tail_counted = self._tail_dirty = True
        else:
            self._tail_done = False          # (an accumulating or data-parallel pass: the optimizer takes its sums from the final buffer)

    def forward(self, inputs, lens=None):
        if inputs.dim() == 2:
            inputs = inputs.unsqueeze(1)
        if not inputs.is_cuda:
            raise SehipError("DCCRN.forward got a CPU tensor: the HIP path needs a gfx950 GPU (no CPU fallback)")
        if torch.is_grad_enabled() and self.training:
            if self._anchor is None or self._anchor.device != inputs.device:
                self._anchor = torch.zeros(1, device=inputs.device, requires_grad=True)
            return _DCCRNFunction.apply(self, inputs, self._anchor)
        ws = self._run_forward(inputs)
        return ws.wav.view(inputs.shape[0], 1, -1).clone()

    def get_params(self, weight_decay=0.0):
        """Same grouping helper as the reference (src/model/dccrn.py:231-246)."""
        weights, biases = [], []
        for name, p in self.named_parameters():
            (biases if "bias" in name else weights).append(p)
        return [{"params": weights, "weight_decay": weight_decay}, {"params": biases, "weight_decay": 0.0}]


def _stft_bases(win_len, fft_len, win_type="hann"):
    """init_kernels (src/model/dccrn.py:649-666): analysis = [cos; -sin] * window, synthesis = pinv(basis).T * window."""
    n = np.arange(win_len, dtype=np.float64)[None, :]
    k = np.arange(fft_len // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * k * n / fft_len
    basis = np.concatenate([np.cos(ang), -np.sin(ang)], axis=0)
    win = ops.window_of(win_type, win_len).astype(np.float64)
    return ((basis * win).astype(np.float32), (np.linalg.pinv(basis).T * win).astype(np.float32), win.astype(np.float32))
