"""Drop-in complex DCUnet on libsehip (reference: src/model/dcunet.py:53-162; BASELINE config C2).

Same constructor arguments, same ``forward(x[B, 1, F, T, 2]) -> [B, 1, F, T, 2]`` on the STFT-domain tensors the Solver
produces with stft_custom (src/solver.py:454-458), same state_dict keys -- including the reference's double registration of
every block (``encoder{i}.*`` through add_module AND ``encoders.{i}.*`` through the ModuleList, src/model/dcunet.py:74-76,
99-100), so its checkpoints load here and vice versa.  Parameters are views into one flat fp32 buffer (one RCCL all-reduce,
one fused clip + Adam launch); forward / backward run the HIP kernels through the C ABI; a CPU tensor raises SehipError.
Built: the complex network (data_type=True), model_depth 10 and 20, zero padding, masking modes E / C / R, up to 128 stored complex
channels (model_complexity <= 90: 63 / 126 channels, the paper's "Large" network).
"""
import math
import os

import torch
from torch import nn

from .. import plan_dcunet as P
from .._lib import SehipError
from .flat import FlatModule

_STATIC_CACHE = {}


class _DCUNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, spec, anchor):
        ctx.model = model
        ctx.ws = model._run_forward(spec)
        ctx.generation = ctx.ws.generation
        return ctx.ws.out.clone()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.generation != ctx.ws.generation or ctx.ws.closed:
            raise SehipError("DCUnet.backward: the workspace of this forward was overwritten by a later forward of the same shape "
                             "(or evicted); run backward before the next forward of that shape")
        # (autograd runs this in its device thread: without a scope of its own every library call of the backward pass looks
        #  torch's current stream up again -- ~60 look-ups of ~7 us per step: round 5, tools/host_profile2.py)
        from .._lib import stream_scope
        with stream_scope():
            ctx.model._run_backward(ctx.ws, grad_out)
        return None, None, None


class DCUnet(FlatModule):
    def __init__(self, audio_channels=1, data_type=False, model_complexity=45, model_depth=20, padding_mode="zeros",
                 masking_mode="E", *args, **kwargs):
        super().__init__()
        self.cfg = cfg = P.DCUNetConfig(audio_channels=audio_channels, data_type=data_type, model_complexity=model_complexity,
                                        model_depth=model_depth, padding_mode=padding_mode, masking_mode=masking_mode)
        if cfg.key() not in _STATIC_CACHE:
            _STATIC_CACHE[cfg.key()] = P.DCUNetStatic(cfg)
        self.static = _STATIC_CACHE[cfg.key()]
        self.data_type, self.padding_mode, self.masking_mode = data_type, padding_mode, masking_mode
        self.model_length = model_depth // 2
        self._plans, self._tables_by_plan = {}, {}
        self._ws_cap = max(1, int(os.environ.get("SEHIP_WS_CACHE", "4")))
        self._build_flat()
        # the reference registers every block twice: add_module("encoder{i}") and the ModuleLists assigned at the end
        self.decoders = nn.ModuleList([getattr(self, f"decoder{i}") for i in range(self.model_length)])
        self.encoders = nn.ModuleList([getattr(self, f"encoder{i}") for i in range(self.model_length)])
        self.reset_parameters()

    def reset_parameters(self):
        """nn.Conv2d / nn.ConvTranspose2d defaults (kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)) for weight and bias; the
        transposed convolution's fan_in is weight.size(1) * kernel area), BatchNorm weight 1 / bias 0 / stats (0, 1)."""
        with torch.no_grad():
            for name, p in self._params:
                if ".bn." in name:
                    p.fill_(1.0 if name.endswith("weight") else 0.0)
                    continue
                wname = name[:-len("bias")] + "weight" if name.endswith("bias") else name
                w = dict(self._params)[wname]
                fan_in = w.shape[1] * w.shape[2] * w.shape[3]
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            for name, node, leaf in self._buffers_named:
                getattr(node, leaf).fill_(1.0 if leaf == "running_var" else 0.0)
            self._nbt.zero_()

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._tables_by_plan = {}
        return self

    # ---- HIP path -------------------------------------------------------------------------------------
    def set_deterministic(self, on=True):
        """The reference's `solver.cudnn_deterministic` switch (src/conf/config.yaml:130, src/utils.py:108-111) for this model.  Nothing
        in the DCUnet plan changes: its BatchNorm sums already go through per-workgroup partial rows, its products take the library's
        fixed-order weight gradients, and the one kernel of its own that used fp32 atomics (the mask layer's weight gradient,
        sehip_dcunet_mask_bwd) follows the process-wide sehip_set_deterministic, which the Solver / sehip.utils.prepare_device switch
        on.  Two runs of the same steps are then bit-identical (tests/test_gpu_deterministic.py)."""
        self._deterministic = bool(on)
        return self

    def workspace(self, batch, n_bins, n_frames):
        dev = self._require_gpu("DCUnet")
        gk = (n_bins, n_frames)
        if gk not in self._plans:
            self._plans[gk] = P.DCUNetPlan(self.static, n_bins, n_frames)
        if gk not in self._tables_by_plan:
            self._tables_by_plan[gk] = P.DCUNetDeviceTables(self._plans[gk], dev)
        return self._lru_get((batch, n_bins, n_frames), self._ws_cap,
                             lambda: P.DCUNetWorkspace(self._plans[gk], self._tables_by_plan[gk], batch, dev))

    def _run_forward(self, spec):
        b, c, f, t, two = spec.shape
        ws = self.workspace(b * c, f, t)
        ws.generation += 1
        ws.forward(spec.contiguous().float(), self._flat, self._bflat, self._nbt, training=self.training)
        return ws

    def _run_backward(self, ws, grad_out):
        if not self.training:
            raise SehipError("DCUnet.backward in eval mode (running-statistics BatchNorm) is not built")
        g = grad_out.contiguous().float()
        tail = self._tail_for_backward()
        self._backward_into_flat(lambda dst: ws.backward(g, self._flat, dst, tail=tail))
        self._tail_mark(tail)

    def forward(self, x):
        if x.dim() != 5 or x.shape[-1] != 2 or x.shape[1] != 1:
            raise SehipError(f"DCUnet.forward: [B, 1, F, T, 2] expected (stft_custom's layout), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise SehipError("DCUnet.forward got a CPU tensor: the HIP path needs a gfx950 GPU (no CPU fallback)")
        if torch.is_grad_enabled() and self.training:
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            return _DCUNetFunction.apply(self, x, self._anchor)
        return self._run_forward(x).out.clone()
