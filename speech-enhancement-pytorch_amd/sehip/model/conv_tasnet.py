"""Drop-in ConvTasNet on libsehip (reference: src/model/conv_tasnet.py:34-154; BASELINE config C4).

Same constructor arguments, same ``forward(mixture[M, ac, T]) -> [M, C, ac, T]`` (C = len(sources)), same state_dict keys
(``encoder.conv1d_U.weight``, ``separator.network.{0,1,2.r.x.net...,3}``, ``decoder.basis_signals.weight``), so the
reference's checkpoints load here and vice versa.  Parameters are views into one flat fp32 buffer; forward / backward run the
HIP kernels through the C ABI; a CPU tensor raises SehipError.  Built: the shipped options (skip=False, gLN, non-causal,
relu or softmax mask), kernel size P = 3 / 5 / 7, channel counts that are multiples of 8 (N up to 512: the paper's N = 512, L = 16 encoder
included; B, H up to 512).
"""
import math
import os

import torch

from .. import plan_tasnet as P
from .._lib import SehipError
from .flat import FlatModule

_STATIC_CACHE = {}


class _TasNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, wav, anchor):
        ctx.model = model
        ctx.ws = model._run_forward(wav)
        ctx.generation = ctx.ws.generation
        return ctx.ws.out.clone()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.generation != ctx.ws.generation or ctx.ws.closed:
            raise SehipError("ConvTasNet.backward: the workspace of this forward was overwritten by a later forward of the same "
                             "shape (or evicted); run backward before the next forward of that shape")
        # (autograd runs this in its device thread: without a scope of its own every library call of the backward pass looks
        #  torch's current stream up again -- ~60 look-ups of ~7 us per step: round 5, tools/host_profile2.py)
        from .._lib import stream_scope
        with stream_scope():
            ctx.model._run_backward(ctx.ws, grad_out)
        return None, None, None


class ConvTasNet(FlatModule):
    def __init__(self, sources, N=128, L=40, B=128, H=256, P=3, X=7, R=2, audio_channels=2, norm_type="gLN", causal=False,
                 mask_nonlinear="relu", sample_rate=44100, segment_length=44100 * 2 * 4, skip=False, *args, **kwargs):
        super().__init__()
        from .. import plan_tasnet
        self.cfg = cfg = plan_tasnet.TasNetConfig(sources, N=N, L=L, B=B, H=H, P=P, X=X, R=R, audio_channels=audio_channels,
                                                  norm_type=norm_type, causal=causal, mask_nonlinear=mask_nonlinear, skip=skip)
        skey = (cfg.key(), bool(os.environ.get("SEHIP_CTN_KEEP_GRADS")))      # (the test switch changes the buffer names of the plan)
        if skey not in _STATIC_CACHE:
            _STATIC_CACHE[skey] = plan_tasnet.TasNetStatic(cfg)
        self.static = _STATIC_CACHE[skey]
        self.sources, self.C = sources, cfg.C
        self.N, self.L, self.B, self.H, self.P, self.X, self.R = N, L, B, H, P, X, R
        self.audio_channels, self.sample_rate, self.segment_length = audio_channels, sample_rate, segment_length
        self._tables = None
        self._ws_cap = max(1, int(os.environ.get("SEHIP_WS_CACHE", "4")))
        self._build_flat()
        self.reset_parameters()

    def reset_parameters(self):
        """The reference's init: nn.Conv1d / nn.Linear defaults, PReLU 0.25, gamma 1 / beta 0 -- then xavier_normal_ on EVERY
        parameter with more than one dimension (src/model/conv_tasnet.py:132-134), which includes the [1, C, 1] gamma / beta
        tensors of the LayerNorms (fan_in = C, fan_out = 1)."""
        with torch.no_grad():
            for name, p in self._params:
                if p.dim() > 1:
                    rf = 1
                    for s in p.shape[2:]:
                        rf *= s
                    fan_in, fan_out = p.shape[1] * rf, p.shape[0] * rf
                    p.normal_(0.0, math.sqrt(2.0 / (fan_in + fan_out)))
                else:
                    p.fill_(0.25)           # nn.PReLU()

    def valid_length(self, length):
        return length

    def set_deterministic(self, on=True):
        """The reference's `solver.cudnn_deterministic` switch (src/conf/config.yaml:130, src/utils.py:108-111) for this model.  The work
        is done by the process-wide sehip_set_deterministic (the Solver / sehip.utils.prepare_device switch it on): the gLN statistics
        leave the 1x1 products' launches for sehip_ctn_gln_stats, every per-utterance sum of csrc/tasnet.hip goes through per-workgroup
        slots added in a fixed order by a second small launch (csrc/det.h), the column sums add their rows in row order, the weight
        gradients take the library's fixed-order kernels, the optimizer the unfused tail, and the whole step runs on ONE queue
        (plan_tasnet.TasNetWorkspace.forward says why).  Two runs of the same steps are then bit-identical
        (tests/test_gpu_deterministic.py)."""
        self._deterministic = bool(on)
        return self

    def workspace(self, batch, nsample):
        dev = self._require_gpu("ConvTasNet")
        if self._tables is None:
            self._tables = P.TasNetDeviceTables(self.static, dev)
        return self._lru_get((batch, nsample), self._ws_cap, lambda: P.TasNetWorkspace(self.static, self._tables, batch, nsample, dev))

    def _run_forward(self, wav):
        ws = self.workspace(wav.shape[0], wav.shape[-1])
        ws.generation += 1
        ws.forward(wav.contiguous().float(), self._flat)
        return ws

    def _run_backward(self, ws, grad_out):
        g = grad_out.contiguous().float()
        tail = self._tail_for_backward()
        self._backward_into_flat(lambda dst: ws.backward(g, self._flat, dst, tail=tail))
        self._tail_mark(tail)

    def forward(self, mixture):
        if mixture.dim() != 3 or mixture.shape[1] != self.audio_channels:
            raise SehipError(f"ConvTasNet.forward: [M, {self.audio_channels}, T] expected, got {tuple(mixture.shape)}")
        if not mixture.is_cuda:
            raise SehipError("ConvTasNet.forward got a CPU tensor: the HIP path needs a gfx950 GPU (no CPU fallback)")
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != mixture.device:
                self._anchor = torch.zeros(1, device=mixture.device, requires_grad=True)
            return _TasNetFunction.apply(self, mixture, self._anchor)
        return self._run_forward(mixture).out.clone()
