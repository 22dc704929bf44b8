"""Drop-in Demucs on libsehip (reference: src/model/demucs.py:272-501; BASELINE config C3).

Same constructor arguments, same ``forward(mix[B, ac, T]) -> [B, S, ac, T]`` (S = len(sources)), same state_dict keys
(``encoder.{i}.0.weight``, ``encoder.{i}.3.layers.{d}.3.lstm.weight_ih_l0_reverse``, ``decoder.{j}.3.weight`` ...), including the
key migration of load_state_dict (:492-501), so the reference's checkpoints load here and vice versa.  Parameters are views
into one flat fp32 buffer; forward / backward run the HIP kernels through the C ABI; a CPU tensor raises SehipError.
Built: the structure of the constructor defaults (sehip/plan_demucs.py lists the options and limits).
"""
import math
import os

import torch

from .. import plan_demucs as P
from .._lib import SehipError
from .flat import FlatModule

_STATIC_CACHE = {}


class _DemucsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, mix, anchor):
        ctx.model = model
        ctx.ws = model._run_forward(mix, need_backward=True)      # (grad mode is off inside Function.forward: say it explicitly)
        ctx.generation = ctx.ws.generation
        return ctx.ws.out.clone()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.generation != ctx.ws.generation or ctx.ws.closed:
            raise SehipError("Demucs.backward: the workspace of this forward was overwritten by a later forward of the same shape "
                             "(or evicted); run backward before the next forward of that shape")
        # (autograd runs this in its device thread: without a scope of its own every library call of the backward pass looks
        #  torch's current stream up again -- ~60 look-ups of ~7 us per step: round 5, tools/host_profile2.py)
        from .._lib import stream_scope
        with stream_scope():
            ctx.model._run_backward(ctx.ws, grad_out)
        return None, None, None


class Demucs(FlatModule):
    def __init__(self, sources, audio_channels=2, channels=64, growth=2., depth=6, rewrite=True, lstm_layers=0, kernel_size=8, stride=4,
                 context=1, gelu=True, glu=True, norm_starts=4, norm_groups=4, dconv_mode=1, dconv_depth=2, dconv_comp=4, dconv_attn=4,
                 dconv_lstm=4, dconv_init=1e-4, normalize=True, resample=True, rescale=0.1, samplerate=44100, segment=4 * 10, *args, **kwargs):
        super().__init__()
        self.cfg = cfg = P.DemucsConfig(sources, audio_channels=audio_channels, channels=channels, growth=growth, depth=depth, rewrite=rewrite,
                                        lstm_layers=lstm_layers, kernel_size=kernel_size, stride=stride, context=context, gelu=gelu, glu=glu,
                                        norm_starts=norm_starts, norm_groups=norm_groups, dconv_mode=dconv_mode, dconv_depth=dconv_depth,
                                        dconv_comp=dconv_comp, dconv_attn=dconv_attn, dconv_lstm=dconv_lstm, dconv_init=dconv_init,
                                        normalize=normalize, resample=resample, rescale=rescale)
        if cfg.key() not in _STATIC_CACHE:
            _STATIC_CACHE[cfg.key()] = P.DemucsStatic(cfg)
        self.static = _STATIC_CACHE[cfg.key()]
        self.audio_channels, self.sources, self.kernel_size, self.context, self.stride, self.depth = audio_channels, sources, kernel_size, context, stride, depth
        self.resample, self.channels, self.normalize, self.samplerate, self.segment = resample, channels, normalize, samplerate, segment
        self._tables = None
        self.grad_range_hook = None   # data-parallel: called with (lo, hi, stream) when flat_grads[lo:hi] is final (plan_demucs.backward)
        self._ws_cap = max(1, int(os.environ.get("SEHIP_WS_CACHE", "2")))
        self._build_flat(list_roots=("encoder", "decoder"))
        self.reset_parameters()

    def reset_parameters(self):
        """PyTorch's default initialisers of the reference's layers (Conv1d / ConvTranspose1d / Linear: U(+-1/sqrt(fan_in)) for
        weight and bias; nn.LSTM: U(+-1/sqrt(hidden)); GroupNorm 1 / 0), LayerScale = dconv_init (src/model/demucs.py:64-65),
        LocalState's decay query: weight * 0.01, bias -2 (:230-232), then rescale_module (:123-136, :427-428): every convolution's
        weight and bias divided by sqrt(std(weight) / rescale)."""
        cfg = self.cfg
        with torch.no_grad():
            named = dict(self._params)
            for name, p in self._params:
                leaf = name.rsplit(".", 1)[-1]
                if ".lstm." in name:
                    hid = p.shape[0] // 4
                    p.uniform_(-1 / math.sqrt(hid), 1 / math.sqrt(hid))
                elif leaf == "scale":
                    p.fill_(cfg.dconv_init)
                elif p.dim() >= 2:
                    transposed = p.dim() == 3 and name.startswith("decoder.") and name.endswith(".3.weight")
                    fan_in = (p.shape[1] if not transposed else p.shape[1]) * (p.shape[2] if p.dim() == 3 else 1)
                    bound = 1 / math.sqrt(fan_in)
                    p.uniform_(-bound, bound)
                    b = named.get(name[:-len("weight")] + "bias")
                    if b is not None:
                        b.uniform_(-bound, bound)
                elif leaf == "weight":          # GroupNorm (every 1-D tensor named weight)
                    p.fill_(1.0)
                elif leaf == "bias" and (name[:-len("bias")] + "weight") in named and named[name[:-len("bias")] + "weight"].dim() == 1:
                    p.zero_()
            for name, p in self._params:
                if name.endswith("query_decay.weight"):
                    p.mul_(0.01)
                    named[name[:-len("weight")] + "bias"].fill_(-2.0)
            if cfg.rescale:
                for name, p in self._params:
                    if p.dim() == 3 and name.endswith("weight"):      # nn.Conv1d / nn.ConvTranspose1d
                        scale = (p.std() / cfg.rescale) ** 0.5
                        p.div_(scale)
                        named[name[:-len("weight")] + "bias"].div_(scale)

    def valid_length(self, length):
        return self.cfg.valid_length(length)

    def set_deterministic(self, on=True):
        """The reference's `solver.cudnn_deterministic` switch (src/conf/config.yaml:130, src/utils.py:108-111) for this model.  The work
        is done by the process-wide sehip_set_deterministic (the Solver / sehip.utils.prepare_device switch it on): the GroupNorm
        statistics and backward sums of csrc/demucs.hip add their threads in thread order and their workgroups through slots that a second
        small launch adds in a fixed order (csrc/det.h), the per-channel partial rows are added in row order, the weight gradients take
        the library's fixed-order kernels, the optimizer the unfused tail, and the whole step runs on ONE queue
        (plan_demucs.DemucsWorkspace._select_streams says why).  Two runs of the same steps are then bit-identical
        (tests/test_gpu_deterministic.py)."""
        self._deterministic = bool(on)
        return self

    def load_state_dict(self, state, strict=True, **kw):
        state = dict(state)
        for idx in range(self.depth):    # the reference's own key migration (src/model/demucs.py:492-501)
            for a in ("encoder", "decoder"):
                for b in ("bias", "weight"):
                    new, old = f"{a}.{idx}.3.{b}", f"{a}.{idx}.2.{b}"
                    if old in state and new not in state:
                        state[new] = state.pop(old)
        return super().load_state_dict(state, strict=strict, **kw)

    def workspace(self, batch, nsample):
        dev = self._require_gpu("Demucs")
        if self._tables is None:
            self._tables = P.DemucsDeviceTables(self.static, dev)
        return self._lru_get((batch, nsample), self._ws_cap, lambda: P.DemucsWorkspace(self.static, self._tables, batch, nsample, dev))

    def step_guard(self):
        """Device word the fused optimizer checks (sehip_opt_step_g): the sticky hand-off time-out word of the workspace the last
        forward ran in -- an optimizer step computed after a time-out is a no-op on the device."""
        ws = getattr(self, "_last_ws", None)
        return None if ws is None or not self.static.lstms else ws.lstm_sync[60:61]

    def check_health(self):
        """Called by the Solver wherever it synchronises anyway (loss read-back, checkpoints, end of evaluate()): True if steps were
        lost to a hand-off time-out (the model has switched to the per-step LSTM launches)."""
        from ..distrib import global_flag
        ws = getattr(self, "_last_ws", None)
        if ws is None or not self.static.lstms:
            return global_flag(False, self._flat.device if self._flat is not None else None)
        lost = bool(ws.check_lstm_handoffs(recover=True, global_flag=global_flag))      # collective: the OR over the ranks decides
        self._note_graph_epoch(ws)
        return lost

    def _run_forward(self, mix, need_backward):
        ws = self.workspace(mix.shape[0], mix.shape[-1])
        self._last_ws = ws
        ws.generation += 1
        ws.forward(mix.contiguous().float(), self._flat, need_backward=need_backward)
        self._note_graph_epoch(ws)       # (the workspace looks at its time-out word every 64th forward)
        return ws

    def _run_backward(self, ws, grad_out):
        g = grad_out.contiguous().float()
        hook = self.grad_range_hook if not (self._grads_live and self._params[0][1].grad is not None) else None   # not when accumulating
        tail = self._tail_for_backward()              # FlatOptimizer's accumulators (single replica, not accumulating)
        self._backward_into_flat(lambda dst: ws.backward(g, self._flat, dst, range_ready=hook, tail=tail))
        self._tail_mark(tail)

    def forward(self, mix):
        if mix.dim() != 3 or mix.shape[1] != self.audio_channels:
            raise SehipError(f"Demucs.forward: [B, {self.audio_channels}, T] expected, got {tuple(mix.shape)}")
        if not mix.is_cuda:
            raise SehipError("Demucs.forward got a CPU tensor: the HIP path needs a gfx950 GPU (no CPU fallback)")
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != mix.device:
                self._anchor = torch.zeros(1, device=mix.device, requires_grad=True)
            return _DemucsFunction.apply(self, mix, self._anchor)
        return self._run_forward(mix, need_backward=False).out.clone()
