"""Host-side plan of the ConvTasNet train step on libsehip (reference: src/model/conv_tasnet.py:34-487 with the shipped options
skip=False, norm_type="gLN", non-causal, mask_nonlinear="relu"; BASELINE config C4).

Activations are channels-last bf16 ``[M][K][C]`` (M utterances, K = (T - L)/(L/2) + 1 frames).  Every 1x1 convolution (bottleneck,
the two pointwise convolutions of each of the R*X temporal blocks, the mask convolution) is a dense product of the implicit-GEMM
engine (csrc/gemm.hip; the block's residual add rides in the product's epilogue through the descriptor's `res`), everything
else -- encoder + cLN, PReLU + global LayerNorm, the depthwise dilated convolution, mask * mixture_w + basis + overlap-add -- the
streaming kernels of csrc/tasnet.hip.  Per temporal block the forward pass is 2 products + 3 streams, the backward pass
4 products + 4 streams.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream, SehipError
from .plan import Arena, CGemmDesc, GemmSpec, ParamLayout, bind_chunk_table, enc_entry, BF16
from .plan_dcunet import Buf


class TasNetConfig:
    """Constructor arguments of the reference model (src/model/conv_tasnet.py:35-84)."""

    def __init__(self, sources, N=128, L=40, B=128, H=256, P=3, X=7, R=2, audio_channels=2, norm_type="gLN", causal=False,
                 mask_nonlinear="relu", sample_rate=44100, segment_length=44100 * 2 * 4, skip=False, **_ignored):
        if skip or norm_type != "gLN" or causal or mask_nonlinear not in ("relu", "softmax"):
            raise SehipError("sehip ConvTasNet: the shipped options (skip=False, norm_type='gLN', causal=False) with "
                             "mask_nonlinear='relu' or 'softmax' are built")
        if mask_nonlinear == "softmax" and len(sources) > 8:
            raise SehipError("sehip ConvTasNet: mask_nonlinear='softmax' is built for at most 8 sources")
        self.mask_nonlinear = mask_nonlinear
        if P not in (3, 5, 7):
            raise SehipError("sehip ConvTasNet: kernel size P must be 3, 5 or 7 (the sizes csrc/tasnet.hip instantiates)")
        for name, v in (("N", N), ("B", B), ("H", H)):
            if v % 8 or v < 8:
                raise SehipError(f"sehip ConvTasNet: {name}={v} must be a multiple of 8")
        if L % 2 or L < 2:
            raise SehipError(f"sehip ConvTasNet: L={L} must be even")
        self.sources = list(sources)
        self.C = len(self.sources)
        # (the codec's backward kernels -- csrc/tasnet.hip sehip_ctn_decoder_bwd: either a register kernel for this (ac * L, N) with one
        #  copy of the basis in LDS, or the wave-per-frame kernel with five)
        al = audio_channels * L
        reg = (al == 40 and N <= 128) or (al in (16, 20, 40) and N <= 128 and al * N * 4 <= 64 * 1024) or (al == 16 and N <= 512)
        if N > 512 or al * (N + 1) * 4 + 16 * N > 64 * 1024 or (not reg and (5 * al * N + 4 * al) * 4 > 160 * 1024):
            raise SehipError(f"sehip ConvTasNet: N={N}, L={L}, audio_channels={audio_channels}: the decoder's basis does not fit the LDS")
        self.N, self.L, self.B, self.H, self.P, self.X, self.R = N, L, B, H, P, X, R
        self.audio_channels = audio_channels

    def key(self):
        return (self.C, self.N, self.L, self.B, self.H, self.P, self.X, self.R, self.audio_channels, self.mask_nonlinear)

    def blocks(self):
        return [(r, x) for r in range(self.R) for x in range(self.X)]

    def param_specs(self):
        """[(name, shape, kind)] in the reference's parameters() order (DepthwiseSeparableConv registers pointwise_conv before
        its inner net, src/model/conv_tasnet.py:380-386)."""
        N, L, B, H, P = self.N, self.L, self.B, self.H, self.P
        out = [("encoder.conv1d_U.weight", (N, self.audio_channels, L), "param"),
               ("separator.network.0.gamma", (1, N, 1), "param"), ("separator.network.0.beta", (1, N, 1), "param"),
               ("separator.network.1.weight", (B, N, 1), "param")]
        for r, x in self.blocks():
            q = f"separator.network.2.{r}.{x}.net."
            out += [(q + "0.weight", (H, B, 1), "param"), (q + "1.weight", (1,), "param"),
                    (q + "2.gamma", (1, H, 1), "param"), (q + "2.beta", (1, H, 1), "param"),
                    (q + "3.pointwise_conv.weight", (B, H, 1), "param"),
                    (q + "3.net.0.weight", (H, 1, P), "param"), (q + "3.net.1.weight", (1,), "param"),
                    (q + "3.net.2.gamma", (1, H, 1), "param"), (q + "3.net.2.beta", (1, H, 1), "param")]
        out += [("separator.network.3.weight", (self.C * N, B, 1), "param"),
                ("decoder.basis_signals.weight", (self.audio_channels * L, N), "param")]
        return out


class TasNetStatic:
    """Products, packed-weight layout and gradient un-packing table (independent of batch and clip length)."""

    def __init__(self, cfg: TasNetConfig):
        self.cfg = cfg
        self.layout = L = ParamLayout(cfg)
        ia = L.index_array
        N, B, H = cfg.N, cfg.B, cfg.H
        self.specs = {}

        def dense(name, wname, cin, cout, src, dst, kind, res=None, transposed=False):
            rows = [(0, 0, 0, 8 * q) for q in range(cin // 8)]
            w = ia(wname).reshape(ia(wname).shape[0], ia(wname).shape[1])      # [Cout, Cin]
            widx = w.T.copy() if transposed else w
            sp = GemmSpec(name, rows, widx, np.zeros_like(widx), cout, None, "K", 1, 1, [(src, "all")], [(dst, 0, 1, 0)], kind=kind, res=res)
            self.specs[name] = sp

        # SEHIP_CTN_KEEP_GRADS=1 (tests): every block keeps its own du / dh2 instead of sharing one pair, so that EVERY product and
        # normalisation kernel of the backward pass can be checked op-locally afterwards (tests/test_gpu_convtasnet_fullwidth.py);
        # same kernels, same launches, 2 x 14 more activation-sized buffers.
        self.keep_grads = bool(os.environ.get("SEHIP_CTN_KEEP_GRADS"))
        self.du_name = (lambda b: f"du{b}") if self.keep_grads else (lambda b: "du")
        self.dh2_name = (lambda b: f"dh2_{b}") if self.keep_grads else (lambda b: "dh2")
        net = "separator.network."
        dense("bott.fwd", net + "1.weight", N, B, "cln", "x0", "fwd")
        dense("bott.dg", net + "1.weight", B, N, "dx0", "dcln", "dgrad", transposed=True)
        self.blocks = cfg.blocks()
        for b, (r, x) in enumerate(self.blocks):
            q = f"{net}2.{r}.{x}.net."
            dense(f"b{b}.in.fwd", q + "0.weight", B, H, f"x{b}", f"h1_{b}", "fwd")
            dense(f"b{b}.in.dg", q + "0.weight", H, B, f"dh1_{b}", f"dx{b}", "dgrad", res=f"dx{b + 1}", transposed=True)
            dense(f"b{b}.pw.fwd", q + "3.pointwise_conv.weight", H, B, f"u{b}", f"x{b + 1}", "fwd", res=f"x{b}")
            dense(f"b{b}.pw.dg", q + "3.pointwise_conv.weight", B, H, f"dx{b + 1}", self.du_name(b), "dgrad", transposed=True)
        nb = len(self.blocks)
        dense("mask.fwd", net + "3.weight", B, cfg.C * N, f"x{nb}", "mlin", "fwd")
        dense("mask.dg", net + "3.weight", cfg.C * N, B, "dmlin", f"dx{nb}", "dgrad", transposed=True)
        # where the weight gradient of each forward product reads dOut
        self.dout_of = {"bott.fwd": "dx0", "mask.fwd": "dmlin"}
        for b in range(nb):
            self.dout_of[f"b{b}.in.fwd"] = f"dh1_{b}"      # per block: the weight gradient reads it later, on the side stream
            self.dout_of[f"b{b}.pw.fwd"] = f"dx{b + 1}"

        wa, ga, kta, nta = Arena(64), Arena(16), Arena(1), Arena(1)
        for name, s in self.specs.items():
            s.kt_off = kta.add(s.ktab)
            s.nt_off = nta.add(s.ntab)
            s.w_off = wa.add(enc_entry(s.widx, s.wneg).reshape(-1))
            if s.kind == "fwd":
                s.dw_off = ga.reserve(s.Npad * s.K)
        AL = cfg.audio_channels * cfg.L
        self.enc_g_off = ga.reserve(N * AL + 2 * N)           # dU | dgamma0 | dbeta0
        self.dec_g_off = ga.reserve(AL * N)                   # dV
        self.blk_g_off = []
        for b in range(nb):
            self.blk_g_off.append(dict(gch1=ga.reserve(2 * H + cfg.P * H), gch2=ga.reserve(2 * H), a1=ga.reserve(1), a2=ga.reserve(1)))
        self.n_wpack, self.n_gpack = wa.size, ga.size
        self.wtab = wa.build(np.int32)
        self.ktab = kta.build(np.int32, 4)
        self.ntab = nta.build(np.int32, 4, fill=0)
        self.utab = self._build_unpack_table()

    def _build_unpack_table(self):
        L, cfg = self.layout, self.cfg
        ia = L.index_array
        N, H, P = cfg.N, cfg.H, cfg.P
        ps, gs = [], []
        for s in self.specs.values():
            if s.dw_off is None:
                continue
            m = s.widx >= 0
            ps.append(s.widx[m]); gs.append(s.dw_off + np.flatnonzero(m.reshape(-1)))

        def lin(name, base):
            idx = ia(name).reshape(-1)
            ps.append(idx); gs.append(base + np.arange(idx.size))

        AL = cfg.audio_channels * cfg.L
        lin("encoder.conv1d_U.weight", self.enc_g_off)
        lin("separator.network.0.gamma", self.enc_g_off + N * AL)
        lin("separator.network.0.beta", self.enc_g_off + N * AL + N)
        lin("decoder.basis_signals.weight", self.dec_g_off)
        for b, (r, x) in enumerate(self.blocks):
            q = f"separator.network.2.{r}.{x}.net."
            o = self.blk_g_off[b]
            lin(q + "2.gamma", o["gch1"]); lin(q + "2.beta", o["gch1"] + H); lin(q + "3.net.0.weight", o["gch1"] + 2 * H)
            lin(q + "3.net.2.gamma", o["gch2"]); lin(q + "3.net.2.beta", o["gch2"] + H)
            lin(q + "1.weight", o["a1"]); lin(q + "3.net.1.weight", o["a2"])
        p = np.concatenate(ps).astype(np.int64)
        g = np.concatenate(gs).astype(np.int64)
        assert len(np.unique(p)) == len(p), "every ConvTasNet parameter has exactly one packed-gradient entry"
        tab = np.full((L.n_params, 4), -1, dtype=np.int32)
        tab[p, 0] = (g << 1).astype(np.int32)
        return tab


class TasNetDeviceTables:
    def __init__(self, st: TasNetStatic, device):
        f = lambda a: torch.from_numpy(a).to(device)
        self.wtab, self.utab, self.ntab = f(st.wtab), f(st.utab), f(st.ntab)
        self.tensor_offsets = f(st.layout.tensor_offsets)
        self.utab_g = self.uperm = None                 # the fused tail's un-pack in gather order (plan.gather_ordered_unpack_table)
        if not os.environ.get("SEHIP_NO_UNPACK_PERM"):
            from .plan import gather_ordered_unpack_table
            tg, pm = gather_ordered_unpack_table(st.utab, st.layout.tensor_offsets)
            self.utab_g, self.uperm = f(tg), f(pm)
        self.wpack = torch.zeros(st.n_wpack, dtype=BF16, device=device)


class TasNetWorkspace:
    def __init__(self, st: TasNetStatic, tables: TasNetDeviceTables, M, T, device):
        cfg = st.cfg
        self.st, self.tb, self.M, self.T, self.device = st, tables, M, T, device
        self.generation, self.pinned, self.closed = 0, False, False
        if T < cfg.L:
            raise SehipError(f"ConvTasNet: a clip of {T} samples is shorter than one analysis window (L={cfg.L})")
        self.K = K = (T - cfg.L) // (cfg.L // 2) + 1
        N, B, H = cfg.N, cfg.B, cfg.H
        nb = len(st.blocks)
        self.bufs = {}

        def add(name, c):
            self.bufs[name] = Buf(torch.zeros(M, K, 1, c, dtype=BF16, device=device), K, 1, c)

        add("cln", N); add("dcln", N)
        for b in range(nb + 1):
            add(f"x{b}", B); add(f"dx{b}", B)
        for b in range(nb):
            add(f"h1_{b}", H); add(f"h2_{b}", H); add(f"u{b}", H); add(f"dh1_{b}", H)
        for name in sorted({st.du_name(b) for b in range(nb)} | {st.dh2_name(b) for b in range(nb)}):
            add(name, H)
        add("mlin", cfg.C * N); add("dmlin", cfg.C * N)
        if cfg.mask_nonlinear == "softmax":      # the mask itself, between the mask product's scores and the decoder (sehip_ctn_mask_softmax_fwd)
            add("msoft", cfg.C * N)
        self.w = torch.empty(M, K, N, dtype=torch.float32, device=device)
        self.dw_dec = torch.empty(M, K, N, dtype=torch.float32, device=device)
        self.out = torch.zeros(M, cfg.C, cfg.audio_channels, T, dtype=torch.float32, device=device)
        self.stats = torch.zeros(nb, 2, M, 2, dtype=torch.float64, device=device)      # forward: (sum, sumsq) per block / gLN / utterance
        self.bsums = torch.zeros(nb, 2, M, 2, dtype=torch.float64, device=device)      # backward: (S1, S2)
        self.gpack = torch.zeros(st.n_gpack, dtype=torch.float32, device=device)
        self.gln_scratch = torch.empty(int(_lib.lib().sehip_ctn_gln_bwd_scratch_floats(M, K, H)), dtype=torch.float32, device=device)
        self.codec_scratch = torch.empty(int(_lib.lib().sehip_ctn_codec_bwd_scratch_floats(M, K, N, cfg.L, cfg.audio_channels)),
                                         dtype=torch.float32, device=device)
        self.wav = None
        self._one_clear, self._bwd_clean = not os.environ.get("SEHIP_CTN_TORCH_ZEROS"), False
        self._side_stream = None if os.environ.get("SEHIP_NO_SIDE_STREAM") else torch.cuda.Stream(device=device)
        self.side = self._side_stream      # (None while the deterministic schedule is on: forward())
        self._events, self._event_i, self._chain_dirty = [], 0, True
        self._bind()

    def close(self):
        if self.closed:
            return
        self.closed = True
        lib = _lib.lib()
        for e in self._events:
            lib.sehip_event_destroy(e)
        self._events = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _bind(self):
        st, tb, M, K = self.st, self.tb, self.M, self.K
        self.desc = {}
        kt = st.ktab.copy()
        for name, s in st.specs.items():
            bind_chunk_table(st.ktab, kt, s.kt_off, s.K // 8, [(self.bufs[b].F, self.bufs[b].C) for b, _ in s.srcs])
        self.ktab_dev = torch.from_numpy(kt).to(self.device)
        for name, s in st.specs.items():
            d = CGemmDesc()
            b = self.bufs[s.srcs[0][0]]
            d.src[0].ptr, d.src[0].T, d.src[0].F, d.src[0].C, d.src[0].tlo, d.src[0].thi = b.ptr, K, 1, b.C, 0, K
            o = self.bufs[s.dsts[0][0]]
            d.dst[0].ptr, d.dst[0].T, d.dst[0].F, d.dst[0].C = o.ptr, K, 1, o.C
            d.dst[0].toff, d.dst[0].fmul, d.dst[0].fadd, d.dst[0].is_f32, d.dst[0].tmul = 0, 1, 0, 0, 1
            d.ktab = self.ktab_dev.data_ptr() + 16 * s.kt_off
            d.ntab = tb.ntab.data_ptr() + 16 * s.nt_off
            d.W = tb.wpack.data_ptr() + 2 * s.w_off
            d.M, d.N, d.Npad, d.K = M * K, s.N, s.Npad, s.K
            d.TT, d.J, d.fmul, d.tmul = K, 1, 1, 1
            if s.res is not None:
                d.res = self.bufs[s.res].ptr
            # (every product here is a 1x1 convolution over dense rows: the library may take the streaming dense-row kernels --
            #  csrc/dgemm.hip forward / input gradient, csrc/wgrad3.hip weight gradient -- where the widths are ones they are built for)
            d.dense_rows = 1 if (b.C == s.K and o.C == s.Npad == s.N) else 0
            self.desc[name] = d
            if s.dw_off is not None:
                w = CGemmDesc.from_buffer_copy(d)
                w.dW = self.gpack.data_ptr() + 4 * s.dw_off
                w.dst[0].ptr = self.bufs[st.dout_of[name]].ptr
                w.res = None
                w.dense_rows = 1 if (b.C == s.K and o.C == s.Npad == s.N) else 0     # every product here is a 1x1 convolution over dense rows
                w.wg_hint = 160          # measured at the C4 shape (ms per step): 96: 4.28, 128: 3.68, 160: 3.59, 192: 3.79, 256: 3.75
                self.desc[name + ".wg"] = w
        # round 6: the gLN statistics of every block's first 1x1 output inside the product's launch, where the dense-row kernel takes it
        # (sehip_gemm_desc.gln_stats; SEHIP_CTN_NO_FUSED_GLN=1: the separate sehip_ctn_gln_stats pass)
        lib_ = _lib.lib()
        self.fused_gln = (not os.environ.get("SEHIP_CTN_NO_FUSED_GLN") and len(st.blocks) > 0 and
                          all(int(lib_.sehip_gemm_takes_gln_stats(C.byref(self.desc[f"b{i}.in.fwd"]))) == 1 for i in range(len(st.blocks))))

    def gemm(self, name):
        self._chain_dirty = True
        call("sehip_gemm", C.byref(self.desc[name]), stream())

    def _event(self):
        if not self._events:
            for _ in range(16):
                e = _lib.lib().sehip_event_create()
                if not e:
                    raise SehipError("sehip_event_create: " + _lib.lib().sehip_last_error().decode())
                self._events.append(e)
        self._event_i = (self._event_i + 1) % len(self._events)
        return self._events[self._event_i]

    def wgrad(self, name):
        main = torch.cuda.current_stream()
        if self.side is None or torch.cuda.is_current_stream_capturing():
            call("sehip_wgrad", C.byref(self.desc[name + ".wg"]), main.cuda_stream)
            return
        if self._chain_dirty:
            call("sehip_stream_depend", self.side.cuda_stream, main.cuda_stream, self._event())
            self._chain_dirty = False
        call("sehip_wgrad", C.byref(self.desc[name + ".wg"]), self.side.cuda_stream)

    def _pp(self, params, name):
        return params.data_ptr() + 4 * self.st.layout.param_off[name][0]

    def forward(self, wav, params):
        """wav [M, ac, T] fp32 on device -> self.out [M, C, ac, T]."""
        st, cfg, b, tb = self.st, self.st.cfg, self.bufs, self.tb
        M, K, N, H = self.M, self.K, cfg.N, cfg.H
        pp = lambda n: self._pp(params, n)
        self.wav = wav
        # every accumulator of the step cleared by ONE launch (round 6: they were four runtime fill kernels -- stats, out, gpack, bsums --
        # with 30-55 us of host gap in front of each on the chain; SEHIP_CTN_TORCH_ZEROS=1 for those)
        nbytes = lambda t: t.numel() * t.element_size()
        if self._one_clear:
            call("sehip_zero_regions", ptr(self.stats), nbytes(self.stats), ptr(self.out), nbytes(self.out), ptr(self.gpack), nbytes(self.gpack),
                 ptr(self.bsums), nbytes(self.bsums), stream())
            self._bwd_clean = True
        else:
            self.stats.zero_()
        call("sehip_pack_bf16", ptr(params), ptr(tb.wtab), st.n_wpack, ptr(tb.wpack), stream())
        net = "separator.network."
        # (the deterministic schedule -- solver.cudnn_deterministic, sehip.utils.set_deterministic -- keeps the statistics out of the product's
        #  launch: its double atomics arrive in a varying order; sehip_ctn_gln_stats adds the workgroups' sums in a fixed one)
        det = bool(_lib.lib().sehip_get_deterministic())
        fused_gln = self.fused_gln and not det
        # ... and runs the step on ONE queue: with the weight gradients on the second stream two runs were not bit-identical in a few
        # passes of ten although every reduction has a fixed order (plan_demucs.DemucsWorkspace._select_streams)
        self.side = None if det else self._side_stream
        call("sehip_ctn_encoder_fwd", ptr(wav), pp("encoder.conv1d_U.weight"), pp(net + "0.gamma"), pp(net + "0.beta"), M,
             cfg.audio_channels, self.T, N, cfg.L, ptr(self.w), b["cln"].ptr, stream())
        self.gemm("bott.fwd")
        for i, (r, x) in enumerate(st.blocks):
            q = f"{net}2.{r}.{x}.net."
            s1 = self.stats[i, 0].data_ptr(); s2 = self.stats[i, 1].data_ptr()
            if fused_gln:           # the product's launch also takes the gLN statistics of what it stores (csrc/dgemm.hip)
                d = self.desc[f"b{i}.in.fwd"]
                d.gln_stats, d.gln_slope = s1, pp(q + "1.weight")
                self.gemm(f"b{i}.in.fwd")
            else:
                self.gemm(f"b{i}.in.fwd")
                call("sehip_ctn_gln_stats", b[f"h1_{i}"].ptr, pp(q + "1.weight"), M, K, H, s1, stream())
            call("sehip_ctn_dwconv_fwd", b[f"h1_{i}"].ptr, pp(q + "1.weight"), s1, pp(q + "2.gamma"), pp(q + "2.beta"),
                 pp(q + "3.net.0.weight"), cfg.P, 2 ** x, pp(q + "3.net.1.weight"), M, K, H, b[f"h2_{i}"].ptr, s2, stream())
            call("sehip_ctn_gln_apply", b[f"h2_{i}"].ptr, pp(q + "3.net.1.weight"), s2, pp(q + "3.net.2.gamma"), pp(q + "3.net.2.beta"),
                 M, K, H, b[f"u{i}"].ptr, stream())
            self.gemm(f"b{i}.pw.fwd")
        self.gemm("mask.fwd")
        if not self._one_clear:
            self.out.zero_()
        mk = b["mlin"]
        if cfg.mask_nonlinear == "softmax":
            call("sehip_ctn_mask_softmax_fwd", b["mlin"].ptr, M * K, cfg.C, N, b["msoft"].ptr, stream())
            mk = b["msoft"]
        call("sehip_ctn_decoder_fwd", ptr(self.w), mk.ptr, pp("decoder.basis_signals.weight"), M, K, N, cfg.L, cfg.audio_channels,
             cfg.C, self.T, ptr(self.out), stream())
        return self.out

    def backward(self, dout, params, grads, tail=None):
        """dout [M, C, ac, T] fp32 -> flat parameter gradients (overwritten)."""
        st, cfg, b, tb = self.st, self.st.cfg, self.bufs, self.tb
        M, K, N, H = self.M, self.K, cfg.N, cfg.H
        pp = lambda n: self._pp(params, n)
        gp = lambda off: self.gpack.data_ptr() + 4 * off
        nb = len(st.blocks)
        net = "separator.network."
        if not getattr(self, "_bwd_clean", False):      # (a second backward pass over the same forward, or the torch-fill switch)
            self.gpack.zero_()
            self.bsums.zero_()
        self._bwd_clean = False
        self._chain_dirty = True
        soft = cfg.mask_nonlinear == "softmax"
        call("sehip_ctn_decoder_bwd", ptr(dout), ptr(self.w), (b["msoft"] if soft else b["mlin"]).ptr, pp("decoder.basis_signals.weight"), M, K, N,
             cfg.L, cfg.audio_channels, cfg.C, self.T, b["dmlin"].ptr, ptr(self.dw_dec), gp(st.dec_g_off), ptr(self.codec_scratch), stream())
        if soft:        # what came back is the gradient of the mask: through the softmax, in place
            call("sehip_ctn_mask_softmax_bwd", b["msoft"].ptr, b["dmlin"].ptr, M * K, cfg.C, N, stream())
        self.wgrad("mask.fwd")
        self.gemm("mask.dg")
        for i in range(nb - 1, -1, -1):
            r, x = st.blocks[i]
            q = f"{net}2.{r}.{x}.net."
            o = st.blk_g_off[i]
            self.wgrad(f"b{i}.pw.fwd")
            self.gemm(f"b{i}.pw.dg")
            du, dh2 = b[st.du_name(i)], b[st.dh2_name(i)]
            call("sehip_ctn_gln_bwd", du.ptr, b[f"h2_{i}"].ptr, pp(q + "3.net.1.weight"), self.stats[i, 1].data_ptr(),
                 pp(q + "3.net.2.gamma"), pp(q + "3.net.2.beta"), pp(q + "3.net.0.weight"), cfg.P, 2 ** x, 0, M, K, H,
                 self.bsums[i, 1].data_ptr(), gp(o["gch2"]), dh2.ptr, gp(o["a2"]), ptr(self.gln_scratch), stream())
            call("sehip_ctn_gln_bwd", dh2.ptr, b[f"h1_{i}"].ptr, pp(q + "1.weight"), self.stats[i, 0].data_ptr(),
                 pp(q + "2.gamma"), pp(q + "2.beta"), pp(q + "3.net.0.weight"), cfg.P, 2 ** x, 1, M, K, H,
                 self.bsums[i, 0].data_ptr(), gp(o["gch1"]), b[f"dh1_{i}"].ptr, gp(o["a1"]), ptr(self.gln_scratch), stream())
            self._chain_dirty = True
            self.wgrad(f"b{i}.in.fwd")
            self.gemm(f"b{i}.in.dg")
        self.wgrad("bott.fwd")
        self.gemm("bott.dg")
        call("sehip_ctn_encoder_bwd", ptr(self.wav), ptr(self.w), b["dcln"].ptr, ptr(self.dw_dec), pp(net + "0.gamma"), M,
             cfg.audio_channels, self.T, N, cfg.L, gp(st.enc_g_off), ptr(self.codec_scratch), stream())
        if self.side is not None and not torch.cuda.is_current_stream_capturing():
            call("sehip_stream_depend", stream(), self.side.cuda_stream, self._event())
        if tail is not None:         # FlatOptimizer's accumulators: the un-pack also takes the clipping norm / metric sums (plan.DCCRNWorkspace.backward)
            if tb.uperm is not None:
                call("sehip_unpack_grad_sums_perm", ptr(self.gpack), ptr(tb.utab_g), ptr(tb.uperm), st.layout.n_params, ptr(grads), tail[2],
                     tail[3], tail[0], tail[1], tail[4], None, stream())
            else:
                call("sehip_unpack_grad_sums", ptr(self.gpack), ptr(tb.utab), st.layout.n_params, ptr(grads), tail[2], tail[3], tail[0],
                     tail[1], tail[4], None, stream())
        else:
            call("sehip_unpack_grad", ptr(self.gpack), ptr(tb.utab), st.layout.n_params, ptr(grads), stream())
        return grads
