"""Host-side plan of the Demucs train step on libsehip (reference: src/model/demucs.py:272-501; BASELINE config C3).

Activations are channels-last bf16 ``[B][T][C]``.  Every convolution / linear layer is a product of the implicit-GEMM engine
(csrc/gemm.hip):
  * ``Conv1d(k=8, stride=4)`` (:386) reads its input as frames of FOUR samples, ``[B][T/4][4 Cin]``: output frame i needs quad
    frames i and i+1, i.e. a stride-1 two-tap convolution with K = 8 Cin -- no strided gather; its input gradient is the
    mirror image, ONE product with N = 4 Cin whose destination is the same quad view;
  * ``ConvTranspose1d(k=8, stride=4)`` (:413) the other way round: N = 4 Cout columns = the four output phases, K = 2 C (input
    frames i and i-1), destination = the quad view of the output;
  * the dilated k=3 convolutions of DConv (:191) and the decoder's context convolution (:408): three taps, K = 3 C;
  * 1x1 convolutions, the LSTM input projections (both directions in one product, N = 8 H), Linear(2H -> H), the four 1x1
    convolutions of LocalState that read x (query | key | content | decay: one product) and its projection: dense.
Residual / skip additions ride in product epilogues (descriptor field `res`) or in the activation kernels.  Everything else is
csrc/demucs.hip.  Built: the constructor defaults' structure (rewrite, GLU, GELU, context=1, kernel 8 / stride 4, DConv in the
encoder, no central LSTM).  The BLSTM's overlapping chunks (sequences longer than max_steps = 200 frames, :91-117) are a gather
before and a pick after the LSTM products (sehip_dmx_frames): the LSTM then runs on B * ceil(T / 100) sequences of 200 frames.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream, SehipError
from .plan import Arena, CGemmDesc, ParamLayout, bind_chunk_table, dense_ntab, npad_of, pad_ktab, BF16
from .plan_dcunet import Buf

HEADS, NDECAY, MAX_STEPS, GN_EPS = 4, 4, 200, 1e-5


def resample_kernels(old_sr, new_sr, zeros=24, rolloff=0.945):
    """Interpolation kernels of julius.resample_frac(x, old_sr, new_sr) (julius 0.2.7 ResampleFrac, restated: windowed sinc,
    `zeros` zero crossings, squared-cosine window, unit DC gain): float32 [new_sr][2 width + old_sr] and width."""
    g = math.gcd(old_sr, new_sr)
    old_sr, new_sr = old_sr // g, new_sr // g
    sr = min(new_sr, old_sr) * rolloff
    width = math.ceil(zeros * old_sr / sr)
    idx = np.arange(-width, width + old_sr, dtype=np.float32)
    ks = []
    for i in range(new_sr):
        t = (np.float32(-i / new_sr) + idx / np.float32(old_sr)) * np.float32(sr)
        t = np.clip(t, -zeros, zeros) * np.float32(math.pi)
        window = np.cos(t / np.float32(zeros) / np.float32(2)) ** 2
        safe = np.where(t == 0, np.float32(1), t)
        k = np.where(t == 0, np.float32(1), np.sin(safe) / safe) * window
        ks.append((k / k.sum(dtype=np.float32)).astype(np.float32))
    return np.stack(ks), width


class DemucsConfig:
    """Constructor arguments of the reference model (src/model/demucs.py:273-309)."""

    def __init__(self, sources, audio_channels=2, channels=64, growth=2.0, depth=6, rewrite=True, lstm_layers=0, kernel_size=8, stride=4,
                 context=1, gelu=True, glu=True, norm_starts=4, norm_groups=4, dconv_mode=1, dconv_depth=2, dconv_comp=4, dconv_attn=4,
                 dconv_lstm=4, dconv_init=1e-4, normalize=True, resample=True, rescale=0.1, **_ignored):
        bad = []
        if not rewrite: bad.append("rewrite=False")
        if lstm_layers: bad.append("lstm_layers>0")
        if kernel_size != 8 or stride != 4: bad.append("kernel_size/stride other than 8/4")
        if context != 1: bad.append("context!=1")
        if not gelu or not glu: bad.append("gelu/glu=False")
        if dconv_mode != 1: bad.append("dconv_mode!=1")
        if dconv_depth < 1: bad.append("dconv_depth<1")
        if bad:
            raise SehipError("sehip Demucs: only the shipped structure is built (src/model/demucs.py:273-309 defaults); unsupported: " + ", ".join(bad))
        self.sources = list(sources)
        self.S = len(self.sources)
        self.audio_channels, self.channels, self.growth, self.depth = audio_channels, channels, growth, depth
        self.norm_starts, self.norm_groups = norm_starts, norm_groups
        self.dconv_depth, self.dconv_comp, self.dconv_attn, self.dconv_lstm, self.dconv_init = dconv_depth, dconv_comp, dconv_attn, dconv_lstm, dconv_init
        self.normalize, self.resample, self.rescale = bool(normalize), bool(resample), rescale
        self.acp = max(2, (audio_channels + 1) // 2 * 2)
        self.co = self.S * audio_channels
        self.cop = max(2, (self.co + 1) // 2 * 2)
        for i, (cin, ch) in enumerate(self.layer_channels()):
            hid = int(ch / dconv_comp)
            if ch % 16 or hid % 8 or hid < 8:
                raise SehipError(f"sehip Demucs: layer {i}: {ch} channels / DConv width {hid}: channel counts must be multiples of 16 (8 inside DConv)")
            if i >= norm_starts and (ch % norm_groups or (ch // norm_groups) % 8):
                raise SehipError(f"sehip Demucs: layer {i}: GroupNorm({norm_groups}) groups must hold a multiple of 8 channels")
            if i >= dconv_lstm and hid % 32:
                raise SehipError(f"sehip Demucs: layer {i}: the BLSTM width {hid} must be a multiple of 32")
            if i >= dconv_attn and (hid % HEADS or (hid // HEADS) % 8):
                raise SehipError(f"sehip Demucs: layer {i}: a LocalState head must hold a multiple of 8 channels (width {hid})")

    def key(self):
        return (tuple(self.sources), self.audio_channels, self.channels, self.growth, self.depth, self.norm_starts, self.norm_groups,
                self.dconv_depth, self.dconv_comp, self.dconv_attn, self.dconv_lstm, self.normalize, self.resample)

    def layer_channels(self):
        out, cin, ch = [], self.audio_channels, self.channels
        for _ in range(self.depth):
            out.append((cin, ch))
            cin, ch = ch, int(self.growth * ch)
        return out

    def valid_length(self, length):
        """src/model/demucs.py:430-451"""
        if self.resample:
            length *= 2
        for _ in range(self.depth):
            length = max(1, math.ceil((length - 8) / 4) + 1)
        for _ in range(self.depth):
            length = (length - 1) * 4 + 8
        if self.resample:
            length = math.ceil(length / 2)
        return int(length)

    def dconv_layout(self, i):
        """Sequential indices inside one DConv layer (src/model/demucs.py:190-201)."""
        j, out = 3, {}
        if i >= self.dconv_lstm:
            out["lstm"] = j; j += 1
        if i >= self.dconv_attn:
            out["attn"] = j; j += 1
        out["conv2"], out["norm2"], out["scale"] = j, j + 1, j + 3
        return out

    def param_specs(self):
        """[(name, shape, "param")] in the reference's parameters() order."""
        out = []
        add = lambda n, s: out.append((n, tuple(s), "param"))

        def conv(n, co, ci, k):
            add(n + ".weight", (co, ci, k)); add(n + ".bias", (co,))

        def gn(n, c):
            add(n + ".weight", (c,)); add(n + ".bias", (c,))

        enc, dec = [], []
        for i, (cin, ch) in enumerate(self.layer_channels()):
            mark = len(out)
            q = f"encoder.{i}"
            normed = i >= self.norm_starts
            conv(q + ".0", ch, cin, 8)
            if normed:
                gn(q + ".1", ch)
            hid = int(ch / self.dconv_comp)
            lay = self.dconv_layout(i)
            for d in range(self.dconv_depth):
                p = f"{q}.3.layers.{d}."
                conv(p + "0", hid, ch, 3); gn(p + "1", hid)
                if "lstm" in lay:
                    b = f"{p}{lay['lstm']}."
                    for l in range(2):
                        for sfx in ("", "_reverse"):
                            add(f"{b}lstm.weight_ih_l{l}{sfx}", (4 * hid, hid if l == 0 else 2 * hid))
                            add(f"{b}lstm.weight_hh_l{l}{sfx}", (4 * hid, hid))
                            add(f"{b}lstm.bias_ih_l{l}{sfx}", (4 * hid,)); add(f"{b}lstm.bias_hh_l{l}{sfx}", (4 * hid,))
                    add(b + "linear.weight", (hid, 2 * hid)); add(b + "linear.bias", (hid,))
                if "attn" in lay:
                    a = f"{p}{lay['attn']}."
                    conv(a + "content", hid, hid, 1); conv(a + "query", hid, hid, 1); conv(a + "key", hid, hid, 1)
                    conv(a + "query_decay", HEADS * NDECAY, hid, 1); conv(a + "proj", hid, hid, 1)
                conv(f"{p}{lay['conv2']}", 2 * ch, hid, 1); gn(f"{p}{lay['norm2']}", 2 * ch)
                add(f"{p}{lay['scale']}.scale", (ch,))
            conv(q + ".4", 2 * ch, ch, 1)
            if normed:
                gn(q + ".5", 2 * ch)
            enc.append(out[mark:]); del out[mark:]
            q = f"decoder.{self.depth - 1 - i}"
            cout = cin if i > 0 else self.co
            conv(q + ".0", 2 * ch, ch, 3)
            if normed:
                gn(q + ".1", 2 * ch)
            add(q + ".3.weight", (ch, cout, 8)); add(q + ".3.bias", (cout,))
            if i > 0 and normed:
                gn(q + ".4", cout)
            dec.append(out[mark:]); del out[mark:]
        for e in enc:
            out.extend(e)
        for d in reversed(dec):
            out.extend(d)
        return out


class Prod:
    """One product of the engine (batch- and length-independent part).  src / dst: (buffer name, quad view?); tt: (level, extra)
    = number of row frames T[level] + extra; rows: [(frame offset, channel offset)] one per 8-channel chunk of K."""

    def __init__(self, name, rows, widx, src, dst, tt, bias=None, res=None, kind="fwd", dout=None, ntab_base=0, wg_only=False, framed=False):
        self.name, self.src, self.dst, self.tt, self.res, self.kind, self.dout, self.wg_only = name, src, dst, tt, res, kind, dout, wg_only
        self.framed = framed     # rows are the BLSTM's chunks: B * nf items of W frames instead of B items of T[level] frames
        self.ktab, self.K = pad_ktab([(0, fo, 0, co) for fo, co in rows])
        n, k0 = widx.shape
        self.N, self.Npad = n, npad_of(n)
        w = np.full((self.Npad, self.K), -1, dtype=np.int32)
        w[:n, :k0] = np.where(widx < 0, -1, widx.astype(np.int64) << 1).astype(np.int32)
        self.wtab = w.reshape(-1)
        self.bias = None
        if bias is not None:
            b = np.full((self.Npad, 2), -1, dtype=np.int32)
            bi = np.asarray(bias, dtype=np.int64).reshape(n, -1)
            b[:n, :bi.shape[1]] = np.where(bi < 0, -1, bi << 1).astype(np.int32)
            self.bias = b
        self.ntab = dense_ntab(n, self.Npad, 0, ntab_base)
        self.w_off = self.b_off = self.dw_off = self.db_off = self.kt_off = self.nt_off = None


def _chunks(frame_off, c0, cn):
    assert cn % 8 == 0 and c0 % 8 == 0
    return [(frame_off, c0 + 8 * q) for q in range(cn // 8)]


class DemucsStatic:
    """Products, packed-weight layout and gradient un-packing table (independent of batch and clip length)."""

    def __init__(self, cfg: DemucsConfig):
        self.cfg = cfg
        self.layout = L = ParamLayout(cfg)
        if L.n_params >= 2 ** 30:
            raise SehipError("sehip Demucs: more than 2^30 parameters do not fit the 32-bit packing tables")
        self.prods = {}
        self.buffers = {}        # name -> (level, channels, dtype, framed)   level -1 = the network input length
        self.norms = []          # (key, y buffer, C, G or 0, mode, gamma name, beta name, scale name)
        self.lstms, self.attns = [], []
        self.gch = {}            # norm key -> offset of {dgamma | dbeta | dscale} in the packed gradients
        self._ga = Arena(16)
        ia = L.index_array
        chans = cfg.layer_channels()
        D = cfg.depth

        def buf(name, level, c, dtype=BF16, framed=False):
            self.buffers[name] = (level, c, dtype, framed)

        def prod(*a, **k):
            p = Prod(*a, **k)
            assert p.name not in self.prods
            self.prods[p.name] = p
            return p

        def dense(name, wname, bname, src, dst, level, res=None, dg_src=None, dg_dst=None, dg_res=None, dout=None, framed=False):
            """1x1 convolution / linear layer: forward and (optionally) input-gradient products"""
            w = ia(wname)
            w = w.reshape(w.shape[0], w.shape[1])
            prod(name, _chunks(0, 0, w.shape[1]), w, (src, False), (dst, False), (level, 0), bias=ia(bname) if bname else None, res=res, dout=dout,
                 framed=framed)
            if dg_src is not None:
                prod(name + ".dg", _chunks(0, 0, w.shape[0]), w.T.copy(), (dg_src, False), (dg_dst, False), (level, 0), res=dg_res, kind="dgrad",
                     framed=framed)

        def conv3(name, wname, bname, src, dst, level, dil, dg_src, dg_dst, dg_res=None, dout=None):
            w = ia(wname)                                       # [N][Cin][3]
            n, cin, _ = w.shape
            rows = [r for p in range(3) for r in _chunks((p - 1) * dil, 0, cin)]
            prod(name, rows, w.transpose(0, 2, 1).reshape(n, 3 * cin), (src, False), (dst, False), (level, 0), bias=ia(bname), dout=dout)
            rows = [r for p in range(3) for r in _chunks(-(p - 1) * dil, 0, n)]
            prod(name + ".dg", rows, w.transpose(1, 2, 0).reshape(cin, 3 * n), (dg_src, False), (dg_dst, False), (level, 0), res=dg_res, kind="dgrad")

        def norm(key, y, c, g, mode, pre, scale=None):
            self.norms.append(key)
            co = c // 2 if mode else c
            self.gch[key] = dict(y=y, C=c, G=g, mode=mode, gamma=pre + "weight" if g else None, beta=pre + "bias" if g else None, scale=scale,
                                 off=self._ga.reserve(2 * c + co) if g else None, Co=co)

        buf("x", -1, cfg.acp)
        for i, (cin, ch) in enumerate(chans):
            hid = int(ch / cfg.dconv_comp)
            cinp = cfg.acp if i == 0 else cin
            normed = i >= cfg.norm_starts
            G = cfg.norm_groups if normed else 0
            e = f"e{i}."
            q = f"encoder.{i}."
            src_in = "x" if i == 0 else f"e{i - 1}.out"
            # ---- strided convolution on the quad view of its input
            w = ia(q + "0.weight")                              # [ch][cin][8]
            wp = np.full((ch, cinp, 8), -1, dtype=np.int64)
            wp[:, :cin] = w
            rows = _chunks(0, 0, 4 * cinp) + _chunks(1, 0, 4 * cinp)
            buf(e + "y", i, ch); buf(e + "dy", i, ch); buf(e + "a", i, ch)
            prod(e + "conv", rows, wp.transpose(0, 2, 1).reshape(ch, 8 * cinp), (src_in, True), (e + "y", False), (i, 0), bias=ia(q + "0.bias"),
                 dout=e + "dy")
            if i > 0:
                # dX[4 i' + q'][c] = sum_j sum_n dY[i' - j][n] W[n][c][4 j + q']
                rows = _chunks(0, 0, ch) + _chunks(-1, 0, ch)
                wd = w.reshape(ch, cin, 2, 4).transpose(3, 1, 2, 0).reshape(4 * cin, 2 * ch)
                prod(e + "conv.dg", rows, wd, (e + "dy", False), (f"e{i - 1}.dout", True), (i, 1), res=f"d{i - 1}.din", kind="dgrad")
            norm(e + "n0", e + "y", ch, G, 0, q + "1.")
            # ---- DConv
            lay = cfg.dconv_layout(i)
            xin = e + "a"
            for d in range(cfg.dconv_depth):
                p = f"{q}3.layers.{d}."
                k = f"{e}d{d}."
                dil = 2 ** d
                for nm, c_ in (("y1", hid), ("dy1", hid), ("h1", hid), ("dh1", hid), ("y2", 2 * ch), ("dy2", 2 * ch), ("x", ch), ("dx", ch)):
                    buf(k + nm, i, c_)
                dx_in = f"{e}d{d - 1}.dx" if d > 0 else e + "da"
                conv3(k + "c1", p + "0.weight", p + "0.bias", xin, k + "y1", i, dil, k + "dy1", dx_in, dg_res=k + "dx", dout=k + "dy1")
                norm(k + "n1", k + "y1", hid, 1, 0, p + "1.")
                last, dlast = k + "h1", k + "dh1"
                if "lstm" in lay:
                    b = f"{p}{lay['lstm']}."
                    for nm, c_, dt in (("pre0", 8 * hid, torch.float32), ("pre1", 8 * hid, torch.float32), ("hs0", 2 * hid, BF16), ("hs1", 2 * hid, BF16),
                                       ("cs0", 2 * hid, torch.float32), ("cs1", 2 * hid, torch.float32), ("dG0", 8 * hid, BF16), ("dG1", 8 * hid, BF16),
                                       ("dhs0", 2 * hid, BF16), ("dhs1", 2 * hid, BF16), ("fr", hid, BF16), ("dfr", hid, BF16), ("h2f", hid, BF16),
                                       ("dh2f", hid, BF16)):
                        buf(k + nm, i, c_, dt, framed=True)
                    buf(k + "h2", i, hid); buf(k + "dh2", i, hid)
                    for l in range(2):
                        wih = np.concatenate([ia(f"{b}lstm.weight_ih_l{l}"), ia(f"{b}lstm.weight_ih_l{l}_reverse")])          # [8H][in]
                        bias = np.stack([np.concatenate([ia(f"{b}lstm.bias_ih_l{l}"), ia(f"{b}lstm.bias_ih_l{l}_reverse")]),
                                         np.concatenate([ia(f"{b}lstm.bias_hh_l{l}"), ia(f"{b}lstm.bias_hh_l{l}_reverse")])], axis=1)
                        src_l = k + "fr" if l == 0 else k + "hs0"
                        prod(f"{k}ih{l}", _chunks(0, 0, wih.shape[1]), wih, (src_l, False), (f"{k}pre{l}", False), (i, 0), bias=bias, dout=f"{k}dG{l}",
                             framed=True)
                        dst_l = k + "dfr" if l == 0 else k + "dhs0"
                        prod(f"{k}ih{l}.dg", _chunks(0, 0, 8 * hid), wih.T.copy(), (f"{k}dG{l}", False), (dst_l, False), (i, 0), kind="dgrad", framed=True)
                        for dr, sfx in enumerate(("", "_reverse")):
                            # dW_hh = sum_t dG[t]^T h[t -+ 1]: weight-gradient product only
                            prod(f"{k}hh{l}.{dr}", _chunks(1 if dr else -1, dr * hid, hid), ia(f"{b}lstm.weight_hh_l{l}{sfx}"), (f"{k}hs{l}", False),
                                 (f"{k}dG{l}", False), (i, 0), dout=f"{k}dG{l}", ntab_base=dr * 4 * hid, wg_only=True, framed=True)
                    # (the skip connection h2 = Linear(...) + h1 is added where the chunks are put back together)
                    dense(k + "lin", b + "linear.weight", b + "linear.bias", k + "hs1", k + "h2f", i, dg_src=k + "dh2f", dg_dst=k + "dhs1",
                          dout=k + "dh2f", framed=True)
                    self.lstms.append(dict(key=k, level=i, H=hid, pre=b))
                    last, dlast = k + "h2", k + "dh2"
                if "attn" in lay:
                    a = f"{p}{lay['attn']}."
                    nq = 3 * hid + HEADS * NDECAY
                    buf(k + "qkv", i, nq); buf(k + "dqkv", i, nq)
                    buf(k + "r", i, hid); buf(k + "dr", i, hid); buf(k + "h3", i, hid); buf(k + "dh3", i, hid)
                    names = ("query", "key", "content", "query_decay")
                    w = np.concatenate([ia(f"{a}{n}.weight")[:, :, 0] for n in names])
                    bias = np.concatenate([ia(f"{a}{n}.bias") for n in names])
                    prod(k + "qkv", _chunks(0, 0, hid), w, (last, False), (k + "qkv", False), (i, 0), bias=bias, dout=k + "dqkv")
                    prod(k + "qkv.dg", _chunks(0, 0, nq), w.T.copy(), (k + "dqkv", False), (dlast, False), (i, 0), res=k + "dh3", kind="dgrad")
                    dense(k + "proj", a + "proj.weight", a + "proj.bias", k + "r", k + "h3", i, res=last, dg_src=k + "dh3", dg_dst=k + "dr", dout=k + "dh3")
                    self.attns.append(dict(key=k, level=i, hid=hid, nq=nq))
                    last, dlast = k + "h3", k + "dh3"
                dense(k + "c2", f"{p}{lay['conv2']}.weight", f"{p}{lay['conv2']}.bias", last, k + "y2", i, dg_src=k + "dy2", dg_dst=dlast, dout=k + "dy2")
                norm(k + "n2", k + "y2", 2 * ch, 1, 1, f"{p}{lay['norm2']}.", scale=f"{p}{lay['scale']}.scale")
                xin = k + "x"
            # ---- rewrite
            buf(e + "da", i, ch); buf(e + "yr", i, 2 * ch); buf(e + "dyr", i, 2 * ch); buf(e + "out", i, ch); buf(e + "dout", i, ch)
            dense(e + "rw", q + "4.weight", q + "4.bias", xin, e + "yr", i, dg_src=e + "dyr", dg_dst=f"{e}d{cfg.dconv_depth - 1}.dx", dout=e + "dyr")
            norm(e + "n3", e + "yr", 2 * ch, G, 1, q + "5.")
            # ---- decoder of the same index
            k = f"d{i}."
            q = f"decoder.{D - 1 - i}."
            cout = cin if i > 0 else cfg.co
            coutp = cin if i > 0 else cfg.cop
            buf(k + "in", i, ch); buf(k + "din", i, ch); buf(k + "yd", i, 2 * ch); buf(k + "dyd", i, 2 * ch); buf(k + "g", i, ch); buf(k + "dg", i, ch)
            conv3(k + "rw", q + "0.weight", q + "0.bias", k + "in", k + "yd", i, 1, k + "dyd", k + "din", dout=k + "dyd")
            norm(k + "n0", k + "yd", 2 * ch, G, 1, q + "1.")
            w = ia(q + "3.weight")                              # [ch][cout][8]
            wp = np.full((ch, coutp, 8), -1, dtype=np.int64)
            wp[:, :cout] = w
            bp = np.full((4, coutp), -1, dtype=np.int64)
            bp[:, :cout] = ia(q + "3.bias")[None]
            last = i == 0
            buf(k + "yt", i - 1, coutp, torch.float32 if last else BF16); buf(k + "dyt", i - 1, coutp)
            # out[4 i' + q'][co] = sum_j sum_c g[i' - j][c] W[c][co][4 j + q']
            rows = _chunks(0, 0, ch) + _chunks(-1, 0, ch)
            prod(k + "ct", rows, wp.reshape(ch, coutp, 2, 4).transpose(3, 1, 2, 0).reshape(4 * coutp, 2 * ch), (k + "g", False), (k + "yt", True), (i, 1),
                 bias=bp.reshape(-1), dout=k + "dyt")
            rows = _chunks(0, 0, 4 * coutp) + _chunks(1, 0, 4 * coutp)
            prod(k + "ct.dg", rows, wp.transpose(0, 2, 1).reshape(ch, 8 * coutp), (k + "dyt", True), (k + "dg", False), (i, 0), kind="dgrad")
            if i > 0:
                norm(k + "n1", k + "yt", cout, cfg.norm_groups if normed else 0, 0, q + "4.")

        # ---- arenas
        wa, ba, kta, nta = Arena(64), Arena(4), Arena(1), Arena(1)
        ga = self._ga
        self.whh = {}
        # packed operands in three runs: what the forward pass reads (packed on the chain's stream at the start of the step) | what only
        # the backward pass reads (packed on the second stream, under the forward pass) | the recurrent weights as weight-gradient
        # placeholders (never read on the device: their table only feeds the un-packing map)
        enc = lambda a: (a.astype(np.int64) << 1).astype(np.int32).reshape(-1)
        hh = {}
        for ls in self.lstms:
            b = ls["pre"]
            for l in range(2):
                hh[(ls["key"], l)] = np.stack([ia(f"{b}lstm.weight_hh_l{l}"), ia(f"{b}lstm.weight_hh_l{l}_reverse")])    # [2][4H][H]
        for p in self.prods.values():
            p.kt_off = kta.add(p.ktab)
            p.nt_off = nta.add(p.ntab)
            if p.bias is not None:
                p.b_off = ba.add(p.bias)
            if p.kind == "fwd":
                p.dw_off = ga.reserve(p.Npad * p.K)
                if p.bias is not None:
                    p.db_off = ga.reserve(p.Npad)
        self.late_level = max(0, D - 2)          # the two deepest levels hold 94 % of the weights and are reached last
        for p in self.prods.values():
            if p.kind == "fwd" and not p.wg_only and p.tt[0] < self.late_level:
                p.w_off = wa.add(p.wtab)
        self.n_wpack_head = wa.size
        for p in self.prods.values():
            if p.kind == "fwd" and not p.wg_only and p.tt[0] >= self.late_level:
                p.w_off = wa.add(p.wtab)
        fwd_hh = {k: wa.add(enc(w)) for k, w in hh.items()}
        self.n_wpack_fwd = wa.size
        for p in self.prods.values():
            if p.kind != "fwd" and p.tt[0] < self.late_level:
                p.w_off = wa.add(p.wtab)
        self.n_wpack_bwd_head = wa.size
        for p in self.prods.values():
            if p.kind != "fwd" and p.tt[0] >= self.late_level:
                p.w_off = wa.add(p.wtab)
        for k, w in hh.items():
            self.whh[k] = (fwd_hh[k], wa.add(enc(w.transpose(0, 2, 1))))
        self.n_wpack_dev = wa.size
        for p in self.prods.values():
            if p.wg_only:
                p.w_off = wa.add(p.wtab)
        self.n_wpack, self.n_bpack, self.n_gpack = wa.size, max(ba.size, 4), ga.size
        self.wtab = wa.build(np.int32)
        self.btab = ba.build(np.int32, 2) if ba.size else np.full((4, 2), -1, dtype=np.int32)
        self.ktab = kta.build(np.int32, 4)
        self.ntab = nta.build(np.int32, 4, fill=0)
        for p in self.prods.values():
            p.wtab = None        # the arena holds the only copy from here on
        wa.pieces = []
        self.utab1, self.ulist, self.utab4 = self._build_unpack_table()
        self.runs, self.side = self._build_pack_runs()
        if cfg.resample:
            self.kup, self.wup = resample_kernels(1, 2)
            self.kdn, self.wdn = resample_kernels(2, 1)

    def _build_unpack_table(self):
        L = self.layout
        ia = L.index_array
        tab = np.full((L.n_params, 4), -1, dtype=np.int32)
        fill = np.zeros(L.n_params, dtype=np.int8)

        def put(pidx, gidx, unique=False):
            pidx = np.asarray(pidx, dtype=np.int64).reshape(-1)
            gidx = np.asarray(gidx, dtype=np.int64).reshape(-1)
            if pidx.size == 0:
                return
            assert gidx.max() < 2 ** 30
            if unique:                                      # a weight element occurs once in its forward product
                slot = fill[pidx]
                assert slot.max() < 4
                tab[pidx, slot] = (gidx << 1).astype(np.int32)
                fill[pidx] += 1
                return
            order = np.argsort(pidx, kind="stable")        # a parameter may occur several times in one call (the four output
            ps, gs = pidx[order], gidx[order]               # phases of a transposed convolution share its bias)
            first = np.flatnonzero(np.r_[True, ps[1:] != ps[:-1]])
            counts = np.diff(np.r_[first, ps.size])
            rank = np.arange(ps.size) - np.repeat(first, counts)
            slot = fill[ps] + rank
            assert slot.max() < 4
            tab[ps, slot] = (gs << 1).astype(np.int32)
            fill[ps[first]] += counts.astype(np.int8)

        for p in self.prods.values():
            if p.dw_off is None:
                continue
            w = self.wtab[p.w_off:p.w_off + p.Npad * p.K]
            m = np.flatnonzero(w >= 0)
            put(w[m] >> 1, p.dw_off + m, unique=True)
            if p.db_off is not None:
                for col in range(p.bias.shape[1]):
                    b = p.bias[:, col]
                    m = np.flatnonzero(b >= 0)
                    put(b[m] >> 1, p.db_off + m)
        for key, n in self.gch.items():
            if n["off"] is None:
                continue
            c = n["C"]
            put(ia(n["gamma"]), n["off"] + np.arange(c)); put(ia(n["beta"]), n["off"] + c + np.arange(c))
            if n["scale"]:
                put(ia(n["scale"]), n["off"] + 2 * c + np.arange(n["Co"]))
        used = np.zeros(L.n_params, dtype=bool)
        for name in L.param_names:
            off, shape = L.param_off[name]
            used[off:off + int(np.prod(shape))] = True
        assert (fill[used] >= 1).all(), "every Demucs parameter has a packed-gradient entry"
        # compact form: one 4-byte entry per parameter + the few parameters with several entries (the biases of the transposed
        # convolutions: one per output phase) as an index list with 16-byte entries
        self.unpack_entries = fill
        multi = np.flatnonzero(fill > 1).astype(np.int32)
        return np.ascontiguousarray(tab[:, 0]), multi, np.ascontiguousarray(tab[multi])

    def _build_pack_runs(self):
        """(base, stride) per 8 packed elements where they are an affine run of parameters, (-1, 0) for 8 padding zeros, otherwise
        (-2 - k, 0) and the 8 ordinary entries in row k of the side table (sehip_pack_bf16_runs)."""
        n = self.n_wpack_dev
        assert n % 8 == 0
        w = self.wtab[:n].reshape(-1, 8)                       # int32 throughout (indices < 2^30), in slabs of 4 M rows
        runs = np.zeros((w.shape[0], 2), dtype=np.int32)
        affine = np.zeros(w.shape[0], dtype=bool)
        nonev = np.zeros(w.shape[0], dtype=bool)
        for a in range(0, w.shape[0], 1 << 22):
            ws = w[a:a + (1 << 22)]
            valid = ws >= 0
            idx = ws >> 1
            d = idx[:, 1:] - idx[:, :-1]
            af = valid.all(axis=1) & (d == d[:, :1]).all(axis=1)
            affine[a:a + ws.shape[0]] = af
            nonev[a:a + ws.shape[0]] = ~valid.any(axis=1)
            r = runs[a:a + ws.shape[0]]
            r[af, 0] = idx[af, 0]
            r[af, 1] = d[af, 0]
        runs[nonev, 0] = -1
        irr = np.flatnonzero(~affine & ~nonev)
        runs[irr, 0] = -2 - np.arange(irr.size)
        side = np.ascontiguousarray(self.wtab[:n].reshape(-1, 8)[irr]).reshape(-1)
        if side.size == 0:
            side = np.full(8, -1, dtype=np.int32)
        # The table is stored sorted by base address inside each separately packed segment (sehip_pack_bf16_runs_to): neighbouring
        # lanes then read neighbouring parameters.  pack_dst[i] = the run entry i produces.
        self.pack_dst = np.arange(runs.shape[0], dtype=np.int32)
        if not os.environ.get("SEHIP_NO_PACK_SORT"):
            cuts = sorted({0, self.n_wpack_head // 8, self.n_wpack_fwd // 8, self.n_wpack_bwd_head // 8, n // 8})
            for a, b_ in zip(cuts[:-1], cuts[1:]):
                key = np.where(runs[a:b_, 0] >= 0, runs[a:b_, 0].astype(np.int64), np.int64(1) << 40)
                order = np.argsort(key, kind="stable")
                runs[a:b_] = runs[a:b_][order]
                self.pack_dst[a:b_] = (order + a).astype(np.int32)
        return runs, side.astype(np.int32)


class DemucsDeviceTables:
    def __init__(self, st: DemucsStatic, device):
        f = lambda a: torch.from_numpy(a).to(device)
        self.btab, self.ntab = f(st.btab), f(st.ntab)
        self.runs, self.side = f(st.runs), f(st.side)                  # weight packing: (base, stride) per 8 elements
        self.pack_dst = f(st.pack_dst)
        self.utab1, self.ulist, self.utab4 = f(st.utab1), f(st.ulist), f(st.utab4 if st.utab4.size else np.full((1, 4), -1, dtype=np.int32))
        self.tensor_offsets = f(st.layout.tensor_offsets)
        self.wpack = torch.zeros(st.n_wpack, dtype=BF16, device=device)
        self.bpack = torch.zeros(st.n_bpack, dtype=torch.float32, device=device)
        if st.cfg.resample:
            self.kup, self.kdn = f(st.kup.reshape(-1)), f(st.kdn.reshape(-1))


class DemucsWorkspace:
    def __init__(self, st: DemucsStatic, tables: DemucsDeviceTables, B, T, device):
        cfg = st.cfg
        self.st, self.tb, self.B, self.T, self.device = st, tables, B, T, device
        self.generation, self.pinned, self.closed = 0, False, False
        self.Tv = cfg.valid_length(T)
        self.padl = (self.Tv - T) // 2
        self.Tin = 2 * self.Tv if cfg.resample else self.Tv
        lens, n = [], self.Tin
        for _ in range(cfg.depth):
            if n < 8 or (n - 8) % 4:
                raise SehipError(f"Demucs: a clip of {T} samples does not reach depth {cfg.depth}")
            n = (n - 8) // 4 + 1
            lens.append(n)
        self.lens = lens                     # T_i = frames of encoder i's output; level -1 = Tin
        # BLSTM chunks per level: (nf chunks per item, W frames each, hop S); a sequence of at most max_steps frames is one chunk
        self.chunks = [(math.ceil(n / (MAX_STEPS // 2)), MAX_STEPS, MAX_STEPS // 2) if n > MAX_STEPS else (1, n, n) for n in lens]
        for a in st.attns:
            if lens[a["level"]] > 580:
                raise SehipError(f"Demucs: {lens[a['level']]} frames at the LocalState attention of layer {a['level']}: the score tile of all keys "
                                 f"does not fit the LDS (at most 580 frames)")
        self.bufs = {}
        for name, (level, c, dt, framed) in st.buffers.items():
            frames = self.Tin if level < 0 else lens[level]
            items = B
            if framed:
                nf, frames, _ = self.chunks[level]
                items = B * nf
            self.bufs[name] = Buf(torch.zeros(items, frames, 1, c, dtype=dt, device=device), frames, 1, c)
        self.ms = torch.zeros(B, 2, dtype=torch.float32, device=device)
        self.ms_acc = torch.zeros(B, 2, dtype=torch.float64, device=device)
        self.out = torch.zeros(B, cfg.S, cfg.audio_channels, T, dtype=torch.float32, device=device)
        nn_ = len(st.norms)
        self.norm_idx = {k: j for j, k in enumerate(st.norms)}
        self.stats = torch.zeros(nn_, B, 8, 2, dtype=torch.float64, device=device)
        self.sums = torch.zeros(nn_, B, 8, 2, dtype=torch.float64, device=device)
        self.gpack = torch.zeros(st.n_gpack, dtype=torch.float32, device=device)
        self.dc = torch.zeros(2 * B * max([ls["H"] * self.chunks[ls["level"]][0] for ls in st.lstms] + [1]), dtype=torch.float32, device=device)
        self.attn_slabs = torch.empty(max([int(_lib.lib().sehip_dmx_attn_bwd_scratch_floats(B, lens[a["level"]], a["hid"])) for a in st.attns] + [1]),
                                      dtype=torch.float32, device=device)      # key / content gradients per query tile (LocalState backward)
        self.lstm_sync = torch.zeros(int(_lib.lib().sehip_dmx_lstm_sync_bytes()) // 4, dtype=torch.int32, device=device)   # arrival counters + time-out word
        self._side_stream = None if os.environ.get("SEHIP_NO_SIDE_STREAM") else torch.cuda.Stream(device=device)
        self.side = self._side_stream      # (None while the deterministic schedule is on: _select_streams)
        self.comm = None     # third stream: early un-pack + all-reduce of finished gradient ranges (data-parallel runs only)
        self._events, self._event_i, self._chain_dirty = [], 0, True
        self._held_events = []
        self._bind()

    def close(self):
        if self.closed:
            return
        self.closed = True
        lib = _lib.lib()
        for e in self._events + self._held_events:
            lib.sehip_event_destroy(e)
        self._events, self._held_events = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, name, quad):
        b = self.bufs[name]
        if not quad:
            return b.Tst, b.C
        assert b.Tst % 4 == 0
        return b.Tst // 4, 4 * b.C

    def _bind(self):
        st, tb, B = self.st, self.tb, self.B
        self.desc = {}
        kt = st.ktab.copy()
        for p in st.prods.values():
            bind_chunk_table(st.ktab, kt, p.kt_off, p.K // 8, [(1, self._view(*p.src)[1])])
        self.ktab_dev = torch.from_numpy(kt).to(self.device)
        for name, p in st.prods.items():
            d = CGemmDesc()
            sb = self.bufs[p.src[0]]
            sT, sC = self._view(*p.src)
            d.src[0].ptr, d.src[0].T, d.src[0].F, d.src[0].C, d.src[0].tlo, d.src[0].thi = sb.ptr, sT, 1, sC, 0, sT
            ob = self.bufs[p.dst[0]]
            oT, oC = self._view(*p.dst)
            d.dst[0].ptr, d.dst[0].T, d.dst[0].F, d.dst[0].C = ob.ptr, oT, 1, oC
            d.dst[0].toff, d.dst[0].fmul, d.dst[0].fadd, d.dst[0].tmul = 0, 1, 0, 1
            d.dst[0].is_f32 = 1 if ob.t.dtype == torch.float32 else 0
            tt = (self.Tin if p.tt[0] < 0 else self.lens[p.tt[0]]) + p.tt[1]
            items = B
            if p.framed:
                nf, tt, _ = self.chunks[p.tt[0]]
                items = B * nf
            assert tt <= oT, (name, tt, oT)
            d.ktab = self.ktab_dev.data_ptr() + 16 * p.kt_off
            d.ntab = tb.ntab.data_ptr() + 16 * p.nt_off
            d.W = tb.wpack.data_ptr() + 2 * p.w_off
            if p.b_off is not None:
                d.bias = tb.bpack.data_ptr() + 4 * p.b_off
            d.M, d.N, d.Npad, d.K = items * tt, p.N, p.Npad, p.K
            d.TT, d.J, d.fmul, d.tmul = tt, 1, 1, 1
            if p.res is not None:
                rb = self.bufs[p.res]
                assert rb.Tst * rb.C == ob.Tst * ob.C and rb.t.dtype == BF16 and ob.t.dtype == BF16, (name, p.res)
                d.res = rb.ptr
            if not p.wg_only:
                self.desc[name] = d
            if p.dw_off is not None:
                w = CGemmDesc.from_buffer_copy(d)
                w.dW = self.gpack.data_ptr() + 4 * p.dw_off
                w.dbias = self.gpack.data_ptr() + 4 * p.db_off if p.db_off is not None else None
                gb = self.bufs[p.dout]
                assert gb.Tst * gb.C == ob.Tst * ob.C and gb.t.dtype == BF16, (name, p.dout)
                w.dst[0].ptr = gb.ptr
                w.dst[0].is_f32 = 0
                w.res = None
                self.desc[name + ".wg"] = w
        self._bind_dense_wgrads()

    def _bind_dense_wgrads(self):
        """The weight gradients the streaming dense-row kernel takes (csrc/dtw.hip: every product here is dense rows of a [B][T][C] or
        quad-view tensor at a few frame offsets): tables per product, built once per binding (sehip_wgrad_dense_group_prepare reads
        the chunk table back), one scratch for the partial tiles of whichever launch is running (they share the second stream).
        SEHIP_DMX_NO_DENSE_WGRAD=1: the table-gathered generic kernel for all of them (round 4)."""
        self._dtw, self._dtw_scratch = {}, None
        if os.environ.get("SEHIP_DMX_NO_DENSE_WGRAD") or os.environ.get("SEHIP_NO_DENSE_GROUP"):
            return
        lib = _lib.lib()
        nbytes, need = int(lib.sehip_wgrad_dense_group_bytes(1)), 1
        only = os.environ.get("SEHIP_DMX_DENSE_ONLY")          # tools: comma-separated product names
        for name, w in self.desc.items():
            if not name.endswith(".wg") or (only and name[:-3] not in only.split(",")):
                continue
            if w.K < 128 and not only:       # 64 columns of A in a 256- (64-) column tile: measured slower than the generic kernel
                continue                     # (e0.rw 177 vs 156 us, e0.d*.c2 157 vs 142, e1.d*.c2 107 vs 81)
            arr = (CGemmDesc * 1)(CGemmDesc.from_buffer_copy(w))
            dbuf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            info = (C.c_int * 8)()
            call("sehip_wgrad_dense_group_prepare", C.cast(arr, C.c_void_p), 1, ptr(dbuf), dbuf.numel(), C.cast(info, C.c_void_p))
            if info[0] == 1:
                self._dtw[name] = (dbuf, info)
                need = max(need, info[4] + (info[5] << 31))
        self._dtw_scratch = torch.empty(need, dtype=torch.float32, device=self.device)

    def _select_streams(self):
        """The deterministic schedule (solver.cudnn_deterministic / sehip.utils.set_deterministic) runs the step on ONE queue.  With the
        weight gradients on the second stream two runs of the same step were NOT bit-identical although every reduction has a fixed order:
        in a few passes of ten, a per-utterance sum of a normalisation kernel came out 1e-4 ... 2e-3 (relative) off when a workgroup of the
        streaming dense-row weight-gradient kernel (csrc/dtw.hip: LDS-DMA operands, 96 KB of LDS) shared its CU, and was exact again when
        the same call was repeated right behind it on the same operands; never with that kernel's workgroups taking a CU's whole LDS,
        never on one queue (tools/dev/det_diff.py, tools/dev/det_actbwd.py, tools/micro/lds_dma_canary.hip).  Not
        understood; one queue is what is bit-stable."""
        # (SEHIP_DET_FORCE_SIDE=1, tools/dev/det_diff.py: keep the second stream under the deterministic schedule -- to reproduce the above)
        one_queue = _lib.lib().sehip_get_deterministic() and not os.environ.get("SEHIP_DET_FORCE_SIDE")
        self.side = None if one_queue else self._side_stream

    def _launch_wgrad(self, name, st):
        h = self._dtw.get(name + ".wg")
        skip = os.environ.get("SEHIP_DET_SKIP_WGRAD")       # tools/dev/det_diff.py bisection (wrong gradients): all | generic | dense
        if skip and (skip == "all" or (skip == "dense") == (h is not None)):
            return
        if h is not None:
            call("sehip_wgrad_dense_group", ptr(h[0]), 1, C.cast(h[1], C.c_void_p), ptr(self._dtw_scratch), st)
        else:
            call("sehip_wgrad", C.byref(self.desc[name + ".wg"]), st)

    # ---- launches -----------------------------------------------------------------------------------------------------
    def gemm(self, name):
        self._chain_dirty = True
        call("sehip_gemm", C.byref(self.desc[name]), stream())

    def _own_event(self, j):
        """Events that stay recorded across many launches (the round-robin pool of _event() would re-use them)."""
        while len(self._held_events) <= j:
            e = _lib.lib().sehip_event_create()
            if not e:
                raise SehipError("sehip_event_create: " + _lib.lib().sehip_last_error().decode())
            self._held_events.append(e)
        return self._held_events[j]

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
            self._launch_wgrad(name, main.cuda_stream)
            return
        if self._chain_dirty:
            call("sehip_stream_depend", self.side.cuda_stream, main.cuda_stream, self._event())
            self._chain_dirty = False
        self._launch_wgrad(name, self.side.cuda_stream)

    def _pp(self, params, name):
        return params.data_ptr() + 4 * self.st.layout.param_off[name][0]

    def _norm_fwd(self, key, params, out, resid=None, add=None):
        n, j = self.st.gch[key], self.norm_idx[key]
        y = self.bufs[n["y"]]
        B, T, Cc = self.B, y.Tst, n["C"]
        sp = None
        if n["G"]:
            sp = self.stats[j].data_ptr()
            call("sehip_dmx_gn_stats", y.ptr, B, T, Cc, n["G"], sp, stream())
        call("sehip_dmx_act_fwd", y.ptr, sp, self._pp(params, n["gamma"]) if n["G"] else None, self._pp(params, n["beta"]) if n["G"] else None,
             max(n["G"], 1), GN_EPS, n["mode"], self._pp(params, n["scale"]) if n["scale"] else None, self.bufs[resid].ptr if resid else None,
             self.bufs[add].ptr if add else None, B, T, Cc, self.bufs[out].ptr, stream())
        self._chain_dirty = True

    def _norm_bwd(self, key, params, dz, dy):
        n, j = self.st.gch[key], self.norm_idx[key]
        y = self.bufs[n["y"]]
        g = bool(n["G"])
        call("sehip_dmx_act_bwd", self.bufs[dz].ptr, y.ptr, self.stats[j].data_ptr() if g else None, self._pp(params, n["gamma"]) if g else None,
             self._pp(params, n["beta"]) if g else None, max(n["G"], 1), GN_EPS, n["mode"], self._pp(params, n["scale"]) if n["scale"] else None,
             self.B, y.Tst, n["C"], self.sums[j].data_ptr() if g else None, self.gpack.data_ptr() + 4 * n["off"] if g else None,
             self.bufs[dy].ptr, stream())
        self._chain_dirty = True

    def check_lstm_handoffs(self, recover=False, global_flag=None):
        """Reads the sticky time-out word of the persistent LSTM kernels (the read waits for the stream).  A time-out means the
        launches since then produced garbage; the fused optimizer never applied it (the word is its device-side guard,
        sehip_opt_step_g).  recover=False raises; recover=True switches THIS model to one launch per time step
        (dmx_lstm_step_*: no inter-workgroup hand-offs), clears the word and returns True -- the caller lost the steps since the
        time-out and carries on from unchanged parameters."""
        tripped = int(self.lstm_sync[60]) != 0
        if global_flag is not None:      # data parallel: every rank switches its launch path at the same call
            tripped = global_flag(tripped, self.lstm_sync.device)
        if not tripped:
            return False
        if not recover:
            raise SehipError("Demucs: a hand-off spin of the persistent LSTM kernels timed out (results since then are invalid and no "
                             "optimizer step was applied); set SEHIP_DMX_LSTM_STEPS=1 to use one launch per time step")
        import warnings
        warnings.warn("sehip Demucs: a hand-off spin of the persistent LSTM kernels timed out; the optimizer steps since then were "
                      "skipped on the device; falling back to one launch per LSTM time step for this model")
        self.st.lstm_per_step = True
        self.lstm_sync[60:62].zero_()
        self.graph_epoch = getattr(self, "graph_epoch", 0) + 1      # launches captured from this workspace are stale
        return True

    def _sync_arg(self):
        """sync block for sehip_dmx_lstm_fwd / _bwd: NULL selects the per-step launches"""
        return None if getattr(self.st, "lstm_per_step", False) else ptr(self.lstm_sync)

    def forward(self, mix, params, need_backward=True):
        """mix [B, ac, T] fp32 on device -> self.out [B, S, ac, T].  need_backward=False (inference): the operands only the backward
        pass reads are not packed."""
        st, cfg, b, tb = self.st, self.st.cfg, self.bufs, self.tb
        B = self.B
        self._select_streams()
        if st.lstms and self.generation % 64 == 2 and not torch.cuda.is_current_stream_capturing():
            # Every 64th call of THIS workspace (the read waits for the previous step).  The counter is per workspace, so under data
            # parallelism the ranks reach this line at different forwards (validation clips of different lengths, a partial last
            # batch, an LRU-evicted workspace restarting at 0): nothing collective may happen here.  A single process
            # recovers at once; with several ranks the word just stays set -- the step guard (made global before every optimizer
            # launch) keeps skipping the steps on every rank -- until the Solver's own synchronisation points, which every rank
            # reaches together, take the decision (model.check_health(): the OR over the ranks).
            import torch.distributed as _dist
            if not (_dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1):
                self.check_lstm_handoffs(recover=True)
        self.stats.zero_()
        # weight packing (134 M parameters in both operand orientations: 2.5 ms of table-driven gathers).  Only the shallow levels'
        # forward operands are packed on the chain's stream; the two deepest levels' (94 % of the weights, reached after most of
        # the encoder) go to the second stream under the shallow encoder levels, and the operands only the backward pass reads
        # (the transposed copies) under the shallow decoder levels -- never beside the deep levels' own products, which are
        # latency-bound and ran 3x slower next to a packing kernel.
        two = self.side is not None and not torch.cuda.is_current_stream_capturing()
        head = st.n_wpack_head if two else (st.n_wpack_dev if need_backward else st.n_wpack_fwd)
        self._pack(params, 0, head, stream())
        call("sehip_pack_f32", ptr(params), ptr(tb.btab), st.n_bpack, ptr(tb.bpack), stream())
        up = 1 if cfg.resample else 0
        call("sehip_dmx_prep", ptr(mix), B, cfg.audio_channels, cfg.acp, self.T, self.padl, self.Tv, 1 if cfg.normalize else 0, up,
             ptr(tb.kup) if up else None, st.wup if up else 0, st.kup.shape[1] if up else 0, ptr(self.ms_acc), ptr(self.ms), b["x"].ptr, stream())
        # (the second stream starts packing behind the resampling kernel: beside it that kernel ran 3.6x slower)
        self._late_pack_event = self._bwd_pack_events = None
        if two:
            sd = self.side.cuda_stream
            call("sehip_stream_depend", sd, stream(), self._event())
            self._pack(params, head, st.n_wpack_fwd, sd)
            self._late_pack_event = self._own_event(0)
            call("sehip_event_record", self._late_pack_event, sd)
        D = cfg.depth
        for i in range(D):
            e = f"e{i}."
            if i == st.late_level and self._late_pack_event is not None:
                call("sehip_stream_wait_event", stream(), self._late_pack_event)
            self.gemm(e + "conv")
            self._norm_fwd(e + "n0", params, e + "a")
            xin = e + "a"
            for d in range(cfg.dconv_depth):
                k = f"{e}d{d}."
                self.gemm(k + "c1")
                self._norm_fwd(k + "n1", params, k + "h1")
                last = k + "h1"
                if (k + "ih0") in st.prods:
                    H, T = b[k + "h1"].C, b[k + "h1"].Tst
                    nf, W, S = self.chunks[i]
                    call("sehip_dmx_frames", 0, b[k + "h1"].ptr, None, B, T, H, nf, W, S, b[k + "fr"].ptr, stream())
                    for l in range(2):
                        self.gemm(f"{k}ih{l}")
                        woff = st.whh[(k, l)][0]
                        call("sehip_dmx_lstm_fwd", b[f"{k}pre{l}"].ptr, tb.wpack.data_ptr() + 2 * woff, B * nf, W, H, b[f"{k}hs{l}"].ptr,
                             b[f"{k}cs{l}"].ptr, self._sync_arg(), stream())
                    self.gemm(k + "lin")
                    call("sehip_dmx_frames", 1, b[k + "h2f"].ptr, b[k + "h1"].ptr, B, T, H, nf, W, S, b[k + "h2"].ptr, stream())
                    last = k + "h2"
                if (k + "qkv") in st.prods:
                    hid, T = b[k + "r"].C, b[k + "r"].Tst
                    self.gemm(k + "qkv")
                    call("sehip_dmx_attn_fwd", b[k + "qkv"].ptr, B, T, hid, HEADS, NDECAY, b[k + "qkv"].C, b[k + "r"].ptr, stream())
                    self.gemm(k + "proj")
                self.gemm(k + "c2")
                self._norm_fwd(k + "n2", params, k + "x", resid=xin)
                xin = k + "x"
            self.gemm(e + "rw")
            self._norm_fwd(e + "n3", params, e + "out")
        top = f"e{D - 1}.out"
        call("sehip_dmx_add", b[top].ptr, b[top].ptr, b[top].t.numel(), b[f"d{D - 1}.in"].ptr, stream())
        for i in range(D - 1, -1, -1):
            k = f"d{i}."
            if need_backward and two and (i == st.late_level - 1 or (st.late_level == 0 and i == 0 and self._bwd_pack_events is None)):
                self._pack_backward_operands(params)
            self.gemm(k + "rw")
            self._norm_fwd(k + "n0", params, k + "g")
            self.gemm(k + "ct")
            if i > 0:
                self._norm_fwd(k + "n1", params, f"d{i - 1}.in", add=f"e{i - 1}.out")
        if need_backward and two and self._bwd_pack_events is None:
            self._pack_backward_operands(params)
        yt = b["d0.yt"]
        call("sehip_dmx_post", yt.ptr, ptr(self.ms), B, cfg.co, cfg.cop, yt.Tst, self.padl, self.T, up, ptr(tb.kdn) if up else None,
             st.wdn if up else 0, st.kdn.shape[1] if up else 0, ptr(self.out), stream())
        return self.out

    def _hand_over(self, lo, hi, grads, range_ready):
        """Data-parallel hook: grads[lo:hi] is final once the chain and the weight-gradient stream have drained as they stand
        now; it is un-packed on a third stream, which waits for both, and handed to range_ready(lo, hi, stream) -- the
        exchange of the decoder's 63 M and the deepest encoder layers' gradients then runs under the rest of the backward pass."""
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=self.device)
        cs = self.comm.cuda_stream
        call("sehip_stream_depend", cs, stream(), self._event())
        if self.side is not None:
            call("sehip_stream_depend", cs, self.side.cuda_stream, self._event())
        self._unpack(lo, hi, grads, cs)
        range_ready(lo, hi, self.comm)

    def _pack(self, params, lo, hi, on_stream):
        """packed bf16 operands [lo, hi) (multiples of 8) from the flat parameters"""
        tb = self.tb
        call("sehip_pack_bf16_runs_to", ptr(params), tb.runs.data_ptr() + lo, tb.pack_dst.data_ptr() + lo // 2, ptr(tb.side), hi - lo,
             tb.wpack.data_ptr(), on_stream)

    def _unpack(self, lo, hi, grads, on_stream, tail=None):
        """flat parameter gradients [lo, hi) (multiples of 4) from the packed-gradient buffer.  tail (whole vector only): FlatOptimizer's
        accumulators -- the un-pack also takes the clipping norm's and the metric's sums (plan.DCCRNWorkspace.backward)"""
        tb, st = self.tb, self.st
        if tail is not None:
            assert lo == 0 and hi == st.layout.n_params
            call("sehip_unpack_grad1_sums", ptr(self.gpack), ptr(tb.utab1), hi, ptr(grads), tail[2], tail[3], tail[0], tail[1], tail[4],
                 (self.lstm_sync[60:61].data_ptr() if self.st.lstms else None), on_stream)      # guard: the sticky hand-off word
            if st.ulist.size:
                call("sehip_unpack_grad_list_sums", ptr(self.gpack), ptr(tb.ulist), ptr(tb.utab4), int(st.ulist.size), ptr(grads), tail[2],
                     tail[3], tail[0], tail[1], on_stream)
            return
        call("sehip_unpack_grad1", ptr(self.gpack), tb.utab1.data_ptr() + 4 * lo, hi - lo, grads.data_ptr() + 4 * lo, on_stream)
        a, b = np.searchsorted(st.ulist, [lo, hi])
        if b > a:
            call("sehip_unpack_grad_list", ptr(self.gpack), tb.ulist.data_ptr() + 4 * int(a), tb.utab4.data_ptr() + 16 * int(a), int(b - a),
                 ptr(grads), on_stream)

    def _pack_backward_operands(self, params):
        """Second stream: the shallow levels' backward operands, then the deep levels' (an event after each)."""
        st, tb, sd = self.st, self.tb, self.side.cuda_stream
        a, m, z = st.n_wpack_fwd, st.n_wpack_bwd_head, st.n_wpack_dev
        ev = []
        for j, (lo, hi) in enumerate(((a, m), (m, z))):
            if hi > lo:
                self._pack(params, lo, hi, sd)
            e = self._own_event(1 + j)
            call("sehip_event_record", e, sd)
            ev.append(e)
        self._bwd_pack_events = ev

    def backward(self, dout, params, grads, range_ready=None, tail=None):
        """dout [B, S, ac, T] fp32 -> flat parameter gradients (overwritten).
        range_ready(lo, hi, stream): called as soon as grads[lo:hi] is final ON `stream` (a torch stream); the ranges tile
        [0, n_params): the decoder's parameters after the decoder's backward pass, then the encoder layers that hold at least
        5 % of the parameters one by one, the remainder at the end."""
        st, cfg, b, tb = self.st, self.st.cfg, self.bufs, self.tb
        B, D = self.B, cfg.depth
        n_params = st.layout.n_params
        enc_off = [st.layout.param_off[f"encoder.{i}.0.weight"][0] for i in range(D)] + [st.layout.param_off["decoder.0.0.weight"][0]]
        done_from = n_params
        handed = []                                    # (lo, hi) of every range given to range_ready: must tile [0, n_params)
        if range_ready is not None:
            # the early hand-overs rest on the flat layout: encoder.i contiguous and increasing, the decoder the tail.  A layout
            # change must fail here, not all-reduce half-written gradients.
            if not getattr(st, "_dp_layout_checked", False):
                if any(a >= b_ for a, b_ in zip(enc_off[:-1], enc_off[1:])) or enc_off[0] != 0:
                    raise SehipError(f"Demucs data-parallel hand-over: encoder offsets {enc_off} are not strictly increasing from 0")
                for nm in st.layout.param_names:
                    off = st.layout.param_off[nm][0]
                    want = "decoder." if off >= enc_off[D] else f"encoder.{max(i for i in range(D) if enc_off[i] <= off)}."
                    if not nm.startswith(want):
                        raise SehipError(f"Demucs data-parallel hand-over: parameter {nm} at offset {off} is outside its layer's range")
                st._dp_layout_checked = True
            inner = range_ready

            def range_ready(lo_, hi_, st_):
                handed.append((lo_, hi_))
                inner(lo_, hi_, st_)
        up = 1 if cfg.resample else 0
        self.gpack.zero_()
        self.sums.zero_()
        self._chain_dirty = True
        bwd_ev = getattr(self, "_bwd_pack_events", None)     # the backward operands were packed on the second stream
        self._bwd_pack_events = None
        if bwd_ev:
            call("sehip_stream_wait_event", stream(), bwd_ev[0])
        yt = b["d0.dyt"]
        call("sehip_dmx_post_bwd", ptr(dout), ptr(self.ms), B, cfg.co, cfg.cop, yt.Tst, self.padl, self.T, up, ptr(tb.kdn) if up else None,
             st.wdn if up else 0, st.kdn.shape[1] if up else 0, yt.ptr, stream())
        for i in range(D):
            k = f"d{i}."
            if bwd_ev and i == st.late_level:
                call("sehip_stream_wait_event", stream(), bwd_ev[1])
            if i > 0:
                self._norm_bwd(k + "n1", params, f"d{i - 1}.din", k + "dyt")
            self.wgrad(k + "ct")
            self.gemm(k + "ct.dg")
            self._norm_bwd(k + "n0", params, k + "dg", k + "dyd")
            self.wgrad(k + "rw")
            self.gemm(k + "rw.dg")
        top = f"d{D - 1}.din"
        call("sehip_dmx_add", b[top].ptr, b[top].ptr, b[top].t.numel(), b[f"e{D - 1}.dout"].ptr, stream())
        if range_ready is not None:
            self._hand_over(enc_off[D], n_params, grads, range_ready)
            done_from = enc_off[D]
        for i in range(D - 1, -1, -1):
            e = f"e{i}."
            self._norm_bwd(e + "n3", params, e + "dout", e + "dyr")
            self.wgrad(e + "rw")
            self.gemm(e + "rw.dg")
            for d in range(cfg.dconv_depth - 1, -1, -1):
                k = f"{e}d{d}."
                self._norm_bwd(k + "n2", params, k + "dx", k + "dy2")
                self.wgrad(k + "c2")
                self.gemm(k + "c2.dg")
                if (k + "qkv") in st.prods:
                    hid, T = b[k + "r"].C, b[k + "r"].Tst
                    self.wgrad(k + "proj")
                    self.gemm(k + "proj.dg")
                    call("sehip_dmx_attn_bwd", b[k + "qkv"].ptr, b[k + "dr"].ptr, B, T, hid, HEADS, NDECAY, b[k + "qkv"].C, ptr(self.attn_slabs),
                         b[k + "dqkv"].ptr, stream())
                    self._chain_dirty = True
                    self.wgrad(k + "qkv")
                    self.gemm(k + "qkv.dg")
                if (k + "ih0") in st.prods:
                    H, T = b[k + "h1"].C, b[k + "h1"].Tst
                    nf, W, S = self.chunks[i]
                    call("sehip_dmx_frames", 2, b[k + "dh2"].ptr, None, B, T, H, nf, W, S, b[k + "dh2f"].ptr, stream())
                    self._chain_dirty = True
                    self.wgrad(k + "lin")
                    self.gemm(k + "lin.dg")
                    for l in (1, 0):
                        woff = st.whh[(k, l)][1]
                        call("sehip_dmx_lstm_bwd", b[f"{k}pre{l}"].ptr, tb.wpack.data_ptr() + 2 * woff, b[f"{k}cs{l}"].ptr, b[f"{k}dhs{l}"].ptr, B * nf, W, H,
                             b[f"{k}dG{l}"].ptr, ptr(self.dc), self._sync_arg(), stream())
                        self._chain_dirty = True
                        self.wgrad(f"{k}ih{l}")
                        self.wgrad(f"{k}hh{l}.0")
                        self.wgrad(f"{k}hh{l}.1")
                        self.gemm(f"{k}ih{l}.dg")
                    call("sehip_dmx_frames", 3, b[k + "dfr"].ptr, b[k + "dh2"].ptr, B, T, H, nf, W, S, b[k + "dh1"].ptr, stream())
                    self._chain_dirty = True
                self._norm_bwd(k + "n1", params, k + "dh1", k + "dy1")
                self.wgrad(k + "c1")
                self.gemm(k + "c1.dg")
            self._norm_bwd(e + "n0", params, e + "da", e + "dy")
            self.wgrad(e + "conv")
            if i > 0:
                self.gemm(e + "conv.dg")
            if range_ready is not None and i > 0 and enc_off[i + 1] - enc_off[i] >= 0.05 * n_params and done_from == enc_off[i + 1]:
                self._hand_over(enc_off[i], done_from, grads, range_ready)
                done_from = enc_off[i]
        if self.side is not None and not torch.cuda.is_current_stream_capturing():
            call("sehip_stream_depend", stream(), self.side.cuda_stream, self._event())
        self._unpack(0, done_from, grads, stream(), tail=tail if range_ready is None else None)
        if range_ready is not None:
            range_ready(0, done_from, torch.cuda.current_stream())
            call("sehip_stream_depend", stream(), self.comm.cuda_stream, self._event())
            ends = sorted(handed)
            if ends[0][0] != 0 or ends[-1][1] != n_params or any(a[1] != b_[0] for a, b_ in zip(ends[:-1], ends[1:])):
                raise SehipError(f"Demucs data-parallel hand-over: the ranges {ends} do not tile [0, {n_params})")
        return grads
