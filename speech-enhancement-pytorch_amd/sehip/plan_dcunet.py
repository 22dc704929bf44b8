"""Host-side plan of the complex DCUnet train step on libsehip (reference: src/model/dcunet.py:53-386): parameter layout,
packing tables, implicit-GEMM descriptors and the launch sequences of forward and backward.

The reference works on ``[B, C, T, F, 2]`` (complex as a trailing dim) with four real convolutions per complex one; here every
activation is channels-last bf16 ``[B][T][F][2*Cs]`` (real half | imaginary half, Cs = channels stored per half: 31 / 62
complex channels are stored as 32 / 64, the padding channels stay exact zeros) and a complex (transposed) convolution is ONE
real product over K = taps x input channels with the packed block weight [[Wre, -Wim], [Wim, Wre]] on the table-driven
implicit-GEMM engine of csrc/gemm.hip (the descriptor's frame stride `tmul` covers the (2, 2)-strided layers):

  encoder conv        rows = output positions; source frame t*st + kt - pt, source row f*sf + kf - pf
  its dgrad           one product per PARITY CLASS of the input position (the taps a stride-2 transposed read can reach)
  decoder deconv      one product per parity class of the OUTPUT position, two sources (decoder output | skip connection,
                      the reference's torch.cat along channels, src/model/dcunet.py:126)
  its dgrad           a strided convolution over dOut with two destinations (d decoder input, d skip)
  weight gradients    the same descriptors with dOut as the second operand (sehip_wgrad)
BatchNorm (two real ones) + LeakyReLU: csrc/rbn.hip.  Input transpose, 1x1 conv + tanh + mask: csrc/dcunet.hip.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream, SehipError
from .plan import (Arena, CGemmDesc, GemmSpec, ParamLayout, bind_chunk_table, dense_ntab, enc_entry, npad_of, round_up, BF16)


def stored_channels(c):
    s = 8
    while s < c:
        s *= 2
    return s


def dcunet_tables(model_complexity, model_depth, audio_channels):
    """Channel / kernel / stride / padding tables of the reference (src/model/dcunet.py:165-307), (time, frequency) pairs."""
    mc = model_complexity
    same = lambda ks, ps: [tuple((k - 1) // 2 for k in kk) if p is None else p for kk, p in zip(ks, ps)]   # 'SAME' (:12-13)
    if model_depth == 10:
        enc_ch = [audio_channels, mc, mc * 2, mc * 2, mc * 2, mc * 2]
        enc_k = [(7, 5), (7, 5), (5, 3), (5, 3), (5, 3)]
        enc_s = [(2, 2), (2, 2), (2, 2), (2, 2), (2, 1)]
        enc_p = same(enc_k, [(2, 1), None, None, None, None])
        dec_ch = [0, mc * 2, mc * 2, mc * 2, mc * 2, mc * 2]
        dec_k = [(4, 3), (4, 4), (6, 4), (6, 4), (7, 5)]
        dec_s = [(2, 1), (2, 2), (2, 2), (2, 2), (2, 2)]
        dec_p = [(1, 1), (1, 1), (2, 1), (2, 1), (2, 1)]
    elif model_depth == 20:      # src/model/dcunet.py:215-305
        enc_ch = [audio_channels, mc, mc] + [mc * 2] * 7 + [128]
        enc_k = [(7, 1), (1, 7), (6, 4), (7, 5)] + [(5, 3)] * 6
        enc_s = [(1, 1), (1, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1)]
        enc_p = same(enc_k, [(3, 0), (0, 3)] + [None] * 8)
        dec_ch = [0] + [mc * 2] * 11
        dec_k = [(4, 3), (4, 2), (4, 3), (4, 2), (4, 3), (4, 2), (6, 3), (7, 5), (1, 7), (7, 1)]
        dec_s = [(2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (1, 1), (1, 1)]
        dec_p = [(1, 1), (1, 0), (1, 1), (1, 0), (1, 1), (1, 0), (2, 1), (2, 1), (0, 3), (3, 0)]
    else:
        raise SehipError(f"Unknown model depth : {model_depth}")
    return dict(n=model_depth // 2, enc_ch=enc_ch, enc_k=enc_k, enc_s=enc_s, enc_p=enc_p, dec_ch=dec_ch, dec_k=dec_k, dec_s=dec_s,
                dec_p=dec_p)


class DCUNetConfig:
    """Constructor arguments of the reference model (src/model/dcunet.py:54-66)."""

    def __init__(self, audio_channels=1, data_type=False, model_complexity=45, model_depth=20, padding_mode="zeros",
                 masking_mode="E", **_ignored):
        if not data_type:
            raise SehipError("sehip DCUnet: only the complex network (data_type=True, the shipped configuration) is built")
        if padding_mode != "zeros":
            raise SehipError("sehip DCUnet: only padding_mode='zeros' is built")
        if audio_channels != 1:
            raise SehipError("sehip DCUnet: audio_channels must be 1 (the Solver folds channels into the batch, src/solver.py:450-452)")
        if masking_mode not in ("E", "C", "R"):
            raise SehipError(f"unknown masking_mode {masking_mode}")
        self.audio_channels, self.model_depth, self.masking_mode = audio_channels, model_depth, masking_mode
        self.model_complexity = int(model_complexity // 1.414)          # src/model/dcunet.py:64-65
        self.tab = dcunet_tables(self.model_complexity, model_depth, audio_channels)
        if stored_channels(max(self.tab["enc_ch"][1:] + self.tab["dec_ch"][1:])) > 128:
            raise SehipError(f"sehip DCUnet: model_complexity {model_complexity} gives {2 * self.model_complexity} complex channels; "
                             "up to 128 are built")

    def key(self):
        return (self.model_complexity, self.model_depth, self.masking_mode)

    def param_specs(self):
        """[(name, shape, kind)] in the reference's parameters() order: encoder0..4, decoder0..4, linear."""
        t = self.tab
        out = []

        def conv(pre, parts, wshape, nb):
            for part in parts:
                out.append((f"{pre}{part}.weight", wshape, "param"))
                out.append((f"{pre}{part}.bias", (nb,), "param"))

        def bn(pre, n):
            for part in ("bn_re", "bn_im"):
                out.append((f"{pre}{part}.weight", (n,), "param"))
                out.append((f"{pre}{part}.bias", (n,), "param"))
                out.append((f"{pre}{part}.running_mean", (n,), "buffer"))
                out.append((f"{pre}{part}.running_var", (n,), "buffer"))
                out.append((f"{pre}{part}.num_batches_tracked", (), "nbt"))

        for i in range(t["n"]):
            ci, co = t["enc_ch"][i], t["enc_ch"][i + 1]
            conv(f"encoder{i}.conv.", ("conv_re", "conv_im"), (co, ci) + t["enc_k"][i], co)
            bn(f"encoder{i}.bn.", co)
        for j in range(t["n"]):
            ci, co = t["dec_ch"][j] + t["enc_ch"][t["n"] - j], t["dec_ch"][j + 1]
            conv(f"decoder{j}.transconv.", ("tconv_re", "tconv_im"), (ci, co) + t["dec_k"][j], co)
            bn(f"decoder{j}.bn.", co)
        conv("linear.", ("conv_re", "conv_im"), (1, t["dec_ch"][-1], 1, 1), 1)
        return out


def padded_index(idx, n_out, n_in, axis_out, axis_in, so, si):
    """Parameter-index array of a conv weight enlarged to the STORED channel counts (-1 on the padding channels)."""
    shape = list(idx.shape)
    shape[axis_out], shape[axis_in] = so, si
    full = np.full(shape, -1, dtype=np.int64)
    sl = [slice(None)] * idx.ndim
    sl[axis_out], sl[axis_in] = slice(0, n_out), slice(0, n_in)
    full[tuple(sl)] = idx
    return full


def complex_block_padded(wr, wi, transposed):
    """Real block matrix of a complex (transposed) convolution over padded channel halves; entries -1 stay -1.

    conv   w [Co, Ci, kt, kf] -> [2Co, 2Ci, kt, kf]:  real = Wre xr - Wim xi, imag = Wim xr + Wre xi   (src/model/dcunet.py:334-337)
    deconv w [Ci, Co, kt, kf] -> [2Ci, 2Co, kt, kf]:  the same with (in, out) swapped                  (:366-370)
    """
    a, b = wr.shape[0], wr.shape[1]
    idx = np.full((2 * a, 2 * b) + wr.shape[2:], -1, dtype=np.int64)
    neg = np.zeros_like(idx)
    idx[:a, :b], idx[a:, b:] = wr, wr
    idx[:a, b:], idx[a:, :b] = wi, wi
    if not transposed:
        neg[:a, b:] = 1          # out real <- in imag: -Wim
    else:
        neg[a:, :b] = 1          # in imag -> out real: -Wim
    return idx, neg


def parity_taps(k, p, parity):
    """Taps kk of a stride-2 transposed read that reach positions of this parity, with their source offset (pos//2 + off)."""
    return [(kk, (parity + p - kk) // 2) for kk in range(k) if (parity + p - kk) % 2 == 0]


class DCUNetStatic:
    """Everything that does not depend on the input size: parameter layout and the packed-weight layout of every product."""

    def __init__(self, cfg: DCUNetConfig):
        self.cfg = cfg
        self.layout = ParamLayout(cfg)
        t = cfg.tab
        self.n = t["n"]
        self.enc_c = [stored_channels(c) for c in t["enc_ch"][1:]]       # stored complex channels of encoder outputs
        self.dec_c = [stored_channels(c) for c in t["dec_ch"][1:]]
        self.enc_cr = t["enc_ch"][1:]
        self.dec_cr = t["dec_ch"][1:]


def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def deconv_out(n, k, s, p):
    return (n - 1) * s - 2 * p + k


class DCUNetPlan:
    """Products of one input geometry (F0 bins x T0 frames): chunk tables, packed-weight tables, row spaces."""

    def __init__(self, st: DCUNetStatic, F0, T0):
        self.st, self.F0, self.T0 = st, F0, T0
        cfg, t, L = st.cfg, st.cfg.tab, st.layout
        ia = L.index_array
        n = st.n
        # ---- geometry
        self.enc_dims = []
        T, F = T0, F0
        for i in range(n):
            (kt, kf), (s_t, s_f), (p_t, p_f) = t["enc_k"][i], t["enc_s"][i], t["enc_p"][i]
            T, F = conv_out(T, kt, s_t, p_t), conv_out(F, kf, s_f, p_f)
            if T < 1 or F < 1:
                raise SehipError(f"DCUnet: input of {F0} bins x {T0} frames is too small for encoder {i}")
            self.enc_dims.append((T, F))
        self.dec_dims = []
        for j in range(n):
            (kt, kf), (s_t, s_f), (p_t, p_f) = t["dec_k"][j], t["dec_s"][j], t["dec_p"][j]
            T, F = deconv_out(T, kt, s_t, p_t), deconv_out(F, kf, s_f, p_f)
            self.dec_dims.append((T, F))
            want = self.enc_dims[n - 2 - j] if j < n - 1 else (T0, F0)
            if (T, F) != want:
                raise SehipError(f"DCUnet: decoder {j} produces {T} x {F} (frames x bins) but its skip connection / the input is "
                                 f"{want[0]} x {want[1]}: the depth-10 network needs 257 bins and frames = 1 mod 32, depth 20 frames = 1 mod 16 "
                                 f"(got {F0} bins x {T0} frames)")
        self.specs = {}
        self.bn = []       # (prefix, buffer tag, Cs, Cr)
        self.bias_group = {}

        def eff_bias(pre_re, pre_im, cs, cr):
            """real rows get b_re - b_im, imaginary rows b_re + b_im (each of the four real convs adds its own bias)."""
            bre, bim = ia(pre_re + ".bias"), ia(pre_im + ".bias")
            pairs = np.full((2 * cs, 2), -1, dtype=np.int32)
            pairs[:cr, 0] = enc_entry(bre, 0); pairs[:cr, 1] = enc_entry(bim, 1)
            pairs[cs:cs + cr, 0] = enc_entry(bre, 0); pairs[cs:cs + cr, 1] = enc_entry(bim, 0)
            return pairs

        def wide(src, toff, fadd, c):
            return [(src, toff, fadd, 8 * q) for q in range(c // 8)]

        # ---------------- encoders ----------------
        self.enc_wg = {}       # encoder -> the products whose weight-gradient twins make up its weight gradient
        for i in range(n):
            (kt, kf), (s_t, s_f), (p_t, p_f) = t["enc_k"][i], t["enc_s"][i], t["enc_p"][i]
            cin_r, cout_r = t["enc_ch"][i], t["enc_ch"][i + 1]
            cout_s = st.enc_c[i]
            cin_s = 1 if i == 0 else st.enc_c[i - 1]
            pre = f"encoder{i}.conv."
            wr = padded_index(ia(pre + "conv_re.weight"), cout_r, cin_r, 0, 1, cout_s, cin_s)
            wi = padded_index(ia(pre + "conv_im.weight"), cout_r, cin_r, 0, 1, cout_s, cin_s)
            full, neg = complex_block_padded(wr, wi, False)          # [2co, 2ci, kt, kf]
            Tout, Fout = self.enc_dims[i]
            src = "x0" if i == 0 else f"ze{i - 1}"
            if i == 0:
                # narrow source (2 channels): a chunk = 4 consecutive bins x (re, im)
                rows, cols_i, cols_n = [], [], []
                for a in range(kt):
                    for b0 in range(0, kf, 4):
                        nv = min(4, kf - b0)
                        rows.append((0, a - p_t, b0 - p_f, nv))
                        blk_i = np.full((2 * cout_s, 8), -1, np.int64); blk_n = np.zeros((2 * cout_s, 8), np.int64)
                        for q in range(nv):
                            for c in range(2):
                                blk_i[:, q * 2 + c] = full[:, c, a, b0 + q]
                                blk_n[:, q * 2 + c] = neg[:, c, a, b0 + q]
                        cols_i.append(blk_i); cols_n.append(blk_n)
                widx, wneg = np.concatenate(cols_i, 1), np.concatenate(cols_n, 1)
            else:
                rows = []
                for a in range(kt):
                    for b in range(kf):
                        rows += wide(0, a - p_t, b - p_f, 2 * cin_s)
                widx = full.transpose(0, 2, 3, 1).reshape(2 * cout_s, -1)   # k order (kt, kf, channel)
                wneg = neg.transpose(0, 2, 3, 1).reshape(2 * cout_s, -1)
            sp = GemmSpec(f"enc{i}.fwd", rows, widx, wneg, 2 * cout_s, eff_bias(pre + "conv_re", pre + "conv_im", cout_s, cout_r),
                          Tout, Fout, s_f, [(src, "all")], [(f"ye{i}", 0, 1, 0)])
            sp.tmul, sp.dst_tmul = s_t, [1]
            self.specs[sp.name] = sp
            self.enc_wg[i] = [sp.name]
            # Weight gradient by the parity class of the TAP: the taps a = pa (mod s_t), b = pb (mod s_f) read the sub-lattice
            # (s_t t + pa - p_t + s_t a', s_f j + pb - p_f + s_f b') of the input, on which they are a stride-1 convolution of
            # ceil-half the taps -- the shape the patch-based weight gradient stages once per tile (csrc/gemm.hip conv_wgrad2_kernel:
            # cv2_* with the descriptor's tmul / fmul as the lattice step).  As ONE table-gathered product the encoders' weight
            # gradients ran at 0.085 of the MFMA peak (enc1: 660 us for 141 GF).  Weight-gradient-only products (kind "wg").
            # (only where the row space is long enough: four launches of the deeper layers' 16 k / 8 k rows measured slower than their one
            #  generic product -- enc3 103 vs 75 us, enc4 81 vs 45 us)
            if (i >= 1 and (2 * cin_s) % 64 == 0 and Tout * Fout >= int(os.environ.get("SEHIP_DCUNET_ENC_WG_MIN", "1024"))
                    and not os.environ.get("SEHIP_DCUNET_NO_ENC_WG_CLASSES")):
                cls = []
                for pa in range(s_t):
                    for pb in range(s_f):
                        at = [a for a in range(kt) if a % s_t == pa]
                        bf_ = [b for b in range(kf) if b % s_f == pb]
                        if at and bf_:
                            cls.append((pa, pb, at, bf_))
                supported = {(4, 3), (3, 3), (4, 2), (3, 2), (2, 2), (2, 3), (3, 1), (2, 1)}
                if all((len(at), len(bf_)) in supported for _, _, at, bf_ in cls):
                    names = []
                    for pa, pb, at, bf_ in cls:
                        rows = []
                        for a in at:
                            for b in bf_:
                                rows += wide(0, a - p_t, b - p_f, 2 * cin_s)
                        wi_ = np.concatenate([full[:, :, a, b] for a in at for b in bf_], 1)      # [2co, taps x 2ci]: k order (a', b', channel)
                        wn_ = np.concatenate([neg[:, :, a, b] for a in at for b in bf_], 1)
                        name = f"enc{i}.wg{pa}{pb}"
                        c = GemmSpec(name, rows, wi_, wn_, 2 * cout_s,
                                     eff_bias(pre + "conv_re", pre + "conv_im", cout_s, cout_r) if not names else None,
                                     Tout, Fout, s_f, [(src, "all")], [(f"ye{i}", 0, 1, 0)], kind="wg")
                        c.tmul, c.dst_tmul = s_t, [1]
                        c.conv2 = (len(at), len(bf_), bf_[0] - p_f, at[0] - p_t)      # (nkt, nf, fadd, t0); tap step = (tmul, fmul)
                        self.specs[name] = c
                        names.append(name)
                    self.enc_wg[i] = names
                    sp.no_wgrad = True
            self.bn.append((f"encoder{i}.bn.", f"e{i}", cout_s, cout_r))
            if i >= 1:
                # dgrad by the parity class of the input position
                Tin, Fin = self.enc_dims[i - 1]
                tcls = [(pt, parity_taps(kt, p_t, pt)) for pt in range(s_t)] if s_t == 2 else [(0, [(a, p_t - a) for a in range(kt)])]
                fcls = [(pf, parity_taps(kf, p_f, pf)) for pf in range(s_f)] if s_f == 2 else [(0, [(b, p_f - b) for b in range(kf)])]
                for pt, ttaps in tcls:
                    for pf, ftaps in fcls:
                        rows, cols_i, cols_n = [], [], []
                        for a, ta in ttaps:
                            for b, fb in ftaps:
                                rows += wide(0, ta, fb, 2 * cout_s)
                                cols_i.append(full[:, :, a, b].T); cols_n.append(neg[:, :, a, b].T)   # [2ci, 2co]
                        if not rows:
                            continue
                        name = f"enc{i}.dg{pt}{pf}"
                        TT = (Tin - pt + s_t - 1) // s_t
                        J = (Fin - pf + s_f - 1) // s_f
                        sp = GemmSpec(name, rows, np.concatenate(cols_i, 1), np.concatenate(cols_n, 1), 2 * cin_s, None, TT, J, 1,
                                      [(f"dye{i}", "all")], [(f"dze{i - 1}", pt, s_f, pf)], kind="dgrad", res=f"dskip{i - 1}")
                        sp.tmul, sp.dst_tmul = 1, [s_t]
                        self.specs[name] = sp

        # ---------------- decoders ----------------
        for j in range(n):
            (kt, kf), (s_t, s_f), (p_t, p_f) = t["dec_k"][j], t["dec_s"][j], t["dec_p"][j]
            c1_r = t["dec_ch"][j]
            c2_r = t["enc_ch"][n - j]
            cout_r, cout_s = t["dec_ch"][j + 1], st.dec_c[j]
            pre = f"decoder{j}.transconv."
            if j == 0:
                srcs = [(f"ze{n - 1}", c2_r, st.enc_c[n - 1], 0)]                  # (buffer, real channels, stored, first cat channel)
            else:
                srcs = [(f"zd{j - 1}", c1_r, st.dec_c[j - 1], 0), (f"ze{n - 1 - j}", c2_r, st.enc_c[n - 1 - j], c1_r)]
            wr_all, wi_all = ia(pre + "tconv_re.weight"), ia(pre + "tconv_im.weight")      # [c1+c2, cout, kt, kf]
            blocks = []
            for (bname, cr, cs, c0) in srcs:
                wr = padded_index(wr_all[c0:c0 + cr], cr, cout_r, 0, 1, cs, cout_s)
                wi = padded_index(wi_all[c0:c0 + cr], cr, cout_r, 0, 1, cs, cout_s)
                blocks.append(complex_block_padded(wr, wi, True))      # [2cs, 2cout, kt, kf]
            Tin, Fin = self.enc_dims[n - 1] if j == 0 else self.dec_dims[j - 1]
            Tout, Fout = self.dec_dims[j]
            tcls = [(pt, parity_taps(kt, p_t, pt)) for pt in range(s_t)] if s_t == 2 else [(0, [(a, p_t - a) for a in range(kt)])]
            fcls = [(pf, parity_taps(kf, p_f, pf)) for pf in range(s_f)] if s_f == 2 else [(0, [(b, p_f - b) for b in range(kf)])]
            group = []
            for pt, ttaps in tcls:
                for pf, ftaps in fcls:
                    # taps in ascending source offset: K is then ordered (time tap, row tap, source, channel) with consecutive
                    # offsets, the regular stride-1 convolution the patch-based weight gradient (cv2_* of sehip_gemm_desc) takes
                    ttaps = sorted(ttaps, key=lambda x: x[1])
                    ftaps = sorted(ftaps, key=lambda x: x[1])
                    rows, cols_i, cols_n = [], [], []
                    for a, ta in ttaps:
                        for b, fb in ftaps:
                            for q, (bname, cr, cs, c0) in enumerate(srcs):
                                rows += wide(q, ta, fb, 2 * cs)
                                cols_i.append(blocks[q][0][:, :, a, b].T); cols_n.append(blocks[q][1][:, :, a, b].T)   # [2cout, 2cs]
                    if not rows:
                        continue
                    name = f"dec{j}.fwd{pt}{pf}"
                    TT = (Tout - pt + s_t - 1) // s_t
                    J = (Fout - pf + s_f - 1) // s_f
                    sp = GemmSpec(name, rows, np.concatenate(cols_i, 1), np.concatenate(cols_n, 1), 2 * cout_s,
                                  eff_bias(pre + "tconv_re", pre + "tconv_im", cout_s, cout_r), TT, J, 1,
                                  [(s[0], "all") for s in srcs], [(f"yd{j}", pt, s_f, pf)])
                    sp.tmul, sp.dst_tmul = 1, [s_t]
                    toffs, foffs = [x[1] for x in ttaps], [x[1] for x in ftaps]
                    if toffs == list(range(toffs[0], toffs[0] + len(toffs))) and foffs == list(range(foffs[0], foffs[0] + len(foffs))):
                        sp.conv2 = (len(toffs), len(foffs), foffs[0], toffs[0])      # (nkt, nf, fadd, t0)
                    self.specs[name] = sp
                    group.append(name)
            self.bias_group[f"dec{j}"] = group
            self.bn.append((f"decoder{j}.bn.", f"d{j}", cout_s, cout_r))
            # dgrad: a strided convolution over dOut, destinations = d(decoder input) [and d(skip)]
            rows, cols_i, cols_n = [], [], []
            for a in range(kt):
                for b in range(kf):
                    rows += wide(0, a - p_t, b - p_f, 2 * cout_s)
                    cols_i.append(np.concatenate([blk[0][:, :, a, b] for blk in blocks], 0))    # [sum 2cs, 2cout]
                    cols_n.append(np.concatenate([blk[1][:, :, a, b] for blk in blocks], 0))
            ntot = sum(2 * s[2] for s in srcs)
            nt = np.concatenate([dense_ntab(2 * s[2], 2 * s[2], q, 0) for q, s in enumerate(srcs)])
            dsts = [(f"dze{n - 1}", 0, 1, 0)] if j == 0 else [(f"dzd{j - 1}", 0, 1, 0), (f"dskip{n - 1 - j}", 0, 1, 0)]
            # 128 + 64 columns (the last decoder of DCUnet-10: decoder input | skip): one 192-wide tile instead of two 128-wide ones
            npad = 192 if ntot == 192 and "SEHIP_DCUNET_NO_N192" not in os.environ else npad_of(ntot)
            sp = GemmSpec(f"dec{j}.dg", rows, np.concatenate(cols_i, 1), np.concatenate(cols_n, 1), ntot, None, Tin, Fin, s_f,
                          [(f"dyd{j}", "all")], dsts, ntab=self._pad_ntab(nt, npad), kind="dgrad", npad=npad)
            sp.tmul, sp.dst_tmul = s_t, [1] * len(dsts)
            self.specs[sp.name] = sp

        # ---------------- arenas ----------------
        wa, ba, ga = Arena(64), Arena(16), Arena(16)
        kta, nta = Arena(1), Arena(1)
        shared_db = {}
        for name, s in self.specs.items():
            s.kt_off = kta.add(s.ktab)
            s.nt_off = nta.add(s.ntab)
            if s.kind != "wg":         # (weight-gradient-only products read no packed weights / bias)
                s.w_off = wa.add(enc_entry(s.widx, s.wneg).reshape(-1))
                if s.bias_pairs is not None:
                    s.b_off = ba.add(s.bias_pairs)
            if (s.kind == "fwd" and not getattr(s, "no_wgrad", False)) or s.kind == "wg":
                s.dw_off = ga.reserve(s.Npad * s.K)
                if s.bias_pairs is not None:
                    # the parity classes of one transposed convolution cover disjoint output positions: their bias sums
                    # accumulate (atomics) into ONE region, so a bias parameter still folds from two entries
                    layer = name.split(".")[0]
                    if layer not in shared_db:
                        shared_db[layer] = ga.reserve(s.Npad)
                    s.db_off = shared_db[layer]
        self.bn_g_off = {}
        for pre, tag, cs, cr in self.bn:
            self.bn_g_off[pre] = {k: ga.reserve(cr) for k in ("w_re", "b_re", "w_im", "b_im")}
        cs_l = st.dec_c[-1]
        self.lin_g_off = ga.reserve(2 * cs_l + 2)
        self.n_wpack, self.n_bpack, self.n_gpack = wa.size, ba.size, ga.size
        self.wtab = wa.build(np.int32)
        self.btab = ba.build(np.int32, 2)
        self.ktab = kta.build(np.int32, 4)
        self.ntab = nta.build(np.int32, 4, fill=0)
        self.utab = self._build_unpack_table()

    @staticmethod
    def _pad_ntab(nt, npad):
        if npad // 4 == nt.shape[0]:
            return nt
        out = np.zeros((npad // 4, 4), dtype=np.int32)
        out[:nt.shape[0]] = nt
        return out

    def _build_unpack_table(self):
        L, st = self.st.layout, self.st
        ps, gs, ns = [], [], []
        done_bias = set()
        for s in self.specs.values():
            if s.dw_off is None:
                continue
            m = s.widx >= 0
            ps.append(s.widx[m]); ns.append(s.wneg[m])
            gs.append(s.dw_off + np.flatnonzero(m.reshape(-1)))
            if s.db_off is not None and s.db_off not in done_bias:
                done_bias.add(s.db_off)
                for col in (0, 1):
                    e = s.bias_pairs[:, col].astype(np.int64)
                    mm = e >= 0
                    ps.append(e[mm] >> 1); ns.append(e[mm] & 1)
                    gs.append(s.db_off + np.flatnonzero(mm))
        for pre, tag, cs, cr in self.bn:
            for k, leaf in (("w_re", "bn_re.weight"), ("b_re", "bn_re.bias"), ("w_im", "bn_im.weight"), ("b_im", "bn_im.bias")):
                ps.append(L.index_array(pre + leaf)); ns.append(np.zeros(cr, np.int64))
                gs.append(self.bn_g_off[pre][k] + np.arange(cr))
        cr_l, cs_l = st.dec_cr[-1], st.dec_c[-1]
        for k, (leaf, base) in enumerate((("linear.conv_re.weight", 0), ("linear.conv_im.weight", cs_l))):
            ps.append(L.index_array(leaf).reshape(-1)); ns.append(np.zeros(cr_l, np.int64))
            gs.append(self.lin_g_off + base + np.arange(cr_l))
        for leaf, pos in (("linear.conv_re.bias", 2 * cs_l), ("linear.conv_im.bias", 2 * cs_l + 1)):
            ps.append(L.index_array(leaf).reshape(-1)); ns.append(np.zeros(1, np.int64))
            gs.append(np.asarray([self.lin_g_off + pos]))
        p = np.concatenate([a.reshape(-1) for a in ps]).astype(np.int64)
        g = np.concatenate([a.reshape(-1) for a in gs]).astype(np.int64)
        nn_ = np.concatenate([a.reshape(-1) for a in ns]).astype(np.int64)
        order = np.argsort(p, kind="stable")
        p, g, nn_ = p[order], g[order], nn_[order]
        first = np.searchsorted(p, p, side="left")
        slot = np.arange(p.shape[0]) - first
        assert slot.max() < 4, "a parameter feeds more than 4 packed-gradient entries"
        tab = np.full((L.n_params, 4), -1, dtype=np.int32)
        tab[p, slot] = ((g << 1) | nn_).astype(np.int32)
        return tab


class Buf:
    def __init__(self, t, T, F, Cc):
        self.t, self.Tst, self.F, self.C = t, T, F, Cc

    @property
    def ptr(self):
        return self.t.data_ptr()


class DCUNetDeviceTables:
    def __init__(self, pl: DCUNetPlan, device):
        f = lambda a: torch.from_numpy(a).to(device)
        self.wtab, self.btab, self.utab, self.ntab = f(pl.wtab), f(pl.btab), f(pl.utab), f(pl.ntab)
        self.tensor_offsets = f(pl.st.layout.tensor_offsets)
        self.utab_g = self.uperm = None                 # the fused tail's un-pack in gather order (plan.gather_ordered_unpack_table)
        if not os.environ.get("SEHIP_NO_UNPACK_PERM"):
            from .plan import gather_ordered_unpack_table
            tg, pm = gather_ordered_unpack_table(pl.utab, pl.st.layout.tensor_offsets)
            self.utab_g, self.uperm = f(tg), f(pm)
        self.wpack = torch.zeros(pl.n_wpack, dtype=BF16, device=device)
        self.bpack = torch.zeros(max(pl.n_bpack, 4), dtype=torch.float32, device=device)


class DCUNetWorkspace:
    """Device buffers and bound descriptors for one (batch, bins, frames)."""

    def __init__(self, pl: DCUNetPlan, tables: DCUNetDeviceTables, B, device):
        self.pl, self.tb, self.B, self.device = pl, tables, B, device
        self.generation, self.pinned, self.closed = 0, False, False
        st, n = pl.st, pl.st.n
        F0, T0 = pl.F0, pl.T0
        self.bufs = {}

        def add(name, T, F, Cc):
            self.bufs[name] = Buf(torch.zeros(B, T, F, Cc, dtype=BF16, device=device), T, F, Cc)

        add("x0", T0, F0, 2)
        for i in range(n):
            T, F = pl.enc_dims[i]
            for pre in ("ye", "ze", "dye", "dze"):
                add(f"{pre}{i}", T, F, 2 * st.enc_c[i])
            if i < n - 1:
                add(f"dskip{i}", T, F, 2 * st.enc_c[i])
        for j in range(n):
            T, F = pl.dec_dims[j]
            for pre in ("yd", "zd", "dyd", "dzd"):
                add(f"{pre}{j}", T, F, 2 * st.dec_c[j])
        self.mask_ws = torch.empty(B, T0, F0, 2, dtype=torch.float32, device=device)
        self.out = torch.empty(B, 1, F0, T0, 2, dtype=torch.float32, device=device)
        self.spec = None          # the caller's input spectrum of the live forward (kept for the backward pass)
        self.gpack = torch.zeros(pl.n_gpack, dtype=torch.float32, device=device)
        lib = _lib.lib()
        need = 16
        for pre, tag, cs, cr in pl.bn:
            rows = self.bufs[("ye" if tag[0] == "e" else "yd") + tag[1:]].t.numel() // (2 * cs)
            need = max(need, int(lib.sehip_rbn_scratch_floats(rows, cs)))
        self.bn_acc = torch.zeros(need, dtype=torch.float32, device=device)
        self.bn_coef = {pre: torch.zeros(2 * cs, 4, dtype=torch.float32, device=device) for pre, tag, cs, cr in pl.bn}
        self.bn_bcoef = torch.zeros(2 * max(cs for _, _, cs, _ in pl.bn), 4, dtype=torch.float32, device=device)     # sehip_rbn_bwd_finalize: [2 cs][4]
        self.mode = {"E": 0, "C": 1, "R": 2}[st.cfg.masking_mode]
        # The fused tail (csrc/dcunet.hip): the last decoder's BatchNorm + LeakyReLU output zd{n-1} and its gradient dzd{n-1} -- 1.08 GB
        # each at B = 64 -- are never written; the mask kernel applies the BatchNorm to the pre-activation it loads, the backward pass
        # rebuilds d(output) from the two values per position the 1x1 convolution's transpose spreads over the channels.
        # SEHIP_DCUNET_NO_TAIL=1: the separate kernels (the op-local tests compare those two tensors with the oracle).
        self.fused_tail = "SEHIP_DCUNET_NO_TAIL" not in os.environ
        # (round 6: the BatchNorm backward finalize step inside the reduce launch's last workgroup -- sehip_rbn_bwd_reduce_fin, built because
        #  the 9 rbn_bwd_finalize launches of the C2 step take ~105 us each waiting for a CU beside the last decoder's whole-CU
        #  weight-gradient workgroups.  Correct, deterministic, and NO gain: 13.69 / 13.86 against 13.67 / 13.82 ms -- the wait moves to
        #  the next launch of the chain; what bounds the step is the work on the machine, not the launch that happens to be waiting.
        #  SEHIP_DCUNET_REDUCE_FIN=1)
        self.reduce_fin = "SEHIP_DCUNET_REDUCE_FIN" in os.environ
        self.bn_ticket = torch.zeros(4, dtype=torch.int32, device=device)
        self.tail_scratch = torch.zeros(int(lib.sehip_dcunet_tail_scratch_floats(B, F0, T0, st.dec_c[-1])), dtype=torch.float32,
                                        device=device)
        self.side = None if os.environ.get("SEHIP_NO_SIDE_STREAM") else torch.cuda.Stream(device=device)
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
        pl, tb, B = self.pl, self.tb, self.B
        self.desc = {}
        kt = pl.ktab.copy()
        for name, s in pl.specs.items():
            geo = [(self.bufs[b].F, self.bufs[b].C) for b, _ in s.srcs]
            bind_chunk_table(pl.ktab, kt, s.kt_off, s.K // 8, geo)
        self.ktab_dev = torch.from_numpy(kt).to(self.device)
        for name, s in pl.specs.items():
            d = CGemmDesc()
            for q, (bname, mode) in enumerate(s.srcs):
                b = self.bufs[bname]
                d.src[q].ptr = b.ptr
                d.src[q].T, d.src[q].F, d.src[q].C = b.Tst, b.F, b.C
                d.src[q].tlo, d.src[q].thi = 0, b.Tst
            for q, (bname, toff, fmul, fadd) in enumerate(s.dsts):
                b = self.bufs[bname]
                d.dst[q].ptr = b.ptr
                d.dst[q].T, d.dst[q].F, d.dst[q].C = b.Tst, b.F, b.C
                d.dst[q].toff, d.dst[q].fmul, d.dst[q].fadd = toff, fmul, fadd
                d.dst[q].is_f32 = 0
                d.dst[q].tmul = s.dst_tmul[q]
            d.ktab = self.ktab_dev.data_ptr() + 16 * s.kt_off
            d.ntab = tb.ntab.data_ptr() + 16 * s.nt_off
            d.W = tb.wpack.data_ptr() + 2 * s.w_off if s.w_off is not None else None
            # Every convolution here feeds a BatchNorm, which cancels its bias exactly (src/model/dcunet.py:323-338 adds it, :374-386
            # removes it): the products store their output WITHOUT the bias (d.bias stays NULL) and the BatchNorm finalize takes
            # the bias as `shift` (sehip_rbn_finalize_s) for the running mean / the inference mean.  With small-amplitude spectra the
            # bias is ~50x the signal and would otherwise cost the bf16 tensor most of its mantissa (17 % output error measured at
            # 0.1-scale white noise, 1 % without the bias).  (The weight-gradient twin below still sums dOut into the bias gradient.)
            d.M, d.N, d.Npad, d.K = B * s.tt * s.J, s.N, s.Npad, s.K
            d.TT, d.J, d.fmul, d.tmul = s.tt, s.J, s.fmul, s.tmul
            if s.res is not None:
                rb, db_ = self.bufs[s.res], self.bufs[s.dsts[0][0]]
                assert (rb.Tst, rb.F, rb.C) == (db_.Tst, db_.F, db_.C)
                d.res = rb.ptr
            if s.kind != "wg":         # (weight-gradient-only products have no forward launch)
                self.desc[name] = d
            if s.dw_off is not None:   # weight-gradient twin: dOut replaces the destination
                w = CGemmDesc.from_buffer_copy(d)
                w.dW = self.gpack.data_ptr() + 4 * s.dw_off
                w.dbias = self.gpack.data_ptr() + 4 * s.db_off if s.db_off is not None else None
                out_name = s.dsts[0][0]
                w.dst[0].ptr = self.bufs["d" + out_name].ptr
                if getattr(s, "conv2", None) is not None:
                    w.cv2_nkt, w.cv2_nf, w.cv2_fadd, w.cv2_t0 = s.conv2
                self.desc[name + ".wg"] = w

    # ---- launches ------------------------------------------------------------------------------------
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
        """Weight gradients are side work (nothing in the backward chain consumes them): second stream, see plan.DCCRNWorkspace."""
        main = torch.cuda.current_stream()
        if self.side is None or torch.cuda.is_current_stream_capturing():
            call("sehip_wgrad", C.byref(self.desc[name + ".wg"]), main.cuda_stream)
            return
        if self._chain_dirty:
            call("sehip_stream_depend", self.side.cuda_stream, main.cuda_stream, self._event())
            self._chain_dirty = False
        call("sehip_wgrad", C.byref(self.desc[name + ".wg"]), self.side.cuda_stream)

    def _bn_ptrs(self, pre, params, buffers, nbt):
        L = self.pl.st.layout
        pp = lambda k: params.data_ptr() + 4 * L.param_off[pre + k][0]
        bp = lambda k: buffers.data_ptr() + 4 * L.buffer_off[pre + k][0]
        nb = lambda k: nbt.data_ptr() + 8 * L.nbt_idx[pre + k + ".num_batches_tracked"]
        return pp, bp, nb

    def bn_forward(self, pre, cs, cr, y, z, params, buffers, nbt, training, apply=True):
        rows = y.t.numel() // (2 * cs)
        pp, bp, nb = self._bn_ptrs(pre, params, buffers, nbt)
        coef = self.bn_coef[pre]
        if training:
            call("sehip_rbn_stats", y.ptr, rows, cs, cr, ptr(self.bn_acc), stream())
        layer = pre.split(".")[0]                                      # encoder{i} / decoder{j}
        prod = f"enc{layer[7:]}.fwd" if layer.startswith("encoder") else self.pl.bias_group[f"dec{layer[7:]}"][0]
        shift = self.tb.bpack.data_ptr() + 4 * self.pl.specs[prod].b_off   # the producing convolution's effective bias [2 * cs]
        call("sehip_rbn_finalize_s", ptr(self.bn_acc), pp("bn_re.weight"), pp("bn_re.bias"), pp("bn_im.weight"), pp("bn_im.bias"),
             bp("bn_re.running_mean"), bp("bn_re.running_var"), bp("bn_im.running_mean"), bp("bn_im.running_var"),
             nb("bn_re"), nb("bn_im"), rows, cs, cr, 1e-5, 0.1, 1 if training else 0, shift, ptr(coef), stream())
        if apply:
            call("sehip_rbn_apply", y.ptr, ptr(coef), rows, cs, cr, z.ptr, stream())

    def materialize_tail(self):
        """Debug / tests: write the last decoder's BatchNorm + LeakyReLU output (zd{n-1}) the fused tail does not store."""
        n, st = self.pl.st.n, self.pl.st
        y, z = self.bufs[f"yd{n - 1}"], self.bufs[f"zd{n - 1}"]
        cs, cr = st.dec_c[-1], st.dec_cr[-1]
        call("sehip_rbn_apply", y.ptr, ptr(self.bn_coef[f"decoder{n - 1}.bn."]), y.t.numel() // (2 * cs), cs, cr, z.ptr, stream())
        return z

    def bn_backward(self, pre, cs, cr, dz, y, dy):
        rows = y.t.numel() // (2 * cs)
        g = lambda k: self.gpack.data_ptr() + 4 * self.pl.bn_g_off[pre][k]
        coef = self.bn_coef[pre]
        self._chain_dirty = True
        if self.reduce_fin:      # the finalize step inside the reduce launch's last workgroup (csrc/rbn.hip: no launch that waits for a CU)
            call("sehip_rbn_bwd_reduce_fin", dz.ptr, y.ptr, ptr(coef), rows, cs, cr, ptr(self.bn_acc), ptr(self.bn_ticket), g("w_re"),
                 g("b_re"), g("w_im"), g("b_im"), ptr(self.bn_bcoef), stream())
        else:
            call("sehip_rbn_bwd_reduce", dz.ptr, y.ptr, ptr(coef), rows, cs, cr, ptr(self.bn_acc), stream())
            call("sehip_rbn_bwd_finalize", ptr(self.bn_acc), ptr(coef), rows, cs, cr, g("w_re"), g("b_re"), g("w_im"), g("b_im"),
                 ptr(self.bn_bcoef), stream())
        call("sehip_rbn_bwd_apply", dz.ptr, y.ptr, ptr(coef), ptr(self.bn_bcoef), rows, cs, cr, dy.ptr, stream())

    def _lin_ptrs(self, params):
        L = self.pl.st.layout
        pp = lambda k: params.data_ptr() + 4 * L.param_off[k][0]
        return pp("linear.conv_re.weight"), pp("linear.conv_im.weight"), pp("linear.conv_re.bias"), pp("linear.conv_im.bias")

    def forward(self, spec, params, buffers, nbt, training=True):
        """spec [B, 1, F0, T0, 2] fp32 on device (what stft_custom returns) -> self.out, same shape."""
        pl, tb, b, st = self.pl, self.tb, self.bufs, self.pl.st
        n, B = st.n, self.B
        self.spec = spec
        call("sehip_pack_bf16", ptr(params), ptr(tb.wtab), pl.n_wpack, ptr(tb.wpack), stream())
        call("sehip_pack_f32", ptr(params), ptr(tb.btab), pl.n_bpack, ptr(tb.bpack), stream())
        call("sehip_dcunet_pack_input", ptr(spec), B, pl.F0, pl.T0, b["x0"].ptr, stream())
        for i in range(n):
            self.gemm(f"enc{i}.fwd")
            self.bn_forward(f"encoder{i}.bn.", st.enc_c[i], st.enc_cr[i], b[f"ye{i}"], b[f"ze{i}"], params, buffers, nbt, training)
        for j in range(n):
            for name in pl.bias_group[f"dec{j}"]:
                self.gemm(name)
            self.bn_forward(f"decoder{j}.bn.", st.dec_c[j], st.dec_cr[j], b[f"yd{j}"], b[f"zd{j}"], params, buffers, nbt, training,
                            apply=not (self.fused_tail and j == n - 1))
        wre, wim, bre, bim = self._lin_ptrs(params)
        if self.fused_tail:
            call("sehip_dcunet_mask_fwd_bn", b[f"yd{n - 1}"].ptr, ptr(self.bn_coef[f"decoder{n - 1}.bn."]), wre, wim, bre, bim, ptr(spec), B,
                 pl.F0, pl.T0, st.dec_c[-1], st.dec_cr[-1], self.mode, ptr(self.mask_ws), ptr(self.out), stream())
        else:
            call("sehip_dcunet_mask_fwd", b[f"zd{n - 1}"].ptr, wre, wim, bre, bim, ptr(spec), B, pl.F0, pl.T0, st.dec_c[-1], st.dec_cr[-1],
                 self.mode, ptr(self.mask_ws), ptr(self.out), stream())
        return self.out

    def backward(self, dout, params, grads, tail=None):
        """dout [B, 1, F0, T0, 2] fp32 -> flat parameter gradients (overwritten).  tail: FlatOptimizer's accumulators (plan.DCCRNWorkspace.backward)."""
        pl, tb, b, st = self.pl, self.tb, self.bufs, self.pl.st
        n, B = st.n, self.B
        self.gpack.zero_()
        self._chain_dirty = True
        wre, wim, _, _ = self._lin_ptrs(params)
        if self.fused_tail:
            # sehip_dcunet_tail_bwd overwrites mask_ws in place (tanh(linear) -> d linear): a second backward pass over the SAME forward
            # (retain_graph=True) would read d linear as the tanh output and return wrong gradients silently
            if getattr(self, "_tail_consumed", -1) == self.generation:
                raise SehipError("DCUnet.backward: a second backward pass over the same forward (retain_graph=True) is not possible with the "
                                 "fused tail (it consumes the forward's mask record); set SEHIP_DCUNET_NO_TAIL=1 for the separate kernels")
            self._tail_consumed = self.generation
            pre = f"decoder{n - 1}.bn."
            g = lambda k: self.gpack.data_ptr() + 4 * pl.bn_g_off[pre][k]
            call("sehip_dcunet_tail_bwd", ptr(dout), ptr(self.spec), ptr(self.mask_ws), b[f"yd{n - 1}"].ptr, ptr(self.bn_coef[pre]), wre, wim,
                 B, pl.F0, pl.T0, st.dec_c[-1], st.dec_cr[-1], self.mode, ptr(self.tail_scratch), g("w_re"), g("b_re"), g("w_im"), g("b_im"),
                 ptr(self.bn_bcoef), b[f"dyd{n - 1}"].ptr, self.gpack.data_ptr() + 4 * pl.lin_g_off, stream())
        else:
            call("sehip_dcunet_mask_bwd", ptr(dout), ptr(self.spec), ptr(self.mask_ws), b[f"zd{n - 1}"].ptr, wre, wim, B, pl.F0, pl.T0,
                 st.dec_c[-1], st.dec_cr[-1], self.mode, b[f"dzd{n - 1}"].ptr, self.gpack.data_ptr() + 4 * pl.lin_g_off, stream())
        for j in range(n - 1, -1, -1):
            if not (self.fused_tail and j == n - 1):
                self.bn_backward(f"decoder{j}.bn.", st.dec_c[j], st.dec_cr[j], b[f"dzd{j}"], b[f"yd{j}"], b[f"dyd{j}"])
            for name in pl.bias_group[f"dec{j}"]:
                self.wgrad(name)
            self.gemm(f"dec{j}.dg")
        for i in range(n - 1, -1, -1):
            self.bn_backward(f"encoder{i}.bn.", st.enc_c[i], st.enc_cr[i], b[f"dze{i}"], b[f"ye{i}"], b[f"dye{i}"])
            for name in pl.enc_wg[i]:
                self.wgrad(name)
            if i > 0:
                for name in [k for k in pl.specs if k.startswith(f"enc{i}.dg")]:
                    self.gemm(name)
        if self.side is not None and not torch.cuda.is_current_stream_capturing():
            call("sehip_stream_depend", stream(), self.side.cuda_stream, self._event())
        if tail is not None:
            if tb.uperm is not None:
                call("sehip_unpack_grad_sums_perm", ptr(self.gpack), ptr(tb.utab_g), ptr(tb.uperm), st.layout.n_params, ptr(grads), tail[2],
                     tail[3], tail[0], tail[1], tail[4], None, stream())
            else:
                call("sehip_unpack_grad_sums", ptr(self.gpack), ptr(tb.utab), st.layout.n_params, ptr(grads), tail[2], tail[3], tail[0],
                     tail[1], tail[4], None, stream())
        else:
            call("sehip_unpack_grad", ptr(self.gpack), ptr(tb.utab), st.layout.n_params, ptr(grads), stream())
        return grads
