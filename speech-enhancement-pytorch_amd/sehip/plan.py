"""Host-side plan of the DCCRN train step on libsehip: parameter layout, packing tables, implicit-GEMM
descriptors and the launch sequences of forward and backward.

Everything numeric runs in libsehip (HIP); this module only does bookkeeping with numpy at construction time:
  * ParamLayout  -- the reference's state_dict schema (src/model/dccrn.py:62-137) laid out in
                    ONE flat fp32 buffer (parameters) + one flat fp32 buffer (BatchNorm running statistics).
  * pack tables  -- how the GEMM-side bf16 weights / fp32 biases are gathered from the flat parameters and how the
                    packed gradients fold back (complex block matrix, tap order, skip-concat order, LSTM permutations).
  * descriptors  -- sehip_gemm_desc instances (include/sehip.h) for every conv / deconv / linear product,
                    its dgrad and its wgrad.
Channel-last layout [B][T][F][C]; decoder tensors that feed a BatchNorm keep the extra first frame the reference
drops after the BatchNorm (src/model/dccrn.py:193-196), i.e. they store T+1 frames and logical frame t lives at t+1.
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream, SehipError
from . import ops

BF16 = torch.bfloat16


# --------------------------------------------------------------------------------------------------
# ctypes mirrors of include/sehip.h
# --------------------------------------------------------------------------------------------------
class CSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("T", C.c_int32), ("tlo", C.c_int32), ("thi", C.c_int32), ("F", C.c_int32),
                ("C", C.c_int32), ("pad_", C.c_int32)]


class CDst(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("T", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("toff", C.c_int32),
                ("fmul", C.c_int32), ("fadd", C.c_int32), ("is_f32", C.c_int32), ("tmul", C.c_int32)]


class CGemmDesc(C.Structure):
    _fields_ = [("src", CSrc * 4), ("dst", CDst * 2), ("ktab", C.c_void_p), ("ntab", C.c_void_p), ("W", C.c_void_p),
                ("bias", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32),
                ("Npad", C.c_int32), ("K", C.c_int32), ("TT", C.c_int32), ("J", C.c_int32), ("fmul", C.c_int32),
                ("tmul", C.c_int32), ("cv_nf", C.c_int32), ("cv_fadd", C.c_int32), ("cv_toff", (C.c_int32 * 2) * 2),
                ("res", C.c_void_p), ("stats", C.c_void_p), ("stats_cr", C.c_int32), ("cv2_nkt", C.c_int32), ("cv2_nf", C.c_int32),
                ("cv2_fadd", C.c_int32), ("cv2_t0", C.c_int32), ("w_tiled", C.c_int32), ("wg_hint", C.c_int32), ("dense_rows", C.c_int32),
                ("dw_split_stride", C.c_int64),
                ("bn_dz", C.c_void_p), ("bn_y", C.c_void_p), ("bn_coef", C.c_void_p), ("bn_bcoef", C.c_void_p),
                ("bn_slope", C.c_void_p),
                ("bnr_y", C.c_void_p), ("bnr_coef", C.c_void_p), ("bnr_slope", C.c_void_p), ("bnr_part", C.c_void_p),
                ("gln_stats", C.c_void_p), ("gln_slope", C.c_void_p)]


# The gradient that arrives over the skip connection is added by the dgrad product that writes the encoder output's gradient
# (descriptor field `res`), so the BatchNorm backward kernels read one gradient tensor instead of two.
FUSE_SKIP_GRAD = not os.environ.get("SEHIP_NO_FUSE_SKIP")
# SEHIP_FUSE_ENC0_BN=1: the first encoder layer's BatchNorm backward apply pass runs inside its weight gradient (sehip_gemm_desc.bn_dz,
# csrc/gemm.hip narrow_wgrad_mfma_kernel): nobody else reads that layer's dOut, 84 MB less traffic.  Built, tested
# (tests/test_gpu_enc0_bn_wgrad.py) and 15-20 us SLOWER in the step (3.61 against 3.59 ms, same box): the fused kernel carries ~600 vector
# instructions per frame on the chain's last launches, where the two plain passes overlap with the tail of the weight-gradient
# stream.  Opt-in.  Needs the skip gradient already added (FUSE_SKIP_GRAD).
FUSE_ENC0_BN_WGRAD = FUSE_SKIP_GRAD and bool(os.environ.get("SEHIP_FUSE_ENC0_BN"))
# Products that conv_gemm_v3 takes get their packed weights in its tile order (the switches that take the kernel away keep [Npad][K])
# the apply pass of a layer with fused sums also finalizes them (sehip_cbn_finalize_apply_n)
# (read when a workspace is built: DCCRNWorkspace.fuse_finalize / fuse_bwd_finalize / BWD_REPLICAS rows of backward sums)
BWD_REPLICAS = int(os.environ.get("SEHIP_BWD_REPLICAS", "8"))
PARALLEL_HEAD = not os.environ.get("SEHIP_NO_PARALLEL_HEAD")      # weight packing + gradient clearing beside the STFT (A/B switch)
L2_GRAPH_EPOCH = 65535       # granule-tag epoch of the fused LSTM launches inside captured graphs; eager calls use 1 .. 65534
TILE_WEIGHTS = not any(os.environ.get(k) for k in ("SEHIP_NO_CONV_V3", "SEHIP_NO_PATCH", "SEHIP_NO_TILE_WEIGHTS"))


def npad_of(n):
    if n <= 16:
        return 16
    if n <= 32:
        return 32
    if n <= 64:
        return 64
    return (n + 127) // 128 * 128


def round_up(x, m):
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------------------------------
# configuration / parameter layout
# --------------------------------------------------------------------------------------------------
class DCCRNConfig:
    """Constructor arguments of the reference model (src/model/dccrn.py:12-27)."""

    def __init__(self, rnn_layers=2, rnn_units=128, win_len=400, win_inc=100, fft_len=512, length=16384,
                 win_type="hann", masking_mode="E", use_clstm=True, use_cbn=True, kernel_size=5,
                 kernel_num=(16, 32, 64, 128, 256, 256), **_ignored):
        # use_clstm=False (round 6): one real nn.LSTM over all channels, hidden rnn_units, ALWAYS two layers (src/model/dccrn.py:98-106
        # passes num_layers=2 whatever rnn_layers says), and the `tranform` Linear; csrc/lstm.hip with one recurrence per launch
        self.use_clstm = bool(use_clstm)
        # use_cbn=False (round 6): nn.BatchNorm2d over the [real half | imaginary half] channels (src/model/dccrn.py:110-113, :130-133) on
        # the ComplexBatchNorm kernels with the cross covariance taken as zero (csrc/cbn.hip cbn_fwd_record: eps < 0 selects it)
        self.use_cbn = bool(use_cbn)
        # (the window is data for the FFT front end: ones for None / 'None', any scipy.signal.get_window name otherwise --
        #  src/model/dccrn.py:650-653; checked here so that a bad name fails at construction)
        ops.window_of(win_type, win_len)
        self.win_type = win_type
        if kernel_size != 5:
            raise SehipError("sehip DCCRN: only kernel_size=5 is built")
        if not 1 <= int(rnn_layers) <= 8:
            raise SehipError("sehip DCCRN: rnn_layers must be 1 .. 8")
        if self.use_clstm and rnn_units not in (64, 128, 192, 256):
            raise SehipError("sehip DCCRN: rnn_units must be 64, 128, 192 or 256 (LSTM hidden 32 / 64 / 96 / 128: the sizes csrc/lstm.hip "
                             "is built for)")
        if not self.use_clstm and rnn_units not in (32, 64, 96, 128):
            raise SehipError("sehip DCCRN(use_clstm=False): rnn_units must be 32, 64, 96 or 128 (the hidden sizes csrc/lstm.hip is built for)")
        if masking_mode not in ("E", "C", "R"):
            raise SehipError(f"unknown masking_mode {masking_mode}")
        self.rnn_layers, self.rnn_units = (rnn_layers if self.use_clstm else 2), rnn_units
        self.win_len, self.win_inc, self.fft_len, self.length = win_len, win_inc, fft_len, length
        self.masking_mode = masking_mode
        self.kernel_size = kernel_size
        self.kernel_num = [2] + list(kernel_num)
        self.n_layers = len(self.kernel_num) - 1
        if self.n_layers != 6:
            raise SehipError("sehip DCCRN: six encoder/decoder layers are built (len(kernel_num) == 6)")
        self.hidden_dim = fft_len // (2 ** len(self.kernel_num))
        for c in self.kernel_num[1:]:
            if c % 16 or (c & (c - 1)):
                raise SehipError(f"sehip DCCRN: channel counts must be powers of two >= 16, got {kernel_num}")
        self.hid = rnn_units // 2 if self.use_clstm else rnn_units
        self.lstm_in = self.hidden_dim * self.kernel_num[-1] // 2  # per real/imag part

    def param_specs(self):
        """[(name, shape, kind)] in the reference's state_dict order; kind in param|buffer|nbt."""
        kn = self.kernel_num
        out = []

        def conv(pre, wshape, nb):
            for part in ("real_conv", "imag_conv"):
                out.append((f"{pre}0.{part}.weight", wshape, "param"))
                out.append((f"{pre}0.{part}.bias", (nb,), "param"))

        def bn(pre, n):
            if not self.use_cbn:      # nn.BatchNorm2d(2 n): weight, bias | running_mean, running_var, num_batches_tracked
                out.append((f"{pre}1.weight", (2 * n,), "param"))
                out.append((f"{pre}1.bias", (2 * n,), "param"))
                out.append((f"{pre}1.running_mean", (2 * n,), "buffer"))
                out.append((f"{pre}1.running_var", (2 * n,), "buffer"))
                out.append((f"{pre}1.num_batches_tracked", (), "nbt"))
                out.append((f"{pre}2.weight", (1,), "param"))
                return
            for k in ("Wrr", "Wri", "Wii", "Br", "Bi"):
                out.append((f"{pre}1.{k}", (n,), "param"))
            for k in ("RMr", "RMi", "RVrr", "RVri", "RVii"):
                out.append((f"{pre}1.{k}", (n,), "buffer"))
            out.append((f"{pre}1.num_batches_tracked", (), "nbt"))
            out.append((f"{pre}2.weight", (1,), "param"))

        for i in range(self.n_layers):
            cin, cout = kn[i] // 2, kn[i + 1] // 2
            conv(f"encoder.{i}.", (cout, cin, self.kernel_size, 2), cout)
            bn(f"encoder.{i}.", cout)
        for j, idx in enumerate(range(self.n_layers, 0, -1)):
            cin, cout = kn[idx], kn[idx - 1] // 2
            conv(f"decoder.{j}.", (cin, cout, self.kernel_size, 2), cout)
            if idx != 1:
                bn(f"decoder.{j}.", cout)
        h = self.hid
        if not self.use_clstm:
            for layer in range(2):
                out.append((f"enhance.weight_ih_l{layer}", (4 * h, 2 * self.lstm_in if layer == 0 else h), "param"))
                out.append((f"enhance.weight_hh_l{layer}", (4 * h, h), "param"))
                out.append((f"enhance.bias_ih_l{layer}", (4 * h,), "param"))
                out.append((f"enhance.bias_hh_l{layer}", (4 * h,), "param"))
            out.append(("tranform.weight", (2 * self.lstm_in, h), "param"))
            out.append(("tranform.bias", (2 * self.lstm_in,), "param"))
            return out
        for layer in range(self.rnn_layers):
            nin = self.lstm_in if layer == 0 else h
            for part in ("real_lstm", "imag_lstm"):
                q = f"enhance.{layer}.{part}."
                out.append((q + "weight_ih_l0", (4 * h, nin), "param"))
                out.append((q + "weight_hh_l0", (4 * h, h), "param"))
                out.append((q + "bias_ih_l0", (4 * h,), "param"))
                out.append((q + "bias_hh_l0", (4 * h,), "param"))
            if layer == self.rnn_layers - 1:
                for part in ("r_trans", "i_trans"):
                    out.append((f"enhance.{layer}.{part}.weight", (self.lstm_in, h), "param"))
                    out.append((f"enhance.{layer}.{part}.bias", (self.lstm_in,), "param"))
        return out


class ParamLayout:
    def __init__(self, cfg):
        self.cfg = cfg
        self.specs = cfg.param_specs()
        self.param_off, self.buffer_off, self.nbt_idx = {}, {}, {}
        po = bo = 0
        self.param_names, self.buffer_names, self.nbt_names = [], [], []
        for name, shape, kind in self.specs:
            n = int(np.prod(shape)) if len(shape) else 1
            if kind == "param":
                self.param_off[name] = (po, shape)
                self.param_names.append(name)
                po += round_up(n, 4)
            elif kind == "buffer":
                self.buffer_off[name] = (bo, shape)
                self.buffer_names.append(name)
                bo += round_up(n, 4)
            else:
                self.nbt_idx[name] = len(self.nbt_names)
                self.nbt_names.append(name)
        self.n_params = po
        self.n_buffers = bo
        # tensor boundaries for the reference's per-tensor grad_norm metric (src/solver.py:494-498)
        offs = [self.param_off[n][0] for n in self.param_names] + [po]
        self.tensor_offsets = np.asarray(offs, dtype=np.int64)

    # ComplexBatchNorm field -> (tensor, element offset inside it); None = the field does not exist (real BatchNorm2d: no cross terms)
    _REAL_BN = {"1.Wrr": ("1.weight", 0), "1.Wii": ("1.weight", 1), "1.Br": ("1.bias", 0), "1.Bi": ("1.bias", 1), "1.Wri": None,
                "1.RMr": ("1.running_mean", 0), "1.RMi": ("1.running_mean", 1), "1.RVrr": ("1.running_var", 0),
                "1.RVii": ("1.running_var", 1), "1.RVri": None}

    def bn_field(self, pre, k, cr):
        """(name, element offset) of ComplexBatchNorm field k ("1.Wrr", "1.RMr", ...) of layer `pre`, or None.  With use_cbn=False the
        layer is an nn.BatchNorm2d over 2 cr channels: its weight / bias / running statistics are the real and imaginary halves."""
        if getattr(self.cfg, "use_cbn", True) or k not in self._REAL_BN:
            return pre + k, 0
        m = self._REAL_BN[k]
        return None if m is None else (pre + m[0], m[1] * cr)

    def index_array(self, name):
        off, shape = self.param_off[name]
        n = int(np.prod(shape))
        return (off + np.arange(n, dtype=np.int64)).reshape(shape)


# --------------------------------------------------------------------------------------------------
# table helpers
# --------------------------------------------------------------------------------------------------
def enc_entry(idx, neg):
    """(index<<1)|neg, -1 where idx < 0."""
    idx = np.asarray(idx, dtype=np.int64)
    neg = np.broadcast_to(np.asarray(neg, dtype=np.int64), idx.shape)
    e = (idx << 1) | neg
    return np.where(idx < 0, -1, e).astype(np.int32)


def complex_block(wr, wi, transposed):
    """Index/sign arrays of the real block matrix of a complex conv.

    conv   (transposed=False): w [Co_r, Ci_r, kf, kt] -> full [Co, Ci, kf, kt]:  [[Wr, -Wi], [Wi, Wr]]
    deconv (transposed=True) : w [Ci_r, Co_r, kf, kt] -> full [Ci, Co, kf, kt]:  [[Wr, Wi], [-Wi, Wr]]
    """
    a, b = wr.shape[0], wr.shape[1]
    idx = np.empty((2 * a, 2 * b) + wr.shape[2:], dtype=np.int64)
    neg = np.zeros_like(idx)
    idx[:a, :b], idx[a:, b:] = wr, wr
    if not transposed:
        idx[:a, b:], idx[a:, :b] = wi, wi
        neg[:a, b:] = 1
    else:
        idx[:a, b:], idx[a:, :b] = wi, wi
        neg[a:, :b] = 1
    return idx, neg


class Arena:
    """Grows a list of numpy pieces that end up as one flat device buffer; returns element offsets."""

    def __init__(self, align):
        self.pieces, self.size, self.align = [], 0, align

    def add(self, arr):
        off = self.size
        self.pieces.append((off, arr))
        self.size += round_up(arr.shape[0], self.align)
        return off

    def reserve(self, n):
        off = self.size
        self.size += round_up(n, self.align)
        return off

    def build(self, dtype, width=None, fill=-1):
        shape = (self.size,) if width is None else (self.size, width)
        out = np.full(shape, fill, dtype=dtype)
        for off, arr in self.pieces:
            out[off:off + arr.shape[0]] = arr
        return out


class Gemm:
    """One implicit-GEMM problem: descriptor + the pieces needed to (re)bind pointers."""

    def __init__(self, name):
        self.name = name
        self.desc = CGemmDesc()
        self.w_off = None      # offset in packed bf16 weights
        self.b_off = None      # offset in packed fp32 bias
        self.dw_off = None     # offset in packed-gradient buffer
        self.db_off = None
        self.ktab = None
        self.ntab = None


def dense_ntab(n, npad, dst=0, base=0):
    t = np.zeros((npad // 4, 4), dtype=np.int32)
    for q in range(npad // 4):
        t[q] = (dst, base + 4 * q, max(0, min(4, n - 4 * q)), 0)
    return t


def wide_chunks(src, toff, fadd, c0, cn):
    """chunks covering channels [c0, c0+cn) of one source row."""
    assert cn % 8 == 0 and c0 % 8 == 0
    return [(src, toff, fadd, c0 + 8 * q) for q in range(cn // 8)]


def pad_ktab(rows):
    k = len(rows) * 8
    kp = round_up(k, 64)
    rows = rows + [(-1, 0, 0, 0)] * ((kp - k) // 8)
    return np.asarray(rows, dtype=np.int32).reshape(-1, 4), kp


def bind_chunk_table(src_tab, out_tab, kt_off, nchunks, geometry):
    """Binds the chunk rows [kt_off, kt_off + nchunks) of a (src, frame offset, row offset, channel offset) table to the
    geometry [(F, C)] of the product's sources: [src, (toff << 16) | (fadd & 0xffff), element delta, rows of a narrow chunk]
    (sehip_kchunk in include/sehip.h)."""
    rows = src_tab[kt_off:kt_off + nchunks]
    out = out_tab[kt_off:kt_off + nchunks]
    for q, (F, Cc) in enumerate(geometry):
        m = rows[:, 0] == q
        if not m.any():
            continue
        toff, fadd, coff = rows[m, 1].astype(np.int64), rows[m, 2].astype(np.int64), rows[m, 3].astype(np.int64)
        narrow = Cc == 2
        delta = (toff * F + fadd) * Cc + (0 if narrow else coff)
        assert np.abs(delta).max() < 2 ** 31 and np.abs(toff).max() < 2 ** 15 and np.abs(fadd).max() < 2 ** 15
        out[m, 1] = (((toff << 16) | (fadd & 0xffff)) & 0xffffffff).astype(np.uint32).view(np.int32)
        out[m, 2] = delta.astype(np.int32)
        out[m, 3] = coff.astype(np.int32) if narrow else 0


# --------------------------------------------------------------------------------------------------
# static (batch-independent) part: tables + packed layouts
# --------------------------------------------------------------------------------------------------
class GemmSpec:
    """Batch-independent description of one product (see sehip_gemm_desc)."""

    def __init__(self, name, rows, widx, wneg, n, bias_pairs, tt, j, fmul, srcs, dsts, ntab=None, kind="fwd", conv=None, res=None, npad=None):
        self.name = name
        self.conv = conv  # (nf, fadd, [[toff(s0,kt0), toff(s0,kt1)], [toff(s1,kt0), toff(s1,kt1)]]) or None
        self.ktab, self.K = pad_ktab(rows)
        self.N = n
        self.Npad = npad if npad is not None else npad_of(n)      # (npad: a width the generic kernel has a tile for, e.g. 192)
        k0 = widx.shape[1]
        wi = np.full((self.Npad, self.K), -1, dtype=np.int64)
        wn = np.zeros((self.Npad, self.K), dtype=np.int64)
        wi[:n, :k0] = widx
        wn[:n, :k0] = wneg
        self.widx, self.wneg = wi, wn
        self.bias_pairs = None
        if bias_pairs is not None:
            bp = np.full((self.Npad, 2), -1, dtype=np.int32)
            bp[:n] = bias_pairs
            self.bias_pairs = bp
        self.tt, self.J, self.fmul = tt, j, fmul      # tt: "T" or "T+1"
        self.srcs = srcs   # list of (buffer name, tlo_mode) ; geometry comes from the buffer
        self.dsts = dsts   # list of (buffer name, toff, fmul, fadd)
        self.ntab = ntab if ntab is not None else dense_ntab(n, self.Npad)
        self.kind = kind
        self.res = res     # buffer added to what goes to dsts[0] (same layout), or None
        self.w_off = self.b_off = self.dw_off = self.db_off = None
        self.kt_off = self.nt_off = None
        self.stats_of = None   # BatchNorm prefix whose batch statistics this forward product accumulates (sehip_gemm_desc.stats)
        self.w_tiled = False   # packed weights in conv_gemm_v3's tile order (sehip_gemm_desc.w_tiled)

    def v3_channels(self):
        """(C0, C1) of the sources when conv_gemm_v3 (csrc/conv3.hip, sehip_try_conv_gemm_v3) takes this product, else None:
        regular convolution with (taps, row stride) in {(5, 2), (3, 1), (2, 1)}, 4 / 8 / 16 / 32 rows per frame, 16-multiple
        source channels, 128-multiple outputs, frame offsets within one frame of each other."""
        if self.conv is None or self.kind == "wgrad_only" or self.Npad % 64 or self.J not in (4, 8, 16, 32):
            return None
        if self.stats_of is not None and self.Npad % 128 and self.Npad != 64:
            return None
        nf, _, toff = self.conv
        if (nf, self.fmul) not in ((5, 2), (3, 1), (2, 1)):
            return None
        cs = [int((self.ktab[:, 0] == q).sum()) * 8 // (2 * nf) for q in (0, 1)]
        if cs[0] == 0 or cs[0] % 16 or cs[1] % 16 or self.K != 2 * nf * (cs[0] + cs[1]):
            return None
        for q in range(2 if cs[1] else 1):
            if max(abs(toff[q][0]), abs(toff[q][1])) > 1 or abs(toff[q][0] - toff[q][1]) > 1:
                return None
        return cs

    def stream_takes(self):
        """True when convs_stream_kernel (csrc/convt.hip, sehip_try_convs_stream) takes this product at any batch size: a 5-tap stride-2
        convolution of ONE source with 32 input channels and 4 KB frames on both sides -- encoder 2's forward product (with the fused
        BatchNorm sums) and decoder 3's input gradient (two destinations) at the headline widths.  Such a product keeps its weights in
        [Npad][K] order (the kernel holds them in registers); the 16-channel layers never qualified for conv_gemm_v3's tile order."""
        if os.environ.get("SEHIP_NO_CONVT_STREAM") or self.conv is None or self.kind == "wgrad_only" or self.res is not None:
            return False
        nf, fadd, _ = self.conv
        if (nf, fadd, self.fmul) != (5, -2, 2) or len(self.srcs) != 1 or self.N != self.Npad:
            return False
        c_in = self.K // 10
        if self.K != 10 * c_in or c_in != 32 or 2 * self.J * c_in != 2048 or self.N * self.J * 2 != 4096 * len(self.dsts):
            return False
        if len(self.dsts) == 1:
            return self.kind == "fwd" and self.stats_of is not None
        return len(self.dsts) == 2 and self.bias_pairs is None

    def tile_weights(self):
        """Moves the packed weights [Npad][K] (K ordered (tap, concatenated channel)) into the tile order conv_gemm_v3 streams:
        [n tile][16-channel chunk][tap pair][tap of the pair][128 (or 64) n][16 channels]; a pure permutation of the packing table."""
        cs = self.v3_channels()
        assert cs is not None and not self.w_tiled
        nf, ctot = self.conv[0], cs[0] + cs[1]
        bn = 64 if self.Npad % 128 else 128                  # the kernel's column tile
        ntn, nch = self.Npad // bn, ctot // 16

        def perm(a):
            a = a.reshape(ntn, bn, nf, 2, nch, 16)            # [nt][n][j][u][ch][c]   (tap index = 2 j + u)
            return a.transpose(0, 4, 2, 3, 1, 5).reshape(self.Npad, self.K)
        self.widx_packed, self.wneg_packed = perm(self.widx), perm(self.wneg)
        self.w_tiled = True

    def tile_complex_columns(self):
        """Re-orders the output columns [re 0..Cr) | im 0..Cr) so that every 128-column tile holds [64 re | 64 im] of the SAME 64
        complex channels (what the fused BatchNorm statistics of conv_gemm_v2 need); rows of W / bias and the column table move
        together, so the weight-gradient product and the un-packing of its dW follow automatically."""
        n = self.N
        assert n == self.Npad and n % 128 == 0 and self.ntab.shape[0] == n // 4
        cr = n // 2
        perm = np.concatenate([np.concatenate([np.arange(64 * t, 64 * t + 64), cr + np.arange(64 * t, 64 * t + 64)])
                               for t in range(cr // 64)])
        self.widx, self.wneg = self.widx[perm], self.wneg[perm]
        if self.bias_pairs is not None:
            self.bias_pairs = self.bias_pairs[perm]
        self.ntab = self.ntab.reshape(n // 4, 4)[perm[::4] // 4].copy()


class DCCRNStatic:
    def _maybe_split_decoder(self, j, pre, full, neg, cat1, cat2, c1, c2, co, f_in, s1, s1_mode, s2, bias_pairs):
        """Round 6: a deep decoder layer's forward product as TWO products over its two sources (ComplexConvTranspose2d of
        complex_cat([main, skip]), src/model/dccrn.py:186-197, :387-450, is linear in the concatenated channels): the skip-connection
        half `dec{j}.fs{p}` (source: the encoder activation, ready since the encoder ran) runs on the weight-gradient stream, idle in the
        forward pass, UNDER the two-layer LSTM (150 us on 64 of 256 CUs) and leaves a bf16 partial tensor `pd{j}`; the main half
        `dec{j}.fm{p}` behind the LSTM has half the K steps, starts from nothing but adds the partial tile before it takes the fused
        BatchNorm sums and stores (conv_gemm_v3's residual, csrc/conv3.hip).  Forward-only products: the weight gradients keep the
        two-source descriptors `dec{j}.fwd{p}`.  Only where conv_gemm_v3 takes both halves."""
        # MEASURED AND NOT THE DEFAULT (SEHIP_DEC_SPLIT=1 to enable): B = 32, same box, ms per step 3.166 / 3.161 with the split against
        # 3.144 / 3.136 without (and 3.211 / 3.210 / 3.202 against 3.184 / 3.193 / 3.186 on another box).  The main halves do shrink
        # (dec0 99 -> 72 us, dec1 105 -> 76, dec2 75 -> 69: prologue + epilogue + the partial tile's read do not halve with K), but the
        # LSTM launch beside the three skip halves takes 216 us instead of 147 (its per-step hand-offs go through the same L2 the
        # convolutions' DMA streams load; wave priority does not help, csrc/lstm2.hip) and its input product 54 instead of 39.
        if not os.environ.get("SEHIP_DEC_SPLIT") or not TILE_WEIGHTS or co % 64 or c1 % 16 or c2 % 16:
            return
        names = []
        for p in (0, 1):
            ds = (-1, 0, 1) if p == 0 else (0, 1)
            off1 = 0 if j == 0 else 1
            rows_m, rows_s, cm_i, cm_n, cs_i, cs_n = [], [], [], [], [], []
            for kt in (0, 1):
                for d in ds:
                    kf = p + 2 - 2 * d
                    rows_m += wide_chunks(0, -kt + off1, d, 0, c1)
                    rows_s += wide_chunks(0, -kt, d, 0, c2)
                    cm_i.append(full[cat1, :, kf, kt].T); cm_n.append(neg[cat1, :, kf, kt].T)
                    cs_i.append(full[cat2, :, kf, kt].T); cs_n.append(neg[cat2, :, kf, kt].T)
            fs = GemmSpec(f"dec{j}.fs{p}", rows_s, np.concatenate(cs_i, 1), np.concatenate(cs_n, 1), co, None, "T+1", f_in, 1,
                          [(s2, "all")], [(f"pd{j}", 0, 2, p)], kind="fwd_only", conv=(len(ds), ds[0], [[0, -1], [0, 0]]))
            fm = GemmSpec(f"dec{j}.fm{p}", rows_m, np.concatenate(cm_i, 1), np.concatenate(cm_n, 1), co, bias_pairs, "T+1", f_in, 1,
                          [(s1, s1_mode)], [(f"yd{j}", 0, 2, p)], kind="fwd_only", conv=(len(ds), ds[0], [[off1, off1 - 1], [0, 0]]),
                          res=f"pd{j}")
            if fs.v3_channels() is None or fm.v3_channels() is None:
                return
            names.append((fs, fm))
        ref = self.specs[f"dec{j}.fwd0"]
        for fs, fm in names:
            if ref.stats_of is not None:
                if co > 128:
                    fs.tile_complex_columns(); fm.tile_complex_columns()
                fm.stats_of = ref.stats_of
            self.specs[fs.name] = fs; self.specs[fm.name] = fm
        self.dec_split.append(j)

    def _maybe_fuse_stats(self, pre, names, co, cins, J):
        """Forward products that conv_gemm_v2 takes (64-multiple source channels, 128-multiple outputs, J | 128, J <= 64) also
        accumulate the batch statistics of the ComplexBatchNorm behind them: no cbn_stats pass for these layers."""
        if self.deterministic or any(os.environ.get(k) for k in ("SEHIP_NO_FUSE_STATS", "SEHIP_NO_PATCH", "SEHIP_NO_CONV_V2")):
            return      # (the experiment switches that take conv_gemm_v2 away also take its statistics away; the deterministic
                        #  schedule takes the sums from the separate pass: the epilogues add them with fp32 atomics)
        # (round 3: conv_gemm_v3 also takes the sums of a 64-output layer -- one [32 re | 32 im] tile; only that kernel does, so not
        #  when one of its switches is set)
        v3_only = co == 64 and all(c % 16 == 0 for c in cins) and J in (4, 8, 16, 32) and TILE_WEIGHTS and not os.environ.get("SEHIP_NO_FUSE_STATS64")
        if co in (16, 32) and not os.environ.get("SEHIP_NO_FUSE_STATS32"):
            # a 32- / 16-output layer on the small-channel kernel: whether that kernel takes the product (its patch must fit into LDS)
            # depends on the workspace, so the workspace asks the library when it binds the descriptors (small_stats)
            self.small_stats[pre] = list(names)
            return
        if (co % 128 or any(c % 64 for c in cins) or J > 64 or 128 % J) and not v3_only:
            return
        for nm in names:
            sp = self.specs[nm]
            if co > 128:
                sp.tile_complex_columns()
            sp.stats_of = pre
        self.fused_stats.add(pre)

    def __init__(self, cfg: DCCRNConfig, deterministic=False):
        self.deterministic = bool(deterministic)
        self.fused_stats = set()
        self.small_stats = {}  # BatchNorm prefix -> forward products of a 32-output layer (fused sums if conv_small2 takes them)
        self.cfg = cfg
        self.layout = L = ParamLayout(cfg)
        kn = cfg.kernel_num
        self.specs = {}
        self.bn = []  # (prefix, Cr)
        self.dec_split = []    # decoder layers whose forward product runs as skip half (under the LSTM) + main half: _maybe_split_decoder
        ia = L.index_array
        self.F0 = cfg.fft_len // 2  # 256 bins after dropping DC

        def eff_bias(pre, cr):
            br, bi = ia(pre + "0.real_conv.bias"), ia(pre + "0.imag_conv.bias")
            pairs = np.empty((2 * cr, 2), dtype=np.int32)
            pairs[:cr, 0] = enc_entry(br, 0); pairs[:cr, 1] = enc_entry(bi, 1)
            pairs[cr:, 0] = enc_entry(br, 0); pairs[cr:, 1] = enc_entry(bi, 0)
            return pairs

        # ---------------- encoder ----------------
        for i in range(6):
            ci, co = kn[i], kn[i + 1]
            pre = f"encoder.{i}."
            full, neg = complex_block(ia(pre + "0.real_conv.weight"), ia(pre + "0.imag_conv.weight"), False)  # [co,ci,5,2]
            src = "enc_in" if i == 0 else f"z{i - 1}"
            if ci >= 8:
                rows = []
                for kt in (0, 1):
                    for kf in range(5):
                        rows += wide_chunks(0, kt - 1, kf - 2, 0, ci)
                widx = full.transpose(0, 3, 2, 1).reshape(co, -1)
                wneg = neg.transpose(0, 3, 2, 1).reshape(co, -1)
            else:
                assert ci == 2
                rows, widx, wneg = [], np.full((co, 32), -1, np.int64), np.zeros((co, 32), np.int64)
                for kt in (0, 1):
                    rows += [(0, kt - 1, -2, 4), (0, kt - 1, 2, 1)]
                    for kf in range(5):
                        for c in range(2):
                            widx[:, kt * 16 + kf * 2 + c] = full[:, c, kf, kt]
                            wneg[:, kt * 16 + kf * 2 + c] = neg[:, c, kf, kt]
            self.specs[f"enc{i}.fwd"] = GemmSpec(f"enc{i}.fwd", rows, widx, wneg, co, eff_bias(pre, co // 2), "T",
                                                  self.F0 >> (i + 1), 2, [(src, "all")], [(f"y{i}", 0, 1, 0)],
                                                  conv=(5, -2, [[-1, 0], [0, 0]]))  # ci == 2: read by the narrow wgrad only
            self.bn.append((pre, co // 2))
            self._maybe_fuse_stats(pre, [f"enc{i}.fwd"], co, [ci], self.F0 >> (i + 1))
            if i >= 1:
                # dgrad by output-row parity: dX[b,t,2j+p,ci] = sum dY[b,t+1-kt,j+d,co] * full[co,ci,kf,kt], kf = p+2-2d
                for p in (0, 1):
                    ds = (-1, 0, 1) if p == 0 else (0, 1)
                    rows = []
                    for kt in (0, 1):
                        for d in ds:
                            rows += wide_chunks(0, 1 - kt, d, 0, co)
                    w = np.empty((ci, 2, len(ds), co), dtype=np.int64)
                    wn = np.empty_like(w)
                    for kt in (0, 1):
                        for q, d in enumerate(ds):
                            kf = p + 2 - 2 * d
                            w[:, kt, q, :] = full[:, :, kf, kt].T
                            wn[:, kt, q, :] = neg[:, :, kf, kt].T
                    self.specs[f"enc{i}.dg{p}"] = GemmSpec(f"enc{i}.dg{p}", rows, w.reshape(ci, -1), wn.reshape(ci, -1), ci,
                                                           None, "T", self.F0 >> (i + 1), 1, [(f"dye{i}", "all")],
                                                           [(f"dz{i - 1}", 0, 2, p)], kind="dgrad",
                                                           conv=(len(ds), ds[0], [[1, 0], [0, 0]]),
                                                           res=f"dskip{i - 1}" if FUSE_SKIP_GRAD else None)

        # ---------------- decoder ----------------
        for j in range(6):
            idx = 6 - j
            c1 = c2 = kn[idx]
            co = kn[idx - 1]
            f_in = self.F0 >> idx
            pre = f"decoder.{j}."
            full, neg = complex_block(ia(pre + "0.real_conv.weight"), ia(pre + "0.imag_conv.weight"), True)  # [cin,co,5,2]
            crt = (c1 + c2) // 2
            # concatenated-channel index of (source s, channel c)   (complex_cat: [a_r, b_r, a_i, b_i])
            cat1 = np.concatenate([np.arange(c1 // 2), crt + np.arange(c1 // 2)])
            cat2 = np.concatenate([c1 // 2 + np.arange(c2 // 2), crt + c1 // 2 + np.arange(c2 // 2)])
            s1 = "P" if j == 0 else f"zd{j - 1}"
            s2 = f"z{5 - j}"
            s1_mode = "all" if j == 0 else "drop1"  # logical frame t lives at stored t+1
            last = j == 5
            for p in (0, 1):
                ds = (-1, 0, 1) if p == 0 else (0, 1)
                rows = []
                cols_i, cols_n = [], []
                for kt in (0, 1):
                    for d in ds:
                        kf = p + 2 - 2 * d
                        base_t = (1 if last else 0) - kt  # last layer produces only frames 1..T of the T+1
                        rows += wide_chunks(0, base_t + (0 if j == 0 else 1), d, 0, c1)
                        rows += wide_chunks(1, base_t, d, 0, c2)
                        cols_i += [full[cat1, :, kf, kt].T, full[cat2, :, kf, kt].T]
                        cols_n += [neg[cat1, :, kf, kt].T, neg[cat2, :, kf, kt].T]
                widx, wneg = np.concatenate(cols_i, 1), np.concatenate(cols_n, 1)
                if last:
                    nt = np.zeros((4, 4), dtype=np.int32)
                    nt[0] = (0, 0, 2, 0)
                    dst = [("mask", 0, 2, p)]
                    tt = "T"
                else:
                    nt = None
                    dst = [(f"yd{j}", 0, 2, p)]
                    tt = "T+1"
                bt = 1 if last else 0
                self.specs[f"dec{j}.fwd{p}"] = GemmSpec(f"dec{j}.fwd{p}", rows, widx, wneg, co, eff_bias(pre, co // 2), tt,
                                                        f_in, 1, [(s1, s1_mode), (s2, "all")], dst, ntab=nt,
                                                        conv=(len(ds), ds[0], [[bt + (0 if j == 0 else 1), bt - 1 + (0 if j == 0 else 1)],
                                                                               [bt, bt - 1]]))
            if not last:
                self.bn.append((pre, co // 2))
                self._maybe_fuse_stats(pre, [f"dec{j}.fwd0", f"dec{j}.fwd1"], co, [c1, c2], f_in)
                self._maybe_split_decoder(j, pre, full, neg, cat1, cat2, c1, c2, co, f_in, s1, s1_mode, s2, eff_bias(pre, co // 2))
            # dgrad: dIn[b,t,fi,(s,c)] = sum dOut'[b,t+kt,2fi-2+kf,co] * full[cin,co,kf,kt]
            gsrc = "dmask" if last else f"dyd{j}"
            if co >= 8:
                rows = []
                for kt in (0, 1):
                    for kf in range(5):
                        rows += wide_chunks(0, kt, kf - 2, 0, co)
                w = full.transpose(0, 3, 2, 1).reshape(c1 + c2, -1)   # [cin, (kt,kf,co)]
                wn = neg.transpose(0, 3, 2, 1).reshape(c1 + c2, -1)
            else:
                assert co == 2
                rows, w, wn = [], np.full((c1 + c2, 32), -1, np.int64), np.zeros((c1 + c2, 32), np.int64)
                for kt in (0, 1):
                    rows += [(0, kt - 1, -2, 4), (0, kt - 1, 2, 1)]  # dmask stores logical frame t' at t'-1
                    for kf in range(5):
                        for c in range(2):
                            w[:, kt * 16 + kf * 2 + c] = full[:, c, kf, kt]
                            wn[:, kt * 16 + kf * 2 + c] = neg[:, c, kf, kt]
            order = np.concatenate([cat1, cat2])
            d1 = ("dP", 0, 1, 0) if j == 0 else (f"dzd{j - 1}", 1, 1, 0)
            nt = np.concatenate([dense_ntab(c1, c1, 0, 0), dense_ntab(c2, c2, 1, 0)])
            assert npad_of(c1 + c2) == c1 + c2
            self.specs[f"dec{j}.dg"] = GemmSpec(f"dec{j}.dg", rows, w[order], wn[order], c1 + c2, None, "T", f_in, 2,
                                                [(gsrc, "all")], [d1, (f"dskip{5 - j}", 0, 1, 0)], ntab=nt, kind="dgrad",
                                                conv=(5, -2, [[0, 1], [0, 0]]) if co >= 8 else (5, -2, [[-1, 0], [0, 0]]))

        # ---------------- complex LSTM ----------------
        c5 = kn[6]
        cp = c5 // 2
        h = cfg.hid
        lstm = ("real_lstm", "imag_lstm")

        if not cfg.use_clstm:
            # ---- use_clstm=False (src/model/dccrn.py:98-106, :184-189): ONE real nn.LSTM over all c5 channels, two layers, hidden
            #      rnn_units, and the `tranform` Linear back to 4 * c5 features.  Feature index of the reference: c * 4 + f.
            def bias_pairs_real(layer):
                bp = np.empty((4 * h, 2), dtype=np.int32)
                bp[:, 0] = enc_entry(ia(f"enhance.bias_ih_l{layer}"), 0)
                bp[:, 1] = enc_entry(ia(f"enhance.bias_hh_l{layer}"), 0)
                return bp

            zero = lambda a: np.zeros_like(a)
            w1 = ia("enhance.weight_ih_l0").reshape(4 * h, c5, 4).transpose(0, 2, 1).reshape(4 * h, -1)      # [4h, (f, c)]
            rows = []
            for f in range(4):
                rows += wide_chunks(0, 0, f, 0, c5)
            self.specs["ih1"] = GemmSpec("ih1", rows, w1, zero(w1), 4 * h, bias_pairs_real(0), "T", 1, 1, [("z5", "all")],
                                         [("pre1", 0, 1, 0)])
            self.specs["dx1"] = GemmSpec("dx1", wide_chunks(0, 0, 0, 0, 4 * h), w1.T.copy(), zero(w1.T), 4 * c5, None, "T", 1, 1,
                                         [("dpre1", "all")], [("dz5l", 0, 1, 0)], kind="dgrad", res="dskip5" if FUSE_SKIP_GRAD else None)
            w2 = ia("enhance.weight_ih_l1")
            self.specs["ih2"] = GemmSpec("ih2", wide_chunks(0, 0, 0, 0, h), w2, zero(w2), 4 * h, bias_pairs_real(1), "T", 1, 1,
                                         [("h1_0", "all")], [("pre2", 0, 1, 0)])
            self.specs["dx2"] = GemmSpec("dx2", wide_chunks(0, 0, 0, 0, 4 * h), w2.T.copy(), zero(w2.T), h, None, "T", 1, 1,
                                         [("dpre2", "all")], [("dx2", 0, 1, 0)], kind="dgrad")
            tr = ia("tranform.weight")                       # [c5 * 4, h], row c * 4 + d
            wt = tr.reshape(c5, 4, h).transpose(1, 0, 2).reshape(4 * c5, h)          # row n' = d * c5 + c = the column of P
            bp = np.full((4 * c5, 2), -1, dtype=np.int32)
            bp[:, 0] = enc_entry(ia("tranform.bias").reshape(c5, 4).T.reshape(-1), 0)
            self.specs["proj"] = GemmSpec("proj", wide_chunks(0, 0, 0, 0, h), wt, zero(wt), 4 * c5, bp, "T", 1, 1, [("h2_0", "all")],
                                          [("P", 0, 1, 0)])
            rows = []
            for d in range(4):
                rows += wide_chunks(0, 0, d, 0, c5)
            self.specs["dproj"] = GemmSpec("dproj", rows, wt.T.copy(), zero(wt.T), h, None, "T", 1, 1, [("dP", "all")],
                                           [("dxo", 0, 1, 0)], kind="dgrad")
            for layer in (1, 2):
                hh = ia(f"enhance.weight_hh_l{layer - 1}")
                self.specs[f"hh{layer}"] = GemmSpec(f"hh{layer}", wide_chunks(0, -1, 0, 0, h), hh, zero(hh), 4 * h, None, "T", 1, 1,
                                                    [(f"h{layer}_0", "all")], [(f"dpre{layer}", 0, 1, 0)],
                                                    ntab=dense_ntab(4 * h, 4 * h, 0, 0), kind="wgrad_only")
        else:
            def ih(layer, l):
                return ia(f"enhance.{layer}.{lstm[l]}.weight_ih_l0")

            def bias_pairs_lstm(layer):
                bp = np.empty((2 * 4 * h, 2), dtype=np.int32)
                for l in (0, 1):
                    bp[l * 4 * h:(l + 1) * 4 * h, 0] = enc_entry(ia(f"enhance.{layer}.{lstm[l]}.bias_ih_l0"), 0)
                    bp[l * 4 * h:(l + 1) * 4 * h, 1] = enc_entry(ia(f"enhance.{layer}.{lstm[l]}.bias_hh_l0"), 0)
                return bp

            zero = lambda a: np.zeros_like(a)
            # layer 1 input products: x[(f,c)] with reference feature index c*4+f  (src/model/dccrn.py:170-176)
            w1 = np.concatenate([ih(0, l).reshape(4 * h, cp, 4).transpose(0, 2, 1).reshape(4 * h, -1) for l in (0, 1)])  # [512,(f,c)]
            for q, tag in enumerate("ri"):
                rows = []
                for f in range(4):
                    rows += wide_chunks(0, 0, f, q * cp, cp)
                self.specs[f"ih1_{tag}"] = GemmSpec(f"ih1_{tag}", rows, w1, zero(w1), 8 * h, bias_pairs_lstm(0), "T", 1, 1,
                                                    [("z5", "all")], [(f"pre1_{tag}", 0, 1, 0)])
                # dx1: d z5[b,t,f,q*cp+c] = dpre1_q @ W
                nt = np.zeros((4 * cp // 4, 4), dtype=np.int32)
                for n4 in range(4 * cp // 4):
                    f, c = divmod(4 * n4, cp)
                    nt[n4] = (0, f * c5 + q * cp + c, 4, 0)
                self.specs[f"dx1_{tag}"] = GemmSpec(f"dx1_{tag}", wide_chunks(0, 0, 0, 0, 8 * h), w1.T.copy(), zero(w1.T), 4 * cp,
                                                    None, "T", 1, 1, [(f"dpre1_{tag}", "all")], [("dz5l", 0, 1, 0)], ntab=nt,
                                                    kind="dgrad", res="dskip5" if FUSE_SKIP_GRAD else None)
                # (res: the gradient that arrived over the innermost skip connection is added where the LSTM's input gradient is stored
                #  -- each of the two products adds it to its own half of the columns -- so that encoder 5's BatchNorm backward reads one
                #  gradient tensor like every other layer's: its apply pass 91 -> 60 us)
            # layer 2 input products and the projection: x2_r = h1[r,real] - h1[i,imag]; x2_i = h1[i,real] + h1[r,imag]
            combos = {"r": (0, 3, 1), "i": (2, 1, 0)}  # (first combo, second combo, negate second)
            # (the reference stacks rnn_layers of them, src/model/dccrn.py:86-96; every layer behind the first reads the one before it this way)
            L = cfg.rnn_layers
            for tag, (ca, cb, ng) in combos.items():
                rows = wide_chunks(0, 0, 0, 0, h) + wide_chunks(1, 0, 0, 0, h)
                for layer in range(2, L + 1):
                    w2 = np.concatenate([ih(layer - 1, l) for l in (0, 1)])  # [512, 64]
                    wi = np.concatenate([w2, w2], 1)
                    wn = np.concatenate([zero(w2), zero(w2) + ng], 1)
                    self.specs[f"ih{layer}_{tag}"] = GemmSpec(f"ih{layer}_{tag}", rows, wi, wn, 8 * h, bias_pairs_lstm(layer - 1), "T", 1, 1,
                                                              [(f"h{layer - 1}_{ca}", "all"), (f"h{layer - 1}_{cb}", "all")],
                                                              [(f"pre{layer}_{tag}", 0, 1, 0)])
                    self.specs[f"dx{layer}_{tag}"] = GemmSpec(f"dx{layer}_{tag}", wide_chunks(0, 0, 0, 0, 8 * h), w2.T.copy(), zero(w2.T), h,
                                                              None, "T", 1, 1, [(f"dpre{layer}_{tag}", "all")],
                                                              [(f"dx{layer}_{tag}", 0, 1, 0)], kind="dgrad")
                q = 0 if tag == "r" else 1
                tr = ia(f"enhance.{L - 1}.{tag}_trans.weight")        # [cp*4, h], row c*4+d
                trb = ia(f"enhance.{L - 1}.{tag}_trans.bias")
                wt = tr.reshape(cp, 4, h).transpose(1, 0, 2).reshape(4 * cp, h)   # row n' = d*cp + c
                bt = trb.reshape(cp, 4).T.reshape(-1)
                bp = np.full((4 * cp, 2), -1, dtype=np.int32)
                bp[:, 0] = enc_entry(bt, 0)
                nt = np.zeros((4 * cp // 4, 4), dtype=np.int32)
                for n4 in range(4 * cp // 4):
                    d, c = divmod(4 * n4, cp)
                    nt[n4] = (0, d * c5 + q * cp + c, 4, 0)
                self.specs[f"proj_{tag}"] = GemmSpec(f"proj_{tag}", rows, np.concatenate([wt, wt], 1),
                                                     np.concatenate([zero(wt), zero(wt) + ng], 1), 4 * cp, bp, "T", 1, 1,
                                                     [(f"h{L}_{ca}", "all"), (f"h{L}_{cb}", "all")], [("P", 0, 1, 0)], ntab=nt)
                rows = []
                for d in range(4):
                    rows += wide_chunks(0, 0, d, q * cp, cp)
                self.specs[f"dproj_{tag}"] = GemmSpec(f"dproj_{tag}", rows, wt.T.copy(), zero(wt.T), h, None, "T", 1, 1,
                                                      [("dP", "all")], [(f"dxo_{tag}", 0, 1, 0)], kind="dgrad")
            # recurrent weight gradients: dW_hh[n,k] = sum dpre[b,t,n] h[b,t-1,k]
            for layer in range(1, L + 1):
                for combo in range(4):
                    part, l = combo >> 1, combo & 1
                    hh = ia(f"enhance.{layer - 1}.{lstm[l]}.weight_hh_l0")
                    nt = dense_ntab(4 * h, 4 * h, 0, l * 4 * h)
                    self.specs[f"hh{layer}_{combo}"] = GemmSpec(f"hh{layer}_{combo}", wide_chunks(0, -1, 0, 0, h), hh, zero(hh),
                                                                4 * h, None, "T", 1, 1, [(f"h{layer}_{combo}", "all")],
                                                                [(f"dpre{layer}_{'ri'[part]}", 0, 1, 0)], ntab=nt, kind="wgrad_only")

        # ---------------- arenas: packed weights / biases / gradient regions / tables ----------------
        wa, ba, ga = Arena(64), Arena(16), Arena(16)
        kta, nta = Arena(1), Arena(1)
        self.wgrad_of = {}
        for name, s in self.specs.items():
            s.kt_off = kta.add(s.ktab)
            s.nt_off = nta.add(s.ntab)
            if s.kind != "wgrad_only":
                if TILE_WEIGHTS and s.v3_channels() is not None and not s.stream_takes():
                    s.tile_weights()      # (widx / wneg themselves keep the [Npad][K] order: the weight-gradient un-packing uses them)
                    s.w_off = wa.add(enc_entry(s.widx_packed, s.wneg_packed).reshape(-1))
                else:
                    s.w_off = wa.add(enc_entry(s.widx, s.wneg).reshape(-1))
            if s.bias_pairs is not None:
                s.b_off = ba.add(s.bias_pairs)
            if s.kind in ("fwd", "wgrad_only"):
                s.dw_off = ga.reserve(s.Npad * s.K)
                if s.bias_pairs is not None:
                    s.db_off = ga.reserve(s.Npad)
        # recurrent weights for the LSTM kernels: whh [layer][lstm][256][64], whhT [layer][lstm][64][256]
        self.whh_off, self.whhT_off = {}, {}
        for layer in range(1, cfg.rnn_layers + 1):
            if cfg.use_clstm:
                hh = np.stack([ia(f"enhance.{layer - 1}.{lstm[l]}.weight_hh_l0") for l in (0, 1)])
            else:
                hh = ia(f"enhance.weight_hh_l{layer - 1}")
            self.whh_off[layer] = wa.add(enc_entry(hh, 0).reshape(-1))
            self.whhT_off[layer] = wa.add(enc_entry(np.swapaxes(hh, -1, -2), 0).reshape(-1))
        # layer 2's input weights for the fused two-layer recurrence (csrc/lstm2.hip): wih2 [lstm][256][64], wihT2 [lstm][64][256]
        if cfg.use_clstm and cfg.rnn_layers == 2 and h == 64:
            ih2 = np.stack([ia(f"enhance.1.{lstm[l]}.weight_ih_l0") for l in (0, 1)])
            self.wih2_off = wa.add(enc_entry(ih2, 0).reshape(-1))
            self.wihT2_off = wa.add(enc_entry(ih2.transpose(0, 2, 1), 0).reshape(-1))
        # BatchNorm / PReLU gradients land in the packed-gradient buffer too
        self.bn_g_off = {}
        for pre, cr in self.bn:
            self.bn_g_off[pre] = {k: ga.reserve(cr) for k in ("Wrr", "Wri", "Wii", "Br", "Bi")}
            self.bn_g_off[pre]["slope"] = ga.reserve(1)
        self.n_wpack, self.n_bpack, self.n_gpack = wa.size, ba.size, ga.size
        self.wtab = wa.build(np.int32)
        self.btab = ba.build(np.int32, 2)
        self.ktab = kta.build(np.int32, 4)
        self.ntab = nta.build(np.int32, 4, fill=0)
        self.utab = self._build_unpack_table()

    def _build_unpack_table(self):
        L = self.layout
        ps, gs, ns = [], [], []
        for s in self.specs.values():
            if s.dw_off is None:
                continue
            m = s.widx >= 0
            ps.append(s.widx[m]); ns.append(s.wneg[m])
            gs.append(s.dw_off + np.flatnonzero(m.reshape(-1)))
            if s.db_off is not None:
                for col in (0, 1):
                    e = s.bias_pairs[:, col].astype(np.int64)
                    mm = e >= 0
                    ps.append(e[mm] >> 1); ns.append(e[mm] & 1)
                    gs.append(s.db_off + np.flatnonzero(mm))
        for pre, cr in self.bn:
            for k in ("Wrr", "Wri", "Wii", "Br", "Bi"):
                f = L.bn_field(pre, "1." + k, cr)
                if f is None:         # (real BatchNorm2d: the cross weight does not exist; its slot of the packed gradients stays unread)
                    continue
                ps.append(L.index_array(f[0]).reshape(-1)[f[1]:f[1] + cr]); ns.append(np.zeros(cr, np.int64))
                gs.append(self.bn_g_off[pre][k] + np.arange(cr))
            ps.append(L.index_array(pre + "2.weight")); ns.append(np.zeros(1, np.int64))
            gs.append(np.asarray([self.bn_g_off[pre]["slope"]]))
        p = np.concatenate([a.reshape(-1) for a in ps]).astype(np.int64)
        g = np.concatenate([a.reshape(-1) for a in gs]).astype(np.int64)
        n = np.concatenate([a.reshape(-1) for a in ns]).astype(np.int64)
        order = np.argsort(p, kind="stable")
        p, g, n = p[order], g[order], n[order]
        first = np.searchsorted(p, p, side="left")
        slot = np.arange(p.shape[0]) - first
        assert slot.max() < 4, "a parameter feeds more than 4 packed-gradient entries"
        tab = np.full((L.n_params, 4), -1, dtype=np.int32)
        tab[p, slot] = ((g << 1) | n).astype(np.int32)
        return tab


def gather_ordered_unpack_table(tab, tensor_offsets):
    """The un-pack table [n_params][4] with the rows of every tensor sorted by the address of their first packed entry, and the
    parameter each row un-packs (int32 [n_params]): the form sehip_unpack_grad_sums_perm takes.  A convolution weight
    [co][ci][kf][kt] reads dW[n][(kt, kf, ci)]: in parameter order neighbouring lanes gather floats 5 C apart (one 64-byte sector per
    4-byte read), in gather order they read a run of `ci`.  Rows never leave their tensor: the per-tensor sums are taken by position."""
    tab = np.asarray(tab)
    n = tab.shape[0]
    first = tab[:, 0].astype(np.int64) >> 1
    first[tab[:, 0] < 0] = np.iinfo(np.int64).max >> 2          # parameters without a packed entry stay where they are, at the end
    offs = np.asarray(tensor_offsets, dtype=np.int64)
    tensor_of = np.searchsorted(offs, np.arange(n, dtype=np.int64), side="right") - 1
    perm = np.lexsort((np.arange(n), first, tensor_of)).astype(np.int32)     # by tensor, then by gather address, stable
    assert np.array_equal(tensor_of[perm], tensor_of)
    return np.ascontiguousarray(tab[perm]), perm


# --------------------------------------------------------------------------------------------------
# dynamic part: buffers for one (batch, length), bound descriptors, launch sequences
# --------------------------------------------------------------------------------------------------
class Buf:
    def __init__(self, t, tst, f, c, t0):
        self.t, self.Tst, self.F, self.C, self.t0 = t, tst, f, c, t0

    @property
    def ptr(self):
        return self.t.data_ptr()


class DeviceTables:
    """Device copies of the static tables (shared by every workspace of a model on one device)."""

    def __init__(self, st: DCCRNStatic, device):
        f = lambda a: torch.from_numpy(a).to(device)
        self.wtab, self.btab, self.utab = f(st.wtab), f(st.btab), f(st.utab)
        self.ktab, self.ntab = f(st.ktab), f(st.ntab)
        self.tensor_offsets = f(st.layout.tensor_offsets)
        self.utab_g = self.uperm = None                 # the fused tail's un-pack in gather order (SEHIP_NO_UNPACK_PERM: parameter order)
        if not os.environ.get("SEHIP_NO_UNPACK_PERM"):
            tg, pm = gather_ordered_unpack_table(st.utab, st.layout.tensor_offsets)
            self.utab_g, self.uperm = f(tg), f(pm)
        cfg = st.cfg
        self.window = f(ops.window_of(cfg.win_type, cfg.win_len))
        self.wpack = torch.zeros(st.n_wpack, dtype=BF16, device=device)
        self.bpack = torch.zeros(max(st.n_bpack, 4), dtype=torch.float32, device=device)


class DCCRNWorkspace:
    def __init__(self, st: DCCRNStatic, tables: DeviceTables, B, N, device):
        cfg = st.cfg
        self.st, self.tb, self.B, self.N, self.device = st, tables, B, N, device
        self.generation = 0     # bumped by every forward (a backward checks that its activations are still the live ones)
        self.pinned = False     # a captured hipGraph holds raw pointers into this workspace: never evict
        self.closed = False
        self.T = T = ops.stft_frames(N, cfg.win_len, cfg.win_inc)
        if T < 1:
            raise SehipError(f"input of {N} samples is shorter than one frame")
        max_len = (T - 1) * cfg.win_inc + cfg.win_len - (cfg.win_len - cfg.win_inc)
        self.length = min(cfg.length, max_len)  # the reference's [..., :length] slice simply truncates
        kn = cfg.kernel_num
        F0 = st.F0
        self.bufs = {}

        Bp = round_up(B, 16)  # gate / cell-state records of the LSTM kernels are stored per 16-row batch tile

        def add(name, tst, f, c, t0=0, dtype=BF16, lead=None, batch=None):
            bb = B if batch is None else batch
            shape = (bb, tst, f, c) if lead is None else (lead, bb, tst, f, c)
            t = torch.zeros(shape, dtype=dtype, device=device)
            if lead is None:
                self.bufs[name] = Buf(t, tst, f, c, t0)
            else:
                for q in range(lead):
                    self.bufs[f"{name}_{q}"] = Buf(t[q], tst, f, c, t0)
                self.bufs[name] = Buf(t, tst, f, c, t0)
            return t

        add("enc_in", T, F0, 2)
        for i in range(6):
            f, c = F0 >> (i + 1), kn[i + 1]
            add(f"y{i}", T, f, c); add(f"z{i}", T, f, c)
            add(f"dye{i}", T, f, c); add(f"dskip{i}", T, f, c)
            if i < 5:
                add(f"dz{i}", T, f, c)
        c5, h = kn[6], cfg.hid
        add("dz5l", T, 4, c5); add("P", T, 4, c5); add("dP", T, 4, c5)
        L = cfg.rnn_layers
        if not cfg.use_clstm:        # one real recurrence per layer (sehip_rlstm_fwd / _bwd): no (real, imaginary) x (real_lstm, imag_lstm) combos
            for layer in (1, 2):
                add(f"pre{layer}", T, 1, 4 * h, dtype=torch.float32)
                add(f"dpre{layer}", T, 1, 4 * h)
                add(f"h{layer}", T, 1, h, lead=1)
                add(f"gates{layer}", T, 1, 4 * h, lead=1, batch=Bp)
                add(f"c{layer}", T, 1, h, dtype=torch.float32, lead=1, batch=Bp)
            add("dxo", T, 1, h); add("dx2", T, 1, h)
        for layer in range(1, L + 1 if cfg.use_clstm else 0):
            for tag in "ri":
                add(f"pre{layer}_{tag}", T, 1, 8 * h, dtype=torch.float32)
                add(f"dpre{layer}_{tag}", T, 1, 8 * h)
            add(f"h{layer}", T, 1, h, lead=4)
            add(f"gates{layer}", T, 1, 4 * h, lead=4, batch=Bp)
            add(f"c{layer}", T, 1, h, dtype=torch.float32, lead=4, batch=Bp)
        for tag in ("ri" if cfg.use_clstm else ""):
            add(f"dxo_{tag}", T, 1, h)
            for layer in range(2, L + 1):
                add(f"dx{layer}_{tag}", T, 1, h)
        for j in range(5):
            idx = 6 - j
            f, c = (F0 >> idx) * 2, kn[idx - 1]
            add(f"yd{j}", T + 1, f, c, 1); add(f"zd{j}", T + 1, f, c, 1)
            add(f"dyd{j}", T + 1, f, c, 1); add(f"dzd{j}", T + 1, f, c, 1)
            if j in st.dec_split:
                add(f"pd{j}", T + 1, f, c, 1)      # the skip-connection half of the layer's forward product (bf16 partial sums)
        add("mask", T, F0, 2, dtype=torch.float32)
        add("dmask", T, F0, 2)
        self.spec = torch.empty(B, T, 257, 2, dtype=torch.float32, device=device)
        self.frames = torch.empty(B, T, cfg.win_len, dtype=torch.float32, device=device)
        self.wav = torch.empty(B, self.length, dtype=torch.float32, device=device)
        self.inv_coff = torch.from_numpy(ops.inv_window_energy(cfg.win_len, cfg.win_inc, T, self.length, cfg.win_type)).to(device)
        self.gpack = torch.zeros(st.n_gpack, dtype=torch.float32, device=device)
        maxcr = max(cr for _, cr in st.bn)
        lib = _lib.lib()
        need = 16
        for pre, cr in st.bn:
            src = self.bufs[("y" if pre.startswith("encoder") else "yd") + pre.split(".")[1]]
            need = max(need, int(lib.sehip_cbn_scratch_floats(src.t.numel() // (2 * cr), cr)))
        self.bn_acc = torch.zeros(need, dtype=torch.float32, device=device)
        self.bn_coef = {pre: torch.zeros(cr, 16, dtype=torch.float32, device=device) for pre, cr in st.bn}
        # [8 replicas][5][Cr] sums per fused layer (sehip_gemm_desc.stats), one allocation so that one memset clears them all
        offs, tot = {}, 0
        for pre, cr in st.bn:
            if pre in st.fused_stats or pre in st.small_stats:
                offs[pre] = tot
                tot += 8 * 5 * cr
        self.bn_stats_all = torch.zeros(max(tot, 1), dtype=torch.float32, device=device)
        # two sets of [BWD_REPLICAS][6 Cr + 1] sums of the backward reduce pass per layer (sehip_cbn_bwd_fused: a call adds to one set
        # and clears the other for the next call)
        self.fuse_finalize = not os.environ.get("SEHIP_NO_FUSE_FINALIZE")
        # (the same for the backward pass -- sehip_cbn_bwd_fused: the reduce pass adds its block sums to BWD_REPLICAS rows with atomics,
        #  the apply pass finalizes them itself.  Round 3 measured no gain (B = 32 step 4.129 ms with 8 rows of sums, 4.159 with 16,
        #  against 4.131 without it) and left it opt-in; in round 6, with the chain 1 ms shorter and the eleven cbn_bwd_finalize
        #  launches 0.10 ms of it, the same code is 3.142 against 3.213 / 3.204 ms (same box, twice): default.  Not in the
        #  deterministic schedule (atomics in a varying order).  SEHIP_NO_FUSE_BWD_FINALIZE=1: the three launches.)
        self.fuse_bwd_finalize = (self.fuse_finalize and not st.deterministic and bool(os.environ.get("SEHIP_FUSE_BWD_FINALIZE")))
        # round 6, second form: the finalize step inside the reduce launch's LAST workgroup (sehip_cbn_bwd_reduce_fin) + the plain apply
        # pass -- two launches like the fused form, but ONE workgroup finalizes and the records travel through global memory as in the
        # three-launch form.  Correct on every box (finite, 46 oracle tests green) and SLOWER: 3.238 / 3.219 against 3.156 / 3.160 ms (the
        # release fence + ticket of 512 workgroups and one workgroup's serial finalize cost more than eleven 9-us launches that overlap
        # with the weight-gradient queue anyway).  SEHIP_BN_REDUCE_FIN=1
        self.bn_reduce_fin = (self.fuse_finalize and not st.deterministic and not self.fuse_bwd_finalize
                              and bool(os.environ.get("SEHIP_BN_REDUCE_FIN")))
        self.bn_ticket = torch.zeros(len(st.bn) + 1, dtype=torch.int32, device=device)
        self._bn_index = {pre: i for i, (pre, _) in enumerate(st.bn)}
        # the two decoder layers whose reduce pass rides in the launch that produces their dz (bnr_rows) keep finalize_n + apply unless
        # SEHIP_FUSE_BWD_ALL=1 (then their rows are written and ignored)
        self.fuse_bwd_all = bool(os.environ.get("SEHIP_FUSE_BWD_ALL"))
        self.bn_brep = {pre: torch.zeros(2, BWD_REPLICAS * (6 * cr + 1), dtype=torch.float32, device=device) for pre, cr in st.bn}
        self._brep_turn = {pre: 0 for pre, _ in st.bn}
        self.bn_stats = {pre: self.bn_stats_all[o:o + 8 * 5 * cr] for (pre, cr) in st.bn if pre in offs for o in [offs[pre]]}
        self.bn_bcoef = torch.zeros(maxcr, 16, dtype=torch.float32, device=device)
        # use_cbn=False: the kernels' eps < 0 convention (csrc/cbn.hip), a zero row for the cross weight, a sink for the cross variance
        self.bn_eps = 1e-5 if cfg.use_cbn else -1e-5
        self.bn_zero = torch.zeros(maxcr, dtype=torch.float32, device=device)
        self.bn_dummy = torch.zeros(maxcr, dtype=torch.float32, device=device)
        self._bn_cr = {pre: cr for pre, cr in st.bn}
        self.mode = {"E": 0, "C": 1, "R": 2}[cfg.masking_mode]
        # the weight-gradient stream.  SEHIP_SIDE_PRIORITY=1: created through the C ABI with the device's lowest priority
        self.side, self._side_handle = None, None
        if not os.environ.get("SEHIP_NO_SIDE_STREAM"):
            if os.environ.get("SEHIP_SIDE_PRIORITY"):
                with torch.cuda.device(device):
                    h = _lib.lib().sehip_stream_create(int(os.environ["SEHIP_SIDE_PRIORITY"]))
                if not h:
                    raise SehipError("sehip_stream_create: " + _lib.lib().sehip_last_error().decode())
                self._side_handle = h
                self.side = torch.cuda.ExternalStream(h, device=device)
            else:
                self.side = torch.cuda.Stream(device=device)
        self._events, self._event_i, self._chain_dirty = [], 0, True
        self._fs_events = []
        self.comm = None     # third stream: early un-pack + all-reduce of the decoder / LSTM gradients (data-parallel runs only)
        # The two stacked complex LSTM layers are pipelined over chunks of time steps: layer 2 (and the input product that
        # feeds it) runs chunk c on a second high-priority stream while layer 1 runs chunk c+1 (backward: the other way
        # round).  Each recurrence keeps 8 workgroups busy for ~0.9 us per step, so two of them side by side cost nothing.
        # Measured at T = 321 (ms per step): whole sequence 6.19-6.25, chunks of 64: 6.12-6.17, of 108 (three chunks): 6.09.
        # Every chunk costs a kernel prologue, an event and two small products on the second stream, so few chunks win.
        # Round 2: with 4-row batch tiles and an 8-step input prefetch a layer takes 131 / 183 us instead of 290, and the
        # pipeline no longer pays for its extra launches, events and chunked products (T = 323, ms per step: whole sequence
        # 4.61, three chunks 4.63, four 4.65, six 4.70): the default is the whole sequence, SEHIP_LSTM_CHUNK=<steps> re-enables it.
        chunk = int(os.environ.get("SEHIP_LSTM_CHUNK", "0"))
        if chunk < 0:
            chunk = (T + 2) // 3 if T >= 96 else 0
        self.lstm_chunks = [(t, min(T, t + chunk)) for t in range(0, T, chunk)] if chunk > 0 else [(0, T)]
        if len(self.lstm_chunks) > 1 and self.lstm_chunks[-1][1] - self.lstm_chunks[-1][0] < chunk // 4:
            last = self.lstm_chunks.pop()            # no tiny trailing chunk
            self.lstm_chunks[-1] = (self.lstm_chunks[-1][0], last[1])
        self.lstm_stream = self.lstm_gemm_stream = None
        if len(self.lstm_chunks) > 1 and not os.environ.get("SEHIP_NO_SIDE_STREAM"):
            hi = torch.cuda.Stream.priority_range()[1]
            self.lstm_stream = torch.cuda.Stream(device=device, priority=hi)
            # the small products between the layers get their own stream: chunk c's product overlaps layer 2's chunk c-1
            self.lstm_gemm_stream = torch.cuda.Stream(device=device, priority=hi)
        self.lstm_state = {layer: torch.zeros(4 * ((B + 3) // 4) * 4 * st.cfg.hid * 2, dtype=torch.float32, device=device)
                           for layer in range(1, st.cfg.rnn_layers + 1)}
        # fused two-layer recurrence (csrc/lstm2.hip, round 4): both layers in ONE persistent launch per direction, layer 2 a dozen steps
        # behind layer 1, hand-off by data-tagged granules.  Default when the whole sequence runs as one chunk; SEHIP_NO_LSTM_FUSE or a
        # hand-off time-out (check_lstm_handoffs) return to the two launches per direction of round 3.
        # (rnn_layers != 2 or rnn_units != 128: one launch per layer and direction, the products between them on the chain)
        self.lstm_fused = (len(self.lstm_chunks) == 1 and not os.environ.get("SEHIP_NO_LSTM_FUSE") and T < 65535
                           and st.cfg.rnn_layers == 2 and st.cfg.hid == 64 and st.cfg.use_clstm)
        self.l2_epoch = 0
        self.graph_epoch = 0         # bumped when captured launches of this workspace go stale (fall-back after a hand-off time-out)
        if self.lstm_fused:
            lib_ = _lib.lib()
            self.l2_gran_f = torch.zeros(int(lib_.sehip_lstm2_gran_bytes(B, T, 0)) // 8, dtype=torch.int64, device=device)
            self.l2_gran_b = torch.zeros(int(lib_.sehip_lstm2_gran_bytes(B, T, 1)) // 8, dtype=torch.int64, device=device)
            self.l2_sync = torch.zeros(int(lib_.sehip_lstm2_sync_bytes()) // 4, dtype=torch.int32, device=device)
        # (two-launch BatchNorm backward -- SEHIP_FUSE_BWD_FINALIZE -- leaves no record array to read)
        # (the kernel that can: 16 outputs, output rows a multiple of 32 and at most 128 -- csrc/gemm.hip try_narrow_wgrad)
        self.enc0_bn_in_wgrad = (FUSE_ENC0_BN_WGRAD and not self.fuse_bwd_finalize and int(st.cfg.kernel_num[1]) == 16
                                 and (st.F0 >> 1) % 32 == 0 and (st.F0 >> 1) <= 128)
        self._bind()

    def close(self):
        """Destroys the HIP events of this workspace (the tensors go with the Python object)."""
        if self.closed:
            return
        self.closed = True
        lib = _lib.lib()
        for e in self._events + getattr(self, "_fs_events", []):
            lib.sehip_event_destroy(e)
        self._events, self._fs_events = [], []
        if self._side_handle:
            lib.sehip_stream_destroy(self._side_handle)
            self._side_handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- descriptors ---------------------------------------------------------------------------
    def _bind(self):
        st, tb, B, T = self.st, self.tb, self.B, self.T
        self.desc = {}
        # the LAST weight gradient of the pass (enc0: narrow_wgrad_mfma, 28 us) runs on the chain's own stream: nothing is left to
        # overlap it with, and two event hops less sit in front of the un-pack (4.125-4.131 against 4.132-4.137 ms; enc1 too: 4.16).
        # SEHIP_WGRAD_ON_CHAIN=<names> / none
        self._wgrad_on_chain = set((os.environ.get("SEHIP_WGRAD_ON_CHAIN", "enc0.fwd") or "").split(",")) - {"", "none"}
        self._chunk_cache = {}
        self._wg_groups = {}
        # chunk table bound to this workspace's source geometry: [src, (toff<<16)|(fadd&0xffff), element delta, npieces]
        kt = st.ktab.copy()
        for name, s in st.specs.items():
            bind_chunk_table(st.ktab, kt, s.kt_off, s.K // 8, [(self.bufs[b].F, self.bufs[b].C) for b, _ in s.srcs])
        self.ktab_dev = torch.from_numpy(kt).to(self.device)
        for name, s in st.specs.items():
            d = CGemmDesc()
            tt = T if s.tt == "T" else T + 1
            for q, (bname, mode) in enumerate(s.srcs):
                b = self.bufs[bname]
                d.src[q].ptr = b.ptr
                d.src[q].T, d.src[q].F, d.src[q].C = b.Tst, b.F, b.C
                if mode == "drop1":
                    d.src[q].tlo, d.src[q].thi = 1, T + 1
                else:
                    d.src[q].tlo, d.src[q].thi = 0, b.Tst
            for q, (bname, toff, fmul, fadd) in enumerate(s.dsts):
                b = self.bufs[bname]
                d.dst[q].ptr = b.ptr
                d.dst[q].T, d.dst[q].F, d.dst[q].C = b.Tst, b.F, b.C
                d.dst[q].toff, d.dst[q].fmul, d.dst[q].fadd = toff, fmul, fadd
                d.dst[q].is_f32 = 1 if b.t.dtype == torch.float32 else 0
                if s.J == 1:  # dense rows: view the destination as one row per (b,t)
                    d.dst[q].F, d.dst[q].C = 1, b.F * b.C
            d.ktab = self.ktab_dev.data_ptr() + 16 * s.kt_off
            d.ntab = tb.ntab.data_ptr() + 16 * s.nt_off
            if s.w_off is not None:
                d.W = tb.wpack.data_ptr() + 2 * s.w_off
            if s.b_off is not None:
                d.bias = tb.bpack.data_ptr() + 4 * s.b_off
            d.M, d.N, d.Npad, d.K = B * tt * s.J, s.N, s.Npad, s.K
            d.TT, d.J, d.fmul = tt, s.J, s.fmul
            d.w_tiled = 1 if s.w_tiled else 0
            if s.stats_of is not None:
                d.stats = self.bn_stats[s.stats_of].data_ptr()
                d.stats_cr = s.N // 2
            if s.res is not None:
                rb, db_ = self.bufs[s.res], self.bufs[s.dsts[0][0]]
                assert (rb.Tst, rb.F, rb.C) == (db_.Tst, db_.F, db_.C) and rb.t.dtype == torch.bfloat16
                d.res = rb.ptr
            if s.conv is not None:
                d.cv_nf, d.cv_fadd = s.conv[0], s.conv[1]
                for q in range(2):
                    for kt in range(2):
                        d.cv_toff[q][kt] = s.conv[2][q][kt]
            self.desc[name] = d
            if s.dw_off is not None:  # weight-gradient twin: dOut replaces the destination
                w = CGemmDesc.from_buffer_copy(d)
                w.dW = self.gpack.data_ptr() + 4 * s.dw_off
                w.dbias = self.gpack.data_ptr() + 4 * s.db_off if s.db_off is not None else None
                if s.kind == "fwd":
                    gname = self._grad_buffer_of(s.dsts[0][0])
                    gb = self.bufs[gname]
                    w.dst[0].ptr = gb.ptr
                    w.dst[0].is_f32 = 0
                if name == "enc0.fwd" and self.enc0_bn_in_wgrad:
                    zb, yb, gb = self.bufs["dz0"], self.bufs["y0"], self.bufs["dye0"]
                    assert (zb.Tst, zb.F, zb.C) == (gb.Tst, gb.F, gb.C) == (yb.Tst, yb.F, yb.C) and zb.t.dtype == yb.t.dtype == BF16
                    w.bn_dz, w.bn_y = zb.ptr, yb.ptr
                    w.bn_coef, w.bn_bcoef = ptr(self.bn_coef["encoder.0."]), ptr(self.bn_bcoef)     # (bn_slope: set per call, from params)
                self.desc[name + ".wg"] = w
        # 32-output layers: fused BatchNorm sums if the small-channel kernel takes the forward product(s) as the step launches them
        # (encoder: one product; decoder: the pair of output-row parities)
        self.fused_small = set()
        lib = _lib.lib()
        for pre, names in st.small_stats.items():
            ds = [self.desc[nm] for nm in names]
            for d in ds:
                d.stats = self.bn_stats[pre].data_ptr()
                d.stats_cr = d.Npad // 2
            if lib.sehip_conv_small_takes(C.byref(ds[0]), C.byref(ds[1]) if len(ds) == 2 else None):
                self.fused_small.add(pre)
            else:
                for d in ds:
                    d.stats = None
                    d.stats_cr = 0
        # the BatchNorm backward REDUCE pass of a decoder layer inside the streaming launch that produces its activation gradient
        # (sehip_gemm_desc.bnr_*, csrc/convt.hip): dec{j}.dg stores dzd{j-1} and leaves one row of sums per workgroup in bn_acc,
        # sehip_cbn_bwd_finalize_n adds them.  The library says whether (and with how many rows) it does this for the product.
        self.bnr_rows = {}
        if not os.environ.get("SEHIP_NO_BNR"):
            for j in range(1, 6):
                pre, cr = f"decoder.{j - 1}.", int(st.cfg.kernel_num[6 - j]) // 2
                d = self.desc[f"dec{j}.dg"]
                rows = int(lib.sehip_bnr_rows(C.byref(d), None))
                yb, zb = self.bufs[f"yd{j - 1}"], self.bufs[f"dzd{j - 1}"]
                if rows <= 0 or d.dst[0].ptr != zb.ptr or (yb.Tst, yb.F, yb.C) != (zb.Tst, zb.F, zb.C) or yb.C != 2 * cr:
                    continue
                if rows * (6 * cr + 1) > self.bn_acc.numel():
                    self.bn_acc = torch.zeros(rows * (6 * cr + 1), dtype=torch.float32, device=self.device)
                self.bnr_rows[pre] = (rows, f"dec{j}.dg")
            for pre, (rows, name) in self.bnr_rows.items():
                d = self.desc[name]
                d.bnr_y, d.bnr_coef, d.bnr_part = self.bufs["yd" + pre.split(".")[1]].ptr, ptr(self.bn_coef[pre]), ptr(self.bn_acc)
                # (bnr_slope: set per call, from params)
        if not os.environ.get("SEHIP_NO_WGRAD_GROUP"):
            nl = self.st.cfg.rnn_layers
            # (a group holds at most 16 products, six per layer: the whole stack in one launch for two layers only)
            for layers in [(layer,) for layer in range(1, nl + 1)] + ([(2, 1)] if nl == 2 else []):
                self._wgrad_group_handle(self._lstm_wgrad_names(layers))

    def _chunk_desc(self, name, t0, t1):
        """Copy of a dense (J == 1) descriptor restricted to the frames [t0, t1): TT and M shrink, every source /
        destination pointer moves t0 frames forward (the storage strides keep describing the whole buffer)."""
        key = (name, t0, t1)
        if key in self._chunk_cache:
            return self._chunk_cache[key]
        d = CGemmDesc.from_buffer_copy(self.desc[name])
        s = self.st.specs[name]
        assert s.J == 1 and s.tt == "T"
        for q, (bname, mode) in enumerate(s.srcs):
            b = self.bufs[bname]
            d.src[q].ptr = b.ptr + t0 * b.F * b.C * b.t.element_size()
            d.src[q].thi = b.Tst - t0
        for q, (bname, toff, fmul, fadd) in enumerate(s.dsts):
            b = self.bufs[bname]
            d.dst[q].ptr = b.ptr + t0 * b.F * b.C * b.t.element_size()
        if s.res is not None:      # laid out like dst[0]
            rb = self.bufs[s.res]
            d.res = rb.ptr + t0 * rb.F * rb.C * rb.t.element_size()
        d.TT = t1 - t0
        d.M = self.B * (t1 - t0)
        self._chunk_cache[key] = d
        return d

    @staticmethod
    def _grad_buffer_of(out_name):
        if out_name.startswith("yd"):
            return "dyd" + out_name[2:]
        if out_name.startswith("y"):
            return "dye" + out_name[1:]
        if out_name == "mask":
            return "dmask"
        if out_name.startswith("pre"):
            return "d" + out_name
        if out_name == "P":
            return "dP"
        raise KeyError(out_name)

    def gemm(self, name):
        self._chain_dirty = True
        call("sehip_gemm", C.byref(self.desc[name]), stream())

    def gemm_pair(self, a, b):
        """Two products over the same sources (the two output-row parities of a transposed convolution): one launch that
        stages the input once where the library can fuse them."""
        self._chain_dirty = True
        call("sehip_gemm_pair", C.byref(self.desc[a]), C.byref(self.desc[b]), stream())

    def _event(self):
        """Round-robin pool of fence-free events (sehip_stream_depend)."""
        if not self._events:
            for _ in range(32):
                e = _lib.lib().sehip_event_create()
                if not e:
                    raise SehipError("sehip_event_create: " + _lib.lib().sehip_last_error().decode())
                self._events.append(e)
        self._event_i = (self._event_i + 1) % len(self._events)
        return self._events[self._event_i]

    def wgrad(self, name):
        """Weight gradients are side work (nothing in the backward chain consumes them): they go to a second HIP stream
        and fill the ~248 CUs the persistent LSTM kernels (8 workgroups) and the latency-bound BatchNorm passes leave
        idle.  The side stream waits for everything enqueued so far on the main stream (which includes the producer of
        dOut) -- once per group of weight gradients: nothing new has been put on the chain between dec.fwd0 / fwd1 or the
        four hh products, and every event record costs the chain a bubble; backward() joins the two streams before the
        gradients are un-packed."""
        main = torch.cuda.current_stream()
        if self.side is None or name in self._wgrad_on_chain:
            # (SEHIP_WGRAD_ON_CHAIN=enc0.fwd,...: named weight gradients on the chain's own stream.  With the 68-us VALU kernel for
            #  enc0 this was no gain (4.23-4.29 against 4.23-4.25 ms for the last one / two / three of the pass); with
            #  narrow_wgrad_mfma the last one is worth 6 us and is the default)
            call("sehip_wgrad", C.byref(self.desc[name + ".wg"]), main.cuda_stream)
            return
        if self._chain_dirty:
            call("sehip_stream_depend", self.side.cuda_stream, main.cuda_stream, self._event())
            self._chain_dirty = False
        call("sehip_wgrad", C.byref(self.desc[name + ".wg"]), self.side.cuda_stream)

    def wgrad_pair(self, a, b):
        """The weight gradients of the two output-row parities of a transposed convolution: one streaming launch for the outer layers
        (sehip_wgrad_pair, csrc/convt.hip), else the two launches."""
        main = torch.cuda.current_stream()
        if self.side is None or a in self._wgrad_on_chain:
            call("sehip_wgrad_pair", C.byref(self.desc[a + ".wg"]), C.byref(self.desc[b + ".wg"]), main.cuda_stream)
            return
        if self._chain_dirty:
            call("sehip_stream_depend", self.side.cuda_stream, main.cuda_stream, self._event())
            self._chain_dirty = False
        call("sehip_wgrad_pair", C.byref(self.desc[a + ".wg"]), C.byref(self.desc[b + ".wg"]), self.side.cuda_stream)

    def _lstm_wgrad_names(self, layers):
        if not self.st.cfg.use_clstm:
            return [nm for layer in layers for nm in (f"ih{layer}", f"hh{layer}")]
        return [nm for layer in layers for nm in [f"ih{layer}_{tag}" for tag in "ri"] + [f"hh{layer}_{combo}" for combo in range(4)]]

    def _wgrad_group_handle(self, names):
        """Device image (descriptors + block table) of a grouped weight-gradient launch; built at bind time, because
        sehip_wgrad_group_prepare copies synchronously and must not run inside a stream capture."""
        key = tuple(names)
        g = self._wg_groups.get(key)
        if g is None:
            n = len(names)
            arr = (CGemmDesc * n)(*[CGemmDesc.from_buffer_copy(self.desc[nm + ".wg"]) for nm in names])
            buf = torch.empty(int(_lib.lib().sehip_wgrad_group_bytes(n)), dtype=torch.uint8, device=self.bufs["enc_in"].t.device)
            total = C.c_int(0)
            call("sehip_wgrad_group_prepare", C.cast(arr, C.c_void_p), n, ptr(buf), C.cast(C.pointer(total), C.c_void_p))
            # the same group on the streaming dense-row kernel (csrc/dtw.hip) where the library takes it: its tables and the scratch
            # its partial tiles go through (caller-owned: nothing is allocated per step; shared by the groups of this workspace --
            # they run on one stream, one after the other)
            dense = None
            if not os.environ.get("SEHIP_NO_DENSE_GROUP"):
                lib_ = _lib.lib()
                dbuf = torch.empty(int(lib_.sehip_wgrad_dense_group_bytes(n)), dtype=torch.uint8, device=buf.device)
                info = (C.c_int * 8)()
                call("sehip_wgrad_dense_group_prepare", C.cast(arr, C.c_void_p), n, ptr(dbuf), dbuf.numel(), C.cast(info, C.c_void_p))
                if info[0] == 1:
                    need = info[4] + (info[5] << 31)
                    if getattr(self, "_dtw_scratch", None) is None or self._dtw_scratch.numel() < need:
                        self._dtw_scratch = torch.empty(need, dtype=torch.float32, device=buf.device)
                    dense = (dbuf, info)
            g = self._wg_groups[key] = (buf, n, total.value, dense)
        return g

    def wgrad_group(self, names):
        """The weight gradients of several plain products as ONE launch on the side stream (sehip_wgrad_group): the LSTM
        products are small grids that took ~30 us each back to back."""
        if os.environ.get("SEHIP_NO_WGRAD_GROUP"):
            for nm in names:
                self.wgrad(nm)
            return
        buf, n, total, dense = self._wgrad_group_handle(names)
        if self.st.deterministic and dense is None:      # (the streaming dense-row launch adds its splits in a fixed order; the
            for nm in names:                             #  table-gathered grouped launch uses atomics: not in the deterministic schedule)
                self.wgrad(nm)
            return
        main = torch.cuda.current_stream()

        def launch(st):
            if dense is not None:
                call("sehip_wgrad_dense_group", ptr(dense[0]), n, C.cast(dense[1], C.c_void_p), ptr(self._dtw_scratch), st)
            else:
                call("sehip_wgrad_group", ptr(buf), n, total, st)
        if self.side is None:
            launch(main.cuda_stream)
            return
        if self._chain_dirty:
            call("sehip_stream_depend", self.side.cuda_stream, main.cuda_stream, self._event())
            self._chain_dirty = False
        launch(self.side.cuda_stream)

    def launch_units(self):
        """The product launches of one train step AS THE STEP ISSUES THEM (bench.py's per-kernel roofline pass times these, not the
        descriptors one by one): single products, the pairs that go through sehip_gemm_pair / sehip_wgrad_pair (one launch where the
        library merges them) and the grouped LSTM weight gradients.  [(label, names, fn(stream))]."""
        d, units = self.desc, []

        def one(fn, name):
            return (name, [name], lambda st, fn=fn, name=name: call(fn, C.byref(d[name]), st))

        def pair(fn, a, b):
            return (a + "+" + b, [a, b], lambda st, fn=fn, a=a, b=b: call(fn, C.byref(d[a]), C.byref(d[b]), st))
        for i in range(6):
            units.append(one("sehip_gemm", f"enc{i}.fwd"))
            units.append(one("sehip_wgrad", f"enc{i}.fwd.wg"))
            units.append(one("sehip_gemm", f"dec{i}.dg"))
            if i in self.st.dec_split and self.side is not None:      # (forward: the skip half under the LSTM + the main half)
                units.append(pair("sehip_gemm_pair", f"dec{i}.fs0", f"dec{i}.fs1"))
                units.append(pair("sehip_gemm_pair", f"dec{i}.fm0", f"dec{i}.fm1"))
            else:
                units.append(pair("sehip_gemm_pair", f"dec{i}.fwd0", f"dec{i}.fwd1"))
            units.append(pair("sehip_wgrad_pair", f"dec{i}.fwd0.wg", f"dec{i}.fwd1.wg"))
            if i > 0:
                units.append(pair("sehip_gemm_pair", f"enc{i}.dg0", f"enc{i}.dg1"))
        if not self.st.cfg.use_clstm:         # one real nn.LSTM: single products, per-layer weight-gradient groups
            for nm in ("ih1", "ih2", "proj", "dproj", "dx2", "dx1"):
                units.append(one("sehip_gemm", nm))
            units += [one("sehip_wgrad", nm + ".wg") for nm in ["proj"] + self._lstm_wgrad_names((2, 1))]
            return units
        for a, b in (("ih1_r", "ih1_i"), ("proj_r", "proj_i"), ("dproj_r", "dproj_i"), ("dx1_r", "dx1_i")):
            units.append(pair("sehip_gemm_pair", a, b))
        nl = self.st.cfg.rnn_layers
        if not self.lstm_fused:
            for layer in range(2, nl + 1):
                for a, b in ((f"ih{layer}_r", f"ih{layer}_i"), (f"dx{layer}_r", f"dx{layer}_i")):
                    units.append(pair("sehip_gemm_pair", a, b))
        for tag in "ri":
            units.append(one("sehip_wgrad", f"proj_{tag}.wg"))
        names = self._lstm_wgrad_names(tuple(range(nl, 0, -1)))
        if not os.environ.get("SEHIP_NO_WGRAD_GROUP") and nl <= 2:
            buf, n, total, dense = self._wgrad_group_handle(names)
            if dense is not None:
                units.append(("lstm.wg (dense group)", [nm + ".wg" for nm in names],
                              lambda st: call("sehip_wgrad_dense_group", ptr(dense[0]), n, C.cast(dense[1], C.c_void_p), ptr(self._dtw_scratch), st)))
            elif not self.st.deterministic:
                units.append(("lstm.wg (group)", [nm + ".wg" for nm in names], lambda st: call("sehip_wgrad_group", ptr(buf), n, total, st)))
            else:
                units += [one("sehip_wgrad", nm + ".wg") for nm in names]
        else:
            units += [one("sehip_wgrad", nm + ".wg") for nm in names]
        return units

    # ---- BatchNorm helpers ---------------------------------------------------------------------
    def _bn_ptrs(self, pre, params, buffers, nbt):
        L = self.st.layout
        cr = self._bn_cr[pre]

        def pp(k):
            f = L.bn_field(pre, k, cr)
            return ptr(self.bn_zero) if f is None else params.data_ptr() + 4 * (L.param_off[f[0]][0] + f[1])

        def bp(k):
            f = L.bn_field(pre, k, cr)
            return ptr(self.bn_dummy) if f is None else buffers.data_ptr() + 4 * (L.buffer_off[f[0]][0] + f[1])
        return pp, bp, nbt.data_ptr() + 8 * L.nbt_idx[pre + "1.num_batches_tracked"]

    def bn_forward(self, pre, cr, y, z, params, buffers, nbt, training):
        rows = y.t.numel() // (2 * cr)
        pp, bp, nb = self._bn_ptrs(pre, params, buffers, nbt)
        coef = self.bn_coef[pre]
        if pre in self.st.fused_stats or pre in self.fused_small:     # the producing convolution accumulated the sums (8 replicas)
            if self.fuse_finalize:          # ... and the apply pass derives the coefficients itself: one launch
                call("sehip_cbn_finalize_apply_n", y.ptr, ptr(self.bn_stats[pre]), 8, pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), pp("1.Br"),
                     pp("1.Bi"), bp("1.RMr"), bp("1.RMi"), bp("1.RVrr"), bp("1.RVri"), bp("1.RVii"), nb, rows, cr, self.bn_eps, 0.1,
                     1 if training else 0, ptr(coef), pp("2.weight"), z.ptr, stream())
                return
            call("sehip_cbn_finalize_n", ptr(self.bn_stats[pre]), 8, pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), pp("1.Br"), pp("1.Bi"),
                 bp("1.RMr"), bp("1.RMi"), bp("1.RVrr"), bp("1.RVri"), bp("1.RVii"), nb, rows, cr, self.bn_eps, 0.1,
                 1 if training else 0, ptr(coef), stream())
            call("sehip_cbn_apply", y.ptr, ptr(coef), pp("2.weight"), rows, cr, z.ptr, stream())
            return
        if training:
            call("sehip_cbn_stats", y.ptr, rows, cr, ptr(self.bn_acc), stream())
        call("sehip_cbn_finalize", ptr(self.bn_acc), pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), pp("1.Br"), pp("1.Bi"),
             bp("1.RMr"), bp("1.RMi"), bp("1.RVrr"), bp("1.RVri"), bp("1.RVii"), nb, rows, cr, self.bn_eps, 0.1,
             1 if training else 0, ptr(coef), stream())
        call("sehip_cbn_apply", y.ptr, ptr(coef), pp("2.weight"), rows, cr, z.ptr, stream())

    def bn_backward(self, pre, cr, dz, dz2, y, dy, params, tfirst, apply=True):
        rows = y.t.numel() // (2 * cr)
        L, st = self.st.layout, self.st

        def pp(k):
            f = L.bn_field(pre, k, cr)
            return ptr(self.bn_zero) if f is None else params.data_ptr() + 4 * (L.param_off[f[0]][0] + f[1])
        g = lambda k: self.gpack.data_ptr() + 4 * st.bn_g_off[pre][k]
        coef = self.bn_coef[pre]
        dz2p = dz2.ptr if dz2 is not None else None
        self._chain_dirty = True
        if pre in self.bnr_rows and dz2 is None and not (self.fuse_bwd_finalize and self.fuse_bwd_all):
            # the launch that produced dz left the reduce pass's sums in bn_acc, one row per workgroup (bnr_rows)
            call("sehip_cbn_bwd_finalize_n", ptr(self.bn_acc), self.bnr_rows[pre][0], ptr(coef), pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), rows, cr,
                 g("Wrr"), g("Wri"), g("Wii"), g("Br"), g("Bi"), g("slope"), ptr(self.bn_bcoef), stream())
            if apply:
                call("sehip_cbn_bwd_apply", dz.ptr, dz2p, y.ptr, ptr(coef), ptr(self.bn_bcoef), pp("2.weight"), rows, cr, y.F,
                     y.Tst, tfirst, dy.ptr, stream())
            return
        if self.fuse_bwd_finalize:  # the reduce pass leaves its sums in a few rows, the apply pass finalizes them: two launches
            rep = self.bn_brep[pre]
            if torch.cuda.is_current_stream_capturing():
                # a captured step replays the same pointers every time: the alternation cannot be captured, so the graph clears
                # its set with a memset node of its own (and leaves the eager turn where it is)
                turn = 0
                rep[0].zero_()
            else:
                turn = self._brep_turn[pre]
                self._brep_turn[pre] = turn ^ 1
            call("sehip_cbn_bwd_fused", dz.ptr, dz2p, y.ptr, ptr(coef), pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), pp("2.weight"), rows, cr,
                 y.F, y.Tst, tfirst, ptr(rep[turn]), ptr(rep[turn ^ 1]), BWD_REPLICAS, g("Wrr"), g("Wri"), g("Wii"), g("Br"), g("Bi"),
                 g("slope"), dy.ptr, stream())
            return
        if self.bn_reduce_fin:
            rep = self.bn_brep[pre]
            call("sehip_cbn_bwd_reduce_fin", dz.ptr, dz2p, y.ptr, ptr(coef), pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), pp("2.weight"), rows, cr,
                 y.F, y.Tst, tfirst, ptr(rep[0]), BWD_REPLICAS, self.bn_ticket.data_ptr() + 4 * self._bn_index[pre], g("Wrr"), g("Wri"),
                 g("Wii"), g("Br"), g("Bi"), g("slope"), ptr(self.bn_bcoef), stream())
        else:
            call("sehip_cbn_bwd_reduce", dz.ptr, dz2p, y.ptr, ptr(coef), pp("2.weight"), rows, cr, y.F, y.Tst, tfirst,
                 ptr(self.bn_acc), stream())
            call("sehip_cbn_bwd_finalize", ptr(self.bn_acc), ptr(coef), pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), rows, cr,
                 g("Wrr"), g("Wri"), g("Wii"), g("Br"), g("Bi"), g("slope"), ptr(self.bn_bcoef), stream())
        if not apply:        # the consumer applies the records itself (enc0's weight gradient)
            return
        call("sehip_cbn_bwd_apply", dz.ptr, dz2p, y.ptr, ptr(coef), ptr(self.bn_bcoef), pp("2.weight"), rows, cr, y.F,
             y.Tst, tfirst, dy.ptr, stream())

    # ---- complex LSTM: two layers pipelined over time chunks ----------------------------------------
    def _lstm_fwd_call(self, layer, t0, t1, st_):
        b, cfg = self.bufs, self.st.cfg
        whh = self.tb.wpack.data_ptr() + 2 * self.st.whh_off[layer]
        call("sehip_lstm_fwd_chunk", b[f"pre{layer}_r"].ptr, b[f"pre{layer}_i"].ptr, whh, self.B, self.T, cfg.hid, t0, t1,
             b[f"h{layer}"].ptr, b[f"gates{layer}"].ptr, b[f"c{layer}"].ptr, st_)

    def _l2_next_epoch(self):
        """Epoch of the granule tags of one forward (+ its backward).  Eager: a new value per call, the granule arrays are never cleared.
        Under stream capture the value is frozen into the graph, so the graph clears the arrays itself (memset nodes)."""
        if torch.cuda.is_current_stream_capturing():
            # Epoch 65535 belongs to captured graphs: eager calls cycle through 1..65534, so the tags a replay leaves in the
            # arrays can never equal an eager call's, whatever the order of replays and eager forwards on this workspace.
            self.l2_gran_f.zero_(); self.l2_gran_b.zero_()
            return L2_GRAPH_EPOCH
        self.l2_epoch = self.l2_epoch % (L2_GRAPH_EPOCH - 1) + 1
        return self.l2_epoch

    def check_lstm_handoffs(self, recover=True, global_flag=None):
        """Reads the sticky time-out word of the fused recurrence (the read waits for the stream).  A time-out means the launches since
        then produced garbage; the fused optimizer did not apply it (the word is its device-side guard, model.step_guard()).
        recover=True: this workspace returns to one launch per layer and direction, the word is cleared, True is returned."""
        # (the word is read while it exists, fused or not: hipGraphs captured before a fall-back keep replaying the fused launches
        #  with this word as their optimizer guard -- ADVICE r4 -- until the Solver re-captures them, see graph_epoch)
        # (a workspace without the fused recurrence -- a chunked LSTM, T >= 65535 -- has no word: it still takes part in the
        #  collective with tripped = False, BEFORE any early return: the other ranks may hold fused workspaces -- ADVICE r5)
        has_word = hasattr(self, "l2_sync")
        tripped = has_word and int(self.l2_sync[0]) != 0
        if global_flag is not None:      # data parallel: every rank takes the same fall-back at the same health check
            tripped = global_flag(tripped, self.device)
        if not tripped:
            return False
        if not has_word:                 # another rank lost steps (they were skipped here too, through the global step guard)
            return True
        if not recover:
            raise SehipError("DCCRN: a hand-off wait of the fused two-layer LSTM kernels timed out (results since then are invalid and "
                             "no optimizer step was applied); set SEHIP_NO_LSTM_FUSE=1 to use one launch per layer")
        import warnings
        warnings.warn("sehip DCCRN: a hand-off wait of the fused two-layer LSTM kernels timed out; the optimizer steps since then were "
                      "skipped on the device; falling back to one launch per LSTM layer for this workspace")
        self.lstm_fused = False
        self.l2_sync.zero_()
        self.graph_epoch += 1        # graphs captured on this workspace still hold the fused launches: the owner re-captures
        return True

    def _lstm_forward(self, B, T, h):
        self._chain_dirty = True
        main = stream()
        if not self.st.cfg.use_clstm:
            b, wp = self.bufs, self.tb.wpack.data_ptr()
            for layer in (1, 2):
                self.gemm(f"ih{layer}")
                call("sehip_rlstm_fwd", b[f"pre{layer}"].ptr, wp + 2 * self.st.whh_off[layer], B, T, h, b[f"h{layer}"].ptr,
                     b[f"gates{layer}"].ptr, b[f"c{layer}"].ptr, main)
            return
        self.gemm_pair("ih1_r", "ih1_i")
        if self.lstm_fused:
            b, st, tb = self.bufs, self.st, self.tb
            wp = tb.wpack.data_ptr()
            self._l2_cur_epoch = self._l2_next_epoch()
            call("sehip_lstm2_fwd", b["pre1_r"].ptr, b["pre1_i"].ptr, wp + 2 * st.whh_off[1], wp + 2 * st.whh_off[2],
                 wp + 2 * st.wih2_off, self.desc["ih2_r"].bias, B, T, h, b["h1"].ptr, b["gates1"].ptr, b["c1"].ptr, b["h2"].ptr,
                 b["gates2"].ptr, b["c2"].ptr, ptr(self.l2_gran_f), ptr(self.l2_sync), self._l2_cur_epoch, main)
            return
        nl = self.st.cfg.rnn_layers
        if self.lstm_stream is None or nl != 2 or torch.cuda.is_current_stream_capturing():  # graph replay serialises the streams
            self._lstm_fwd_call(1, 0, T, main)
            for layer in range(2, nl + 1):
                self.gemm_pair(f"ih{layer}_r", f"ih{layer}_i")
                self._lstm_fwd_call(layer, 0, T, main)
            return
        s2, s3 = self.lstm_stream.cuda_stream, self.lstm_gemm_stream.cuda_stream
        for (t0, t1) in self.lstm_chunks:
            self._lstm_fwd_call(1, t0, t1, main)
            call("sehip_stream_depend", s3, main, self._event())
            call("sehip_gemm_pair", C.byref(self._chunk_desc("ih2_r", t0, t1)), C.byref(self._chunk_desc("ih2_i", t0, t1)), s3)
            call("sehip_stream_depend", s2, s3, self._event())
            self._lstm_fwd_call(2, t0, t1, s2)
        call("sehip_stream_depend", main, s2, self._event())

    def _lstm_bwd_call(self, layer, t0, t1, st_):
        b, cfg = self.bufs, self.st.cfg
        # a layer's output gradient: the projection's input gradient for the last one, the next layer's input gradient otherwise
        dha, dhb = (b["dxo_r"], b["dxo_i"]) if layer == cfg.rnn_layers else (b[f"dx{layer + 1}_r"], b[f"dx{layer + 1}_i"])
        whhT = self.tb.wpack.data_ptr() + 2 * self.st.whhT_off[layer]
        call("sehip_lstm_bwd_chunk", dha.ptr, dhb.ptr, whhT, b[f"gates{layer}"].ptr, b[f"c{layer}"].ptr, self.B, self.T, cfg.hid,
             t0, t1, ptr(self.lstm_state[layer]), b[f"dpre{layer}_r"].ptr, b[f"dpre{layer}_i"].ptr, st_)

    def _lstm_backward(self, B, T, h):
        self._chain_dirty = True
        main = stream()
        if not self.st.cfg.use_clstm:
            b, wp = self.bufs, self.tb.wpack.data_ptr()
            for layer in (2, 1):
                dh = b["dxo"] if layer == 2 else b["dx2"]
                call("sehip_rlstm_bwd", dh.ptr, wp + 2 * self.st.whhT_off[layer], b[f"gates{layer}"].ptr, b[f"c{layer}"].ptr, B, T, h,
                     b[f"dpre{layer}"].ptr, main)
                self._chain_dirty = True
                self.wgrad_group(self._lstm_wgrad_names((layer,)))
                self.gemm(f"dx{layer}")
            return
        if self.lstm_fused:
            b, st, tb = self.bufs, self.st, self.tb
            wp = tb.wpack.data_ptr()
            call("sehip_lstm2_bwd", b["dxo_r"].ptr, b["dxo_i"].ptr, wp + 2 * st.whhT_off[1], wp + 2 * st.whhT_off[2],
                 wp + 2 * st.wihT2_off, b["gates1"].ptr, b["c1"].ptr, b["gates2"].ptr, b["c2"].ptr, B, T, h, b["dpre1_r"].ptr,
                 b["dpre1_i"].ptr, b["dpre2_r"].ptr, b["dpre2_i"].ptr, ptr(self.l2_gran_b), ptr(self.l2_sync),
                 getattr(self, "_l2_cur_epoch", 1), main)
            self._chain_dirty = True
            self.wgrad_group(self._lstm_wgrad_names((2, 1)))
            self.gemm_pair("dx1_r", "dx1_i")
            return
        nl = self.st.cfg.rnn_layers
        if self.lstm_stream is None or nl != 2 or torch.cuda.is_current_stream_capturing():
            for layer in range(nl, 0, -1):
                self._lstm_bwd_call(layer, 0, T, main)
                self.wgrad_group(self._lstm_wgrad_names((layer,)))
                self.gemm_pair(f"dx{layer}_r", f"dx{layer}_i")
            return
        s2, s3 = self.lstm_stream.cuda_stream, self.lstm_gemm_stream.cuda_stream
        for (t0, t1) in reversed(self.lstm_chunks):
            self._lstm_bwd_call(2, t0, t1, main)                  # layer 2, later chunks first
            call("sehip_stream_depend", s3, main, self._event())
            # its input gradient = layer 1's output gradient
            call("sehip_gemm_pair", C.byref(self._chunk_desc("dx2_r", t0, t1)), C.byref(self._chunk_desc("dx2_i", t0, t1)), s3)
            call("sehip_stream_depend", s2, s3, self._event())
            self._lstm_bwd_call(1, t0, t1, s2)
        call("sehip_stream_depend", main, s2, self._event())
        self._chain_dirty = True
        self.wgrad_group(self._lstm_wgrad_names((2, 1)))
        self.gemm_pair("dx1_r", "dx1_i")

    # ---- forward / backward --------------------------------------------------------------------
    def pack_weights(self, params):
        st, tb = self.st, self.tb
        call("sehip_pack_bf16", ptr(params), ptr(tb.wtab), st.n_wpack, ptr(tb.wpack), stream())
        call("sehip_pack_f32", ptr(params), ptr(tb.btab), st.n_bpack, ptr(tb.bpack), stream())

    def forward(self, wav_in, params, buffers, nbt, training=True):
        """wav_in [B,N] fp32 on device -> self.wav [B,length]."""
        st, cfg, b, tb = self.st, self.st.cfg, self.bufs, self.tb
        B, T, h = self.B, self.T, cfg.hid
        # both packings and the clearing of the fused BatchNorm sums: one launch
        zero = self.bn_stats_all if (self.st.fused_stats or self.fused_small) else None
        # The step's head in parallel (round 5): the weight packing (28 us) and the clearing of the packed-gradient buffer (8 us, used
        # to open the backward pass) run on the weight-gradient stream, which is idle now, while the STFT -- which needs neither --
        # runs on the chain; the first encoder layer waits for both.
        head = self.side.cuda_stream if (self.side is not None and PARALLEL_HEAD) else stream()
        if head != stream():
            call("sehip_stream_depend", head, stream(), self._event())          # the previous step's optimizer wrote the parameters
        call("sehip_pack_head", ptr(params), ptr(tb.wtab), st.n_wpack, ptr(tb.wpack), ptr(tb.btab), st.n_bpack, ptr(tb.bpack),
             ptr(zero) if zero is not None else None, zero.numel() if zero is not None else 0, head)
        if head != stream() and training:
            with torch.cuda.stream(self.side):
                self.gpack.zero_()
            self._gpack_clean = True
        call("sehip_stft_fwd", ptr(wav_in), ptr(tb.window), B, self.N, cfg.win_len, cfg.win_inc, cfg.fft_len,
             ptr(self.spec), b["enc_in"].ptr, stream())
        if head != stream():
            call("sehip_stream_depend", stream(), head, self._event())
        for i in range(6):
            self.gemm(f"enc{i}.fwd")
            self.bn_forward(f"encoder.{i}.", cfg.kernel_num[i + 1] // 2, b[f"y{i}"], b[f"z{i}"], params, buffers, nbt, training)
        # the skip-connection halves of the deep decoders' forward products: on the weight-gradient stream (idle in the forward pass),
        # behind the encoder, beside the LSTM's 64 workgroups (DCCRNStatic._maybe_split_decoder); one event per layer
        split = st.dec_split if self.side is not None else []       # (also inside a stream capture: a fork / join of the graph, as the head)
        if split:
            sd = self.side.cuda_stream
            call("sehip_stream_depend", sd, stream(), self._event())
            if not self._fs_events:
                for _ in range(6):
                    e = _lib.lib().sehip_event_create()
                    if not e:
                        raise SehipError("sehip_event_create: " + _lib.lib().sehip_last_error().decode())
                    self._fs_events.append(e)
            for j in split:
                call("sehip_gemm_pair", C.byref(self.desc[f"dec{j}.fs0"]), C.byref(self.desc[f"dec{j}.fs1"]), sd)
                call("sehip_event_record", self._fs_events[j], sd)
        self._lstm_forward(B, T, h)
        if cfg.use_clstm:
            self.gemm_pair("proj_r", "proj_i")
        else:
            self.gemm("proj")
        for j in range(6):
            if j in split:
                call("sehip_stream_wait_event", stream(), self._fs_events[j])
                self.gemm_pair(f"dec{j}.fm0", f"dec{j}.fm1")
            else:
                self.gemm_pair(f"dec{j}.fwd0", f"dec{j}.fwd1")
            if j < 5:
                self.bn_forward(f"decoder.{j}.", cfg.kernel_num[5 - j] // 2, b[f"yd{j}"], b[f"zd{j}"], params, buffers, nbt,
                                training)
        call("sehip_istft_fwd", ptr(self.spec), b["mask"].ptr, ptr(tb.window), ptr(self.inv_coff), B, T, cfg.win_len,
             cfg.win_inc, cfg.fft_len, self.length, self.mode, ptr(self.frames), ptr(self.wav), stream())
        return self.wav

    def backward(self, dwav, params, grads, range_ready=None, tail=None):
        """dwav [B,length] fp32 -> flat parameter gradients (overwritten).

        range_ready(lo, hi, stream) -- data-parallel hook: called as soon as grads[lo:hi] is final ON `stream` (a torch stream),
        first for the decoder + LSTM parameters (the tail of the flat buffer in state_dict order), whose weight gradients are
        complete once the LSTM backward has been enqueued, so their all-reduce overlaps the encoder's backward pass; then for
        the encoder range at the end."""
        st, cfg, b, tb = self.st, self.st.cfg, self.bufs, self.tb
        B, T, h = self.B, self.T, cfg.hid
        if not getattr(self, "_gpack_clean", False):      # (forward() has cleared it on the idle side stream; a second backward pass
            self.gpack.zero_()                            #  over the same forward clears it here, as before)
        self._gpack_clean = False
        self._chain_dirty = True
        call("sehip_istft_bwd", ptr(dwav), ptr(self.wav), ptr(self.spec), b["mask"].ptr, ptr(tb.window), ptr(self.inv_coff),
             B, T, cfg.win_len, cfg.win_inc, cfg.fft_len, self.length, self.mode, b["dmask"].ptr, stream())
        for j in range(5, -1, -1):
            if j < 5:
                self.bn_backward(f"decoder.{j}.", cfg.kernel_num[5 - j] // 2, b[f"dzd{j}"], None, b[f"yd{j}"], b[f"dyd{j}"],
                                 params, 1)
            self.wgrad_pair(f"dec{j}.fwd0", f"dec{j}.fwd1")
            if j >= 1 and f"decoder.{j - 1}." in self.bnr_rows:
                self.desc[f"dec{j}.dg"].bnr_slope = params.data_ptr() + 4 * st.layout.param_off[f"decoder.{j - 1}.2.weight"][0]
            self.gemm(f"dec{j}.dg")
        if cfg.use_clstm:
            for tag in "ri":
                self.wgrad(f"proj_{tag}")
            self.gemm_pair("dproj_r", "dproj_i")
        else:
            self.wgrad("proj")
            self.gemm("dproj")
        self._lstm_backward(B, T, h)
        n_params = st.layout.n_params
        lo = st.layout.param_off["decoder.0.0.real_conv.weight"][0] if range_ready is not None else 0
        if range_ready is not None:
            # the early hand-over rests on the flat layout: [0, lo) = encoder parameters only (their gradients are written
            # below), [lo, n) = LSTM + decoder parameters only (final by now).  A layout change must fail here, not all-reduce
            # half-written gradients.
            L = st.layout
            if not getattr(st, "_dp_layout_checked", False):
                for nm in L.param_names:
                    early = L.param_off[nm][0] >= lo
                    if early != (not nm.startswith("encoder.")):
                        raise SehipError(f"DCCRN data-parallel hand-over: parameter {nm} at offset {L.param_off[nm][0]} is on the wrong "
                                         f"side of the decoder/LSTM split {lo}")
                st._dp_layout_checked = True
            # decoder / LSTM gradients: un-packed and handed over on a third stream that waits for the chain (BatchNorm / PReLU
            # gradients) and for the weight-gradient stream as they stand now -- the chain itself does not wait
            if self.comm is None:
                self.comm = torch.cuda.Stream(device=self.device)
            cs = self.comm.cuda_stream
            call("sehip_stream_depend", cs, stream(), self._event())
            if self.side is not None:
                call("sehip_stream_depend", cs, self.side.cuda_stream, self._event())
            call("sehip_unpack_grad", ptr(self.gpack), tb.utab.data_ptr() + 16 * lo, n_params - lo, grads.data_ptr() + 4 * lo, cs)
            range_ready(lo, n_params, self.comm)
        for i in range(5, -1, -1):
            dz = b["dz5l"] if i == 5 else b[f"dz{i}"]
            dz2 = b[f"dskip{i}"] if not FUSE_SKIP_GRAD else None   # enc{i+1}.dg0/dg1 (i = 5: the LSTM's dx1 products) already added it (res)
            in_wgrad = i == 0 and self.enc0_bn_in_wgrad and dz2 is None
            self.bn_backward(f"encoder.{i}.", cfg.kernel_num[i + 1] // 2, dz, dz2, b[f"y{i}"], b[f"dye{i}"], params, 0, apply=not in_wgrad)
            if i == 0:
                wd = self.desc["enc0.fwd.wg"]
                wd.bn_dz = dz.ptr if in_wgrad else None
                wd.bn_slope = params.data_ptr() + 4 * self.st.layout.param_off["encoder.0.2.weight"][0] if in_wgrad else None
            self.wgrad(f"enc{i}.fwd")
            if i > 0:
                self.gemm_pair(f"enc{i}.dg0", f"enc{i}.dg1")     # one streaming launch for the outer layers (csrc/convt.hip), else the two products
        if self.side is not None:
            call("sehip_stream_depend", stream(), self.side.cuda_stream, self._event())
        if tail is not None and range_ready is None:
            # tail = (sumsq, tensor_sums, offsets, ntensors, step counter) of the fused optimizer: the un-pack also takes its sums and
            # advances its device step counter (guarded by this workspace's hand-off word): see FlatOptimizer._arm_fused_tail
            guard = ptr(self.l2_sync) if hasattr(self, "l2_sync") else None
            if tb.uperm is not None:
                call("sehip_unpack_grad_sums_perm", ptr(self.gpack), ptr(tb.utab_g), ptr(tb.uperm), n_params, ptr(grads), tail[2], tail[3],
                     tail[0], tail[1], tail[4], guard, stream())
            else:
                call("sehip_unpack_grad_sums", ptr(self.gpack), ptr(tb.utab), n_params, ptr(grads), tail[2], tail[3], tail[0], tail[1], tail[4],
                     guard, stream())
        else:
            call("sehip_unpack_grad", ptr(self.gpack), ptr(tb.utab), lo if range_ready is not None else n_params, ptr(grads), stream())
        if range_ready is not None:
            range_ready(0, lo, torch.cuda.current_stream())
            call("sehip_stream_depend", stream(), self.comm.cuda_stream, self._event())
        return grads
