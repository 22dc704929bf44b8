"""CPU oracle for the DCCRN train step -- TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU *restatement* of the reference's algorithm for the
hot path.  It is written functionally (one dict of tensors keyed by
the reference's state_dict names, pure functions over it) and shares no code with the
reference.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product package (``speech-enhancement-pytorch_amd/sehip``) never
does and fails loudly when its HIP library is missing.

Parity pinning: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which ``oracle/gen_golden.py`` produced by importing the real
reference from /root/reference in the build container (forward activations, loss,
every parameter gradient, two Solver steps of Adam state and the logged scalars).

Reference anchors (paths relative to /root/reference):
  conv-STFT / iSTFT kernels ....... src/model/dccrn.py:649-747
  ComplexConv2d ................... src/model/dccrn.py:316-384
  ComplexConvTranspose2d .......... src/model/dccrn.py:387-450
  complex_cat ..................... src/model/dccrn.py:304-314
  ComplexBatchNorm ................ src/model/dccrn.py:457-634
  NavieComplexLSTM ................ src/model/dccrn.py:264-302
  DCCRN.forward ................... src/model/dccrn.py:145-229
  si_snr / loss_sisdr ............. src/loss.py:14-29
  train step (clip, Adam, metric) . src/solver.py:480-500, src/distrib.py:244-261
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_KERNEL_NUM = (16, 32, 64, 128, 256, 256)


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
class DCCRNConfig:
    """Constructor arguments of the reference model (src/model/dccrn.py:12-27)."""

    def __init__(self, rnn_layers=2, rnn_units=128, win_len=400, win_inc=100, fft_len=512,
                 length=16384, masking_mode="E", kernel_size=5,
                 kernel_num=DEFAULT_KERNEL_NUM, win_type="hann", use_cbn=True, use_clstm=True, **_ignored):
        self.win_type = win_type
        self.use_cbn = bool(use_cbn)
        self.use_clstm = bool(use_clstm)
        self.rnn_layers = rnn_layers
        self.rnn_units = rnn_units
        self.win_len = win_len
        self.win_inc = win_inc
        self.fft_len = fft_len
        self.length = length
        self.masking_mode = masking_mode
        self.kernel_size = kernel_size
        self.kernel_num = [2] + list(kernel_num)  # src/model/dccrn.py:51
        self.hidden_dim = fft_len // (2 ** len(self.kernel_num))  # src/model/dccrn.py:82
        self.n_layers = len(self.kernel_num) - 1


# --------------------------------------------------------------------------------------
# optional bf16 storage simulation (used by the GPU parity tests to separate rounding from logic:
# the HIP path stores activations / activation gradients in bf16 and feeds bf16 operands to the MFMA)
# --------------------------------------------------------------------------------------
class _RoundAct(torch.autograd.Function):
    """bf16 round-trip of an activation; its gradient is stored in bf16 too."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundWeight(torch.autograd.Function):
    """bf16 round-trip of a GEMM weight; the weight gradient is accumulated in fp32 (straight-through)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class Bf16Sim:
    act = staticmethod(_RoundAct.apply)
    weight = staticmethod(_RoundWeight.apply)


class NoSim:
    act = staticmethod(lambda x: x)
    weight = staticmethod(lambda x: x)


# --------------------------------------------------------------------------------------
# STFT bases (src/model/dccrn.py:649-666)
# --------------------------------------------------------------------------------------
def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True) == 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def window_of(win_type, n: int) -> np.ndarray:
    """init_kernels' window (src/model/dccrn.py:650-653): ones for None / 'None', else scipy.signal.get_window(win_type, n, fftbins=True)
    (the `**0.5` there is commented out).  'hann' and 'hamming' in closed form (the periodic cosine windows), anything else through scipy."""
    if win_type is None or win_type == "None":
        return np.ones(n, dtype=np.float64)
    k = np.arange(n, dtype=np.float64)
    if win_type == "hann":
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)
    if win_type == "hamming":
        return 0.54 - 0.46 * np.cos(2.0 * np.pi * k / n)
    from scipy.signal import get_window
    return np.asarray(get_window(win_type, n, fftbins=True), dtype=np.float64)


def stft_bases(win_len: int, fft_len: int, win_type="hann"):
    """Returns (analysis[2*(fft/2+1), win], synthesis[2*(fft/2+1), win], window[win]) as fp32.

    analysis rows = [cos rows ; -sin rows] * window; synthesis = pinv(unwindowed).T * window.
    """
    n = np.arange(win_len, dtype=np.float64)[None, :]
    k = np.arange(fft_len // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * k * n / fft_len
    basis = np.concatenate([np.cos(ang), -np.sin(ang)], axis=0)  # [2F, win]
    win = window_of(win_type, win_len)
    analysis = basis * win[None, :]
    synthesis = np.linalg.pinv(basis).T * win[None, :]
    return (torch.from_numpy(analysis.astype(np.float32)),
            torch.from_numpy(synthesis.astype(np.float32)),
            torch.from_numpy(win.astype(np.float32)))


def conv_stft(wav: torch.Tensor, analysis: torch.Tensor, win_len: int, hop: int) -> torch.Tensor:
    """[B,1,N] -> [B, 2F, T]; zero pad (win-hop) both sides (src/model/dccrn.py:687-694)."""
    if wav.dim() == 2:
        wav = wav.unsqueeze(1)
    pad = win_len - hop
    return F.conv1d(F.pad(wav, [pad, pad]), analysis[:, None, :], stride=hop)


def conv_istft(spec: torch.Tensor, synthesis: torch.Tensor, window: torch.Tensor,
               win_len: int, hop: int, length: int) -> torch.Tensor:
    """[B, 2F, T] -> [B,1,length]: overlap-add / window-energy, trim (src/model/dccrn.py:723-747)."""
    frames = spec.shape[-1]
    ola = F.conv_transpose1d(spec, synthesis[:, None, :], stride=hop)
    wsq = (window.view(1, -1, 1) ** 2).repeat(1, 1, frames)
    energy = F.conv_transpose1d(wsq, torch.eye(win_len)[:, None, :], stride=hop)
    out = ola / (energy + 1e-8)
    out = out[..., win_len - hop:]
    return out[..., :length]


# --------------------------------------------------------------------------------------
# complex layers
# --------------------------------------------------------------------------------------
def complex_conv2d(x, wr, br, wi, bi, freq_pad=2, time_pad=1):
    """Channel axis = [real half | imag half]; causal left time pad (src/model/dccrn.py:358-384).

    Each real conv carries its own bias, so real gets (br - bi) and imag gets (bi + br).
    """
    x = F.pad(x, [time_pad, 0, 0, 0])
    xr, xi = torch.chunk(x, 2, 1)
    conv = lambda a, w, b: F.conv2d(a, w, b, stride=(2, 1), padding=(freq_pad, 0))
    real = conv(xr, wr, br) - conv(xi, wi, bi)
    imag = conv(xr, wi, bi) + conv(xi, wr, br)
    return torch.cat([real, imag], 1)


def complex_deconv2d(x, wr, br, wi, bi):
    """Transposed complex conv k(5,2) s(2,1) p(2,0) op(1,0) (src/model/dccrn.py:423-450)."""
    xr, xi = torch.chunk(x, 2, 1)
    dec = lambda a, w, b: F.conv_transpose2d(a, w, b, stride=(2, 1), padding=(2, 0),
                                             output_padding=(1, 0))
    real = dec(xr, wr, br) - dec(xi, wi, bi)
    imag = dec(xr, wi, bi) + dec(xi, wr, br)
    return torch.cat([real, imag], 1)


def complex_cat(a, b):
    """[a_r, b_r, a_i, b_i] along channels (src/model/dccrn.py:304-314)."""
    ar, ai = torch.chunk(a, 2, 1)
    br, bi = torch.chunk(b, 2, 1)
    return torch.cat([ar, br, ai, bi], 1)


def whitening_matrix(vrr, vri, vii):
    """Inverse square root of [[vrr, vri],[vri, vii]] (src/model/dccrn.py:593-602)."""
    tau = vrr + vii
    delta = vrr * vii - vri * vri
    s = delta.sqrt()
    t = (tau + 2 * s).sqrt()
    rst = 1.0 / (s * t)
    return (s + vii) * rst, -vri * rst, (s + vrr) * rst  # Urr, Uri, Uii


def complex_batchnorm(x, p, prefix, training, eps=1e-5, momentum=0.1, stats_out=None):
    """src/model/dccrn.py:520-630.  ``p`` holds Wrr/Wri/Wii/Br/Bi and RM*/RV* under prefix.

    Training mode uses *biased* batch moments and (when ``stats_out`` is a dict) reports
    the lerp-updated running statistics without mutating ``p``.
    """
    xr, xi = torch.chunk(x, 2, 1)
    dims = [0, 2, 3]
    shape = [1, -1, 1, 1]
    if training:
        mr = xr.mean(dims)
        mi = xi.mean(dims)
        cr = xr - mr.view(shape)
        ci = xi - mi.view(shape)
        vrr = (cr * cr).mean(dims)
        vri = (cr * ci).mean(dims)
        vii = (ci * ci).mean(dims)
        if stats_out is not None:
            for name, val in (("RMr", mr), ("RMi", mi), ("RVrr", vrr), ("RVri", vri), ("RVii", vii)):
                old = p[prefix + name]
                stats_out[prefix + name] = (old + momentum * (val.detach() - old))
            stats_out[prefix + "num_batches_tracked"] = p[prefix + "num_batches_tracked"] + 1
    else:
        mr, mi = p[prefix + "RMr"], p[prefix + "RMi"]
        vrr, vri, vii = p[prefix + "RVrr"], p[prefix + "RVri"], p[prefix + "RVii"]
        cr = xr - mr.view(shape)
        ci = xi - mi.view(shape)
    urr, uri, uii = whitening_matrix(vrr + eps, vri, vii + eps)
    wrr, wri, wii = p[prefix + "Wrr"], p[prefix + "Wri"], p[prefix + "Wii"]
    zrr = wrr * urr + wri * uri
    zri = wrr * uri + wri * uii
    zir = wri * urr + wii * uri
    zii = wri * uri + wii * uii
    yr = zrr.view(shape) * cr + zri.view(shape) * ci + p[prefix + "Br"].view(shape)
    yi = zir.view(shape) * cr + zii.view(shape) * ci + p[prefix + "Bi"].view(shape)
    return torch.cat([yr, yi], 1)


def real_batchnorm(x, p, prefix, training, eps=1e-5, momentum=0.1, stats_out=None):
    """use_cbn=False (src/model/dccrn.py:110-113, :130-133): nn.BatchNorm2d over the [real half | imaginary half] channels.  Training mode
    normalises with the BIASED batch variance and reports (when ``stats_out`` is a dict) the running statistics nn.BatchNorm2d would
    hold afterwards -- mean by lerp, variance by lerp towards the UNBIASED batch variance -- without mutating ``p``."""
    dims = [0, 2, 3]
    shape = [1, -1, 1, 1]
    if training:
        m = x.mean(dims)
        c = x - m.view(shape)
        v = (c * c).mean(dims)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            old_m, old_v = p[prefix + "running_mean"], p[prefix + "running_var"]
            stats_out[prefix + "running_mean"] = old_m + momentum * (m.detach() - old_m)
            stats_out[prefix + "running_var"] = old_v + momentum * (v.detach() * (n / max(n - 1, 1)) - old_v)
            stats_out[prefix + "num_batches_tracked"] = p[prefix + "num_batches_tracked"] + 1
    else:
        m, v = p[prefix + "running_mean"], p[prefix + "running_var"]
        c = x - m.view(shape)
    return c * (p[prefix + "weight"] / (v + eps).sqrt()).view(shape) + p[prefix + "bias"].view(shape)


def lstm_single(x, w_ih, w_hh, b_ih, b_hh, sim=NoSim):
    """One-layer unidirectional nn.LSTM with zero initial state; x [T,B,I] -> [T,B,H].

    Gate order i,f,g,o (PyTorch).  Written as an explicit recurrence so that the HIP
    kernel's math has a line-by-line CPU statement.
    """
    steps, batch, _ = x.shape
    hidden = w_hh.shape[1]
    w_ih, w_hh = sim.weight(w_ih), sim.weight(w_hh)
    pre = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(batch, hidden)
    c = x.new_zeros(batch, hidden)
    outs = []
    for t in range(steps):
        gates = pre[t] + h @ w_hh.t()
        i, f, g, o = gates.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = sim.act(torch.sigmoid(o) * torch.tanh(c))
        outs.append(h)
    return torch.stack(outs, 0)


def complex_lstm(real, imag, p, prefix, has_projection, sim=NoSim):
    """src/model/dccrn.py:283-298: four LSTM passes + optional 2 Linear projections."""
    def run(which, x):
        q = prefix + which + "."
        return lstm_single(x, p[q + "weight_ih_l0"], p[q + "weight_hh_l0"],
                           p[q + "bias_ih_l0"], p[q + "bias_hh_l0"], sim)
    r2r = run("real_lstm", real)
    r2i = run("imag_lstm", real)
    i2r = run("real_lstm", imag)
    i2i = run("imag_lstm", imag)
    out_r = r2r - i2i
    out_i = i2r + r2i
    if has_projection:
        out_r = F.linear(out_r, sim.weight(p[prefix + "r_trans.weight"]), p[prefix + "r_trans.bias"])
        out_i = F.linear(out_i, sim.weight(p[prefix + "i_trans.weight"]), p[prefix + "i_trans.bias"])
    return out_r, out_i


# --------------------------------------------------------------------------------------
# DCCRN forward (src/model/dccrn.py:145-229)
# --------------------------------------------------------------------------------------
def dccrn_forward(p, wav, cfg: DCCRNConfig, training=True, capture=None, stats_out=None,
                  bases=None, sim=NoSim):
    """p: dict name->tensor with the reference's state_dict keys.  wav [B,1,N] -> [B,1,length].

    ``capture`` (dict) receives named intermediates; ``stats_out`` the updated BN buffers.
    """
    if bases is None:
        bases = stft_bases(cfg.win_len, cfg.fft_len, getattr(cfg, "win_type", "hann"))
    analysis, synthesis, window = bases
    nbin = cfg.fft_len // 2 + 1
    cap = capture if capture is not None else {}

    spec = conv_stft(wav, analysis, cfg.win_len, cfg.win_inc)
    cap["stft"] = spec
    real, imag = spec[:, :nbin], spec[:, nbin:]
    mags = torch.sqrt(real ** 2 + imag ** 2 + 1e-8)
    phase = torch.atan2(imag, real)
    out = sim.act(torch.stack([real, imag], 1)[:, :, 1:])  # drop DC bin

    skips = []
    for i in range(cfg.n_layers):
        pre = f"encoder.{i}."
        out = complex_conv2d(out, sim.weight(p[pre + "0.real_conv.weight"]), p[pre + "0.real_conv.bias"],
                             sim.weight(p[pre + "0.imag_conv.weight"]), p[pre + "0.imag_conv.bias"])
        out = sim.act(out)
        cap[f"enc{i}.conv"] = out
        out = (complex_batchnorm if cfg.use_cbn else real_batchnorm)(out, p, pre + "1.", training, stats_out=stats_out)
        out = sim.act(F.prelu(out, p[pre + "2.weight"]))
        cap[f"enc{i}"] = out
        skips.append(out)

    b, ch, d, t = out.shape
    seq = out.permute(3, 0, 1, 2)  # [T,B,C,D]
    r_in = seq[:, :, : ch // 2].reshape(t, b, ch // 2 * d)
    i_in = seq[:, :, ch // 2:].reshape(t, b, ch // 2 * d)
    if not getattr(cfg, "use_clstm", True):
        # src/model/dccrn.py:184-189: ONE real nn.LSTM over all channels (two layers whatever rnn_layers says, :98-106) and a Linear
        x = seq.reshape(t, b, ch * d)
        for layer in range(2):
            x = lstm_single(x, p[f"enhance.weight_ih_l{layer}"], p[f"enhance.weight_hh_l{layer}"],
                            p[f"enhance.bias_ih_l{layer}"], p[f"enhance.bias_hh_l{layer}"], sim)
            cap[f"lstm{layer}"] = x
        x = sim.act(F.linear(x, sim.weight(p["tranform.weight"]), p["tranform.bias"]))
        out = x.reshape(t, b, ch, d).permute(1, 2, 3, 0)
    else:
        for layer in range(cfg.rnn_layers):
            r_in, i_in = complex_lstm(r_in, i_in, p, f"enhance.{layer}.",
                                      has_projection=(layer == cfg.rnn_layers - 1), sim=sim)
            if layer == cfg.rnn_layers - 1:
                r_in, i_in = sim.act(r_in), sim.act(i_in)
            cap[f"lstm{layer}.r"] = r_in
            cap[f"lstm{layer}.i"] = i_in
        r_in = r_in.reshape(t, b, ch // 2, d)
        i_in = i_in.reshape(t, b, ch // 2, d)
        out = torch.cat([r_in, i_in], 2).permute(1, 2, 3, 0)

    for i in range(cfg.n_layers):
        pre = f"decoder.{i}."
        out = complex_cat(out, skips[-1 - i])
        out = complex_deconv2d(out, sim.weight(p[pre + "0.real_conv.weight"]), p[pre + "0.real_conv.bias"],
                               sim.weight(p[pre + "0.imag_conv.weight"]), p[pre + "0.imag_conv.bias"])
        if i != cfg.n_layers - 1:
            out = sim.act(out)
            cap[f"dec{i}.conv"] = out[..., 1:]
            out = (complex_batchnorm if cfg.use_cbn else real_batchnorm)(out, p, pre + "1.", training, stats_out=stats_out)
            out = sim.act(F.prelu(out, p[pre + "2.weight"]))
        out = out[..., 1:]  # drop the first frame (src/model/dccrn.py:196)
        cap[f"dec{i}"] = out

    m_r = F.pad(out[:, 0], [0, 0, 1, 0])
    m_i = F.pad(out[:, 1], [0, 0, 1, 0])
    if cfg.masking_mode == "E":
        m_mag = (m_r ** 2 + m_i ** 2) ** 0.5
        m_phase = torch.atan2(m_i / (m_mag + 1e-8), m_r / (m_mag + 1e-8))
        est_mag = torch.tanh(m_mag) * mags
        est_phase = phase + m_phase
        real = est_mag * torch.cos(est_phase)
        imag = est_mag * torch.sin(est_phase)
    elif cfg.masking_mode == "C":
        real, imag = real * m_r - imag * m_i, real * m_i + imag * m_r
    elif cfg.masking_mode == "R":
        real, imag = real * m_r, imag * m_i
    else:
        raise ValueError(cfg.masking_mode)
    est = torch.cat([real, imag], 1)
    cap["est_spec"] = est
    wav_out = conv_istft(est, synthesis, window, cfg.win_len, cfg.win_inc, cfg.length)
    cap["istft"] = wav_out
    return torch.clamp(wav_out, -1, 1)


# NOTE on BatchNorm inside the decoder: the reference normalises the (T+1)-frame tensor
# and drops the first frame afterwards, so the batch statistics include that frame.


# --------------------------------------------------------------------------------------
# loss (src/loss.py:14-29)
# --------------------------------------------------------------------------------------
def si_snr(est, ref, eps=1e-8):
    dot = (est * ref).sum(-1, keepdim=True)
    ref_energy = (ref * ref).sum(-1, keepdim=True)
    target = dot / (ref_energy + eps) * ref
    noise = est - target
    ratio = (target * target).sum(-1, keepdim=True) / ((noise * noise).sum(-1, keepdim=True) + eps)
    return (10 * torch.log10(ratio + eps)).mean()


def loss_sisdr(est, ref):
    return -si_snr(est, ref)


# --------------------------------------------------------------------------------------
# parameter initialisation (same distributions as the reference constructors; the values are
# only bit-identical to the reference when loaded from its state_dict)
# --------------------------------------------------------------------------------------
def init_params(cfg: DCCRNConfig, seed=0) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    kn = cfg.kernel_num

    def conv_block(pre, w_shape, nb):
        for part in ("real_conv", "imag_conv"):
            p[f"{pre}0.{part}.weight"] = torch.randn(w_shape, generator=g) * 0.05
            p[f"{pre}0.{part}.bias"] = torch.zeros(nb)

    def bn_block(pre, n):
        if not cfg.use_cbn:          # nn.BatchNorm2d(2 n): weight 1, bias 0, running (0, 1)
            p[pre + "1.weight"] = torch.ones(2 * n)
            p[pre + "1.bias"] = torch.zeros(2 * n)
            p[pre + "1.running_mean"] = torch.zeros(2 * n)
            p[pre + "1.running_var"] = torch.ones(2 * n)
            p[pre + "1.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
            p[pre + "2.weight"] = torch.full((1,), 0.25)
            return
        p[pre + "1.Wrr"] = torch.ones(n)
        p[pre + "1.Wri"] = torch.rand(n, generator=g) * 1.8 - 0.9
        p[pre + "1.Wii"] = torch.ones(n)
        p[pre + "1.Br"] = torch.zeros(n)
        p[pre + "1.Bi"] = torch.zeros(n)
        p[pre + "1.RMr"] = torch.zeros(n)
        p[pre + "1.RMi"] = torch.zeros(n)
        p[pre + "1.RVrr"] = torch.ones(n)
        p[pre + "1.RVri"] = torch.zeros(n)
        p[pre + "1.RVii"] = torch.ones(n)
        p[pre + "1.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        p[pre + "2.weight"] = torch.full((1,), 0.25)

    for i in range(cfg.n_layers):
        cin, cout = kn[i] // 2, kn[i + 1] // 2
        conv_block(f"encoder.{i}.", (cout, cin, cfg.kernel_size, 2), cout)
        bn_block(f"encoder.{i}.", cout)
    hid = cfg.rnn_units // 2
    width = cfg.hidden_dim * kn[-1] // 2
    if not getattr(cfg, "use_clstm", True):       # nn.LSTM(hidden_dim * kernel_num[-1], rnn_units, num_layers=2) + nn.Linear
        hr, k = cfg.rnn_units, 1.0 / math.sqrt(cfg.rnn_units)
        for layer in range(2):
            nin = 2 * width if layer == 0 else hr
            p[f"enhance.weight_ih_l{layer}"] = (torch.rand(4 * hr, nin, generator=g) * 2 - 1) * k
            p[f"enhance.weight_hh_l{layer}"] = (torch.rand(4 * hr, hr, generator=g) * 2 - 1) * k
            p[f"enhance.bias_ih_l{layer}"] = (torch.rand(4 * hr, generator=g) * 2 - 1) * k
            p[f"enhance.bias_hh_l{layer}"] = (torch.rand(4 * hr, generator=g) * 2 - 1) * k
        p["tranform.weight"] = (torch.rand(2 * width, hr, generator=g) * 2 - 1) * k
        p["tranform.bias"] = (torch.rand(2 * width, generator=g) * 2 - 1) * k
    for layer in range(cfg.rnn_layers if getattr(cfg, "use_clstm", True) else 0):
        nin = width if layer == 0 else hid
        k = 1.0 / math.sqrt(hid)
        for part in ("real_lstm", "imag_lstm"):
            q = f"enhance.{layer}.{part}."
            p[q + "weight_ih_l0"] = (torch.rand(4 * hid, nin, generator=g) * 2 - 1) * k
            p[q + "weight_hh_l0"] = (torch.rand(4 * hid, hid, generator=g) * 2 - 1) * k
            p[q + "bias_ih_l0"] = (torch.rand(4 * hid, generator=g) * 2 - 1) * k
            p[q + "bias_hh_l0"] = (torch.rand(4 * hid, generator=g) * 2 - 1) * k
        if layer == cfg.rnn_layers - 1:
            for part in ("r_trans", "i_trans"):
                p[f"enhance.{layer}.{part}.weight"] = (torch.rand(width, hid, generator=g) * 2 - 1) * k
                p[f"enhance.{layer}.{part}.bias"] = (torch.rand(width, generator=g) * 2 - 1) * k
    for j, idx in enumerate(range(cfg.n_layers, 0, -1)):
        cin, cout = kn[idx], kn[idx - 1] // 2  # cin_r = (2*kn[idx])//2
        conv_block(f"decoder.{j}.", (cin, cout, cfg.kernel_size, 2), cout)
        if idx != 1:
            bn_block(f"decoder.{j}.", cout)
    return p


def perturb_params(p, seed=5):
    """Non-trivial biases / BatchNorm affine terms / PReLU slopes (the constructors' zeros and ones would hide sign and
    indexing errors).  Deterministic in `seed`: the golden generator and the tests rebuild the same weights from it."""
    g = torch.Generator().manual_seed(seed)
    for k in p:
        if k.endswith(".bias") or k.endswith((".Br", ".Bi")):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g)
        if k.endswith((".Wrr", ".Wii")) or (k.endswith(".1.weight") and k.startswith(("encoder.", "decoder."))):
            p[k] = 1.0 + 0.2 * torch.randn(p[k].shape, generator=g)
        if k.endswith("2.weight"):
            p[k] = 0.25 + 0.1 * torch.randn(p[k].shape, generator=g)
    return p


BUFFER_SUFFIXES = ("RMr", "RMi", "RVrr", "RVri", "RVii", "num_batches_tracked", "running_mean", "running_var")


def is_trainable(name: str) -> bool:
    return not name.endswith(BUFFER_SUFFIXES) and not name.startswith(("stft.", "istft."))


# --------------------------------------------------------------------------------------
# one Solver train step (src/solver.py:461-500): forward, SI-SNR, backward, clip, Adam,
# the reference's sum-based grad_norm metric
# --------------------------------------------------------------------------------------
class AdamState:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics: coef = min(1, max/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def adam_update(params, grads, st: AdamState):
    st.step += 1
    b1, b2 = st.betas
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    for k, g in grads.items():
        st.m[k].mul_(b1).add_(g, alpha=1 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(st.eps)
        params[k].addcdiv_(st.m[k], denom, value=-st.lr / bc1)


def train_step(p, noisy, clean, cfg, adam: AdamState, clip_grad=5.0, bases=None):
    """Mutates ``p`` (trainable tensors + BN buffers) in place; returns (loss, grad_norm_metric, grads)."""
    names = [k for k in p if is_trainable(k)]
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(p)
    work.update(leaves)
    stats = {}
    est = dccrn_forward(work, noisy, cfg, training=True, stats_out=stats, bases=bases)
    loss = loss_sisdr(est, clean)
    grads_list = torch.autograd.grad(loss, [leaves[k] for k in names])
    grads = {k: g.clone() for k, g in zip(names, grads_list)}
    if clip_grad:
        clip_grad_norm(grads, float(clip_grad))
    tr = {k: p[k] for k in names}
    with torch.no_grad():
        adam_update(tr, grads, adam)
        for k, v in stats.items():
            p[k] = v
    metric = math.sqrt(sum(float(g.sum()) ** 2 for g in grads.values()))  # src/solver.py:494-498
    return float(loss.detach()), metric, grads
