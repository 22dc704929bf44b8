"""CPU oracle for the complex DCUNet forward / loss / gradients -- TEST INFRASTRUCTURE ONLY.

Functional fp32 PyTorch-CPU restatement of the reference's DCUnet with ``data_type=True`` (complex) for depth 10 and 20
(reference: src/model/dcunet.py:53-162 DCUnet, :164-321 set_size, :8-50 Encoder / Decoder, :323-386 ComplexConv2d /
ComplexConvTranspose2d / ComplexBatchNorm2d).  It shares no code with the reference: parameters live in one dict keyed by
the reference's state_dict names (``encoder{i}.conv.conv_re.weight`` ...; the reference registers every block a second
time under ``encoders.{i}`` / ``decoders.{i}`` -- the same tensors, ignored here).  No HIP path consumes it yet: it is the
pinned starting point of the next widening step.  Only tests/ may import it.

Parity pinning: tests/test_oracle_golden.py::test_dcunet_oracle_matches_reference checks forward activations, the masked
output, an mse loss and every parameter gradient against tests/golden/dcunet_tiny.npz, which oracle/gen_golden_dcunet.py
produced by importing the real reference in the build container.

Shapes: input ``[B, C, F=257, T, 2]`` (stft_custom's layout); the network works on ``[B, C, T, F, 2]`` after
``transpose(2, 3)`` (src/model/dcunet.py:106), i.e. kernel / stride / padding pairs are (time, frequency).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .dccrn_oracle import Bf16Sim, NoSim   # bf16 round-trips at the HIP path's storage points (tests only)


def dcunet_sizes(model_complexity: int, model_depth: int, audio_channels: int = 1, complex_: bool = True):
    """Channel / kernel / stride / padding tables (src/model/dcunet.py:164-321); complex halves the width by sqrt(2) (:64-65)."""
    mc = int(model_complexity // 1.414) if complex_ else model_complexity
    if model_depth == 10:
        enc_ch = [audio_channels, mc, mc * 2, mc * 2, mc * 2, mc * 2]
        enc_k = [(7, 5), (7, 5), (5, 3), (5, 3), (5, 3)]
        enc_s = [(2, 2), (2, 2), (2, 2), (2, 2), (2, 1)]
        enc_p = [(2, 1), None, None, None, None]
        dec_ch = [0, mc * 2, mc * 2, mc * 2, mc * 2, mc * 2]
        dec_k = [(4, 3), (4, 4), (6, 4), (6, 4), (7, 5)]
        dec_s = [(2, 1), (2, 2), (2, 2), (2, 2), (2, 2)]
        dec_p = [(1, 1), (1, 1), (2, 1), (2, 1), (2, 1)]
    elif model_depth == 20:
        enc_ch = [audio_channels, mc, mc, mc * 2, mc * 2, mc * 2, mc * 2, mc * 2, mc * 2, mc * 2, 128]
        enc_k = [(7, 1), (1, 7), (6, 4), (7, 5), (5, 3), (5, 3), (5, 3), (5, 3), (5, 3), (5, 3)]
        enc_s = [(1, 1), (1, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1)]
        enc_p = [(3, 0), (0, 3)] + [None] * 8
        dec_ch = [0] + [mc * 2] * 11
        dec_k = [(4, 3), (4, 2), (4, 3), (4, 2), (4, 3), (4, 2), (6, 3), (7, 5), (1, 7), (7, 1)]
        dec_s = [(2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (1, 1), (1, 1)]
        dec_p = [(1, 1), (1, 0), (1, 1), (1, 0), (1, 1), (1, 0), (2, 1), (2, 1), (0, 3), (3, 0)]
    else:
        raise ValueError(f"Unknown model depth : {model_depth}")
    enc_p = [tuple((k - 1) // 2 for k in ks) if p is None else p for ks, p in zip(enc_k, enc_p)]  # 'SAME' (:12-13)
    return dict(n=model_depth // 2, enc_ch=enc_ch, enc_k=enc_k, enc_s=enc_s, enc_p=enc_p, dec_ch=dec_ch, dec_k=dec_k,
                dec_s=dec_s, dec_p=dec_p)


def complex_conv2d(x, p, pre, stride, padding, sim=NoSim):
    """src/model/dcunet.py:323-338: real = Wre*xr - Wim*xi, imag = Wre*xi + Wim*xr, each conv with its own bias
    (so the real part carries b_re - b_im and the imaginary part b_re + b_im).
    sim=Bf16Sim: bf16 weights, and the bias-free product is what is stored in bf16 (the HIP path keeps its pre-BatchNorm tensors
    without the bias, which the BatchNorm cancels); the bias terms are added to the rounded value."""
    xr, xi = x[..., 0], x[..., 1]
    if sim is NoSim:
        cre = lambda t: F.conv2d(t, p[pre + "conv_re.weight"], p[pre + "conv_re.bias"], stride=stride, padding=padding)
        cim = lambda t: F.conv2d(t, p[pre + "conv_im.weight"], p[pre + "conv_im.bias"], stride=stride, padding=padding)
        return torch.stack((cre(xr) - cim(xi), cre(xi) + cim(xr)), dim=-1)
    wre, wim = sim.weight(p[pre + "conv_re.weight"]), sim.weight(p[pre + "conv_im.weight"])
    cre = lambda t: F.conv2d(t, wre, None, stride=stride, padding=padding)
    cim = lambda t: F.conv2d(t, wim, None, stride=stride, padding=padding)
    bre, bim = p[pre + "conv_re.bias"][None, :, None, None], p[pre + "conv_im.bias"][None, :, None, None]
    return torch.stack((sim.act(cre(xr) - cim(xi)) + (bre - bim), sim.act(cre(xi) + cim(xr)) + (bre + bim)), dim=-1)


def complex_conv_transpose2d(x, p, pre, stride, padding, sim=NoSim):
    """src/model/dcunet.py:341-371 (output_padding 0, dilation 1).  sim: as complex_conv2d."""
    xr, xi = x[..., 0], x[..., 1]
    if sim is NoSim:
        tre = lambda t: F.conv_transpose2d(t, p[pre + "tconv_re.weight"], p[pre + "tconv_re.bias"], stride=stride, padding=padding)
        tim = lambda t: F.conv_transpose2d(t, p[pre + "tconv_im.weight"], p[pre + "tconv_im.bias"], stride=stride, padding=padding)
        return torch.stack((tre(xr) - tim(xi), tre(xi) + tim(xr)), dim=-1)
    wre, wim = sim.weight(p[pre + "tconv_re.weight"]), sim.weight(p[pre + "tconv_im.weight"])
    tre = lambda t: F.conv_transpose2d(t, wre, None, stride=stride, padding=padding)
    tim = lambda t: F.conv_transpose2d(t, wim, None, stride=stride, padding=padding)
    bre, bim = p[pre + "tconv_re.bias"][None, :, None, None], p[pre + "tconv_im.bias"][None, :, None, None]
    return torch.stack((sim.act(tre(xr) - tim(xi)) + (bre - bim), sim.act(tre(xi) + tim(xr)) + (bre + bim)), dim=-1)


def complex_batchnorm2d(x, p, pre, training, stats_out=None, momentum=0.1, eps=1e-5):
    """src/model/dcunet.py:374-386: two INDEPENDENT real BatchNorm2d (no whitening): per channel over (B, T, F), biased
    variance for the normalisation, running_var updated with the unbiased one (nn.BatchNorm2d)."""
    outs = []
    for part, idx in (("bn_re.", 0), ("bn_im.", 1)):
        t = x[..., idx]
        if training:
            mean = t.mean(dim=(0, 2, 3))
            var = t.var(dim=(0, 2, 3), unbiased=False)
            if stats_out is not None:
                n = t.numel() // t.shape[1]
                stats_out[pre + part + "running_mean"] = (1 - momentum) * p[pre + part + "running_mean"] + momentum * mean.detach()
                stats_out[pre + part + "running_var"] = (1 - momentum) * p[pre + part + "running_var"] + momentum * var.detach() * n / (n - 1)
        else:
            mean, var = p[pre + part + "running_mean"], p[pre + part + "running_var"]
        y = (t - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
        outs.append(y * p[pre + part + "weight"][None, :, None, None] + p[pre + part + "bias"][None, :, None, None])
    return torch.stack(outs, dim=-1)


def _lrelu(h, tag, act_masks):
    """LeakyReLU(0.01) of Encoder / Decoder (src/model/dcunet.py:8-50).  With ``act_masks`` (tests only) the branch of every
    element is taken from the given boolean tensor instead of the sign of h: the gradient then flows through the SAME branches as
    in the run the masks came from -- a comparison of backward passes that a handful of sign flips of near-zero
    pre-activations (1 % forward error of a bf16 path) does not dominate."""
    if act_masks is None:
        return F.leaky_relu(h, 0.01)
    return h * torch.where(act_masks[tag], torch.ones((), dtype=h.dtype), torch.full((), 0.01, dtype=h.dtype))


def dcunet_forward(p, x, model_complexity=45, model_depth=10, masking_mode="E", training=True, stats_out=None, taps=None,
                   act_masks=None, sim=NoSim):
    """x [B, C, F, T, 2] -> enhanced spectrum of the same shape (src/model/dcunet.py:102-162).  ``taps``: optional dict that
    receives the output of every encoder / decoder block; ``act_masks``: see _lrelu.
    ``sim=Bf16Sim`` (tests only): bf16 round-trips where the HIP path stores bf16 -- the packed input spectrum, every (bias-free)
    convolution output, every BatchNorm + LeakyReLU output, the convolution weights as MFMA operands, and (through the autograd
    twins of those round-trips) the activation gradients at the same points; the final 1x1 convolution / tanh / mask is fp32 on
    both sides.  It answers ONE question: how far does bf16 STORAGE alone move the gradients of this non-smooth network?"""
    sz = dcunet_sizes(model_complexity, model_depth, x.shape[1])
    real, imag = x[..., 0], x[..., 1]
    h = sim.act(x.transpose(2, 3))
    xs = []
    for i in range(sz["n"]):
        xs.append(h)
        h = complex_conv2d(h, p, f"encoder{i}.conv.", sz["enc_s"][i], sz["enc_p"][i], sim)
        h = complex_batchnorm2d(h, p, f"encoder{i}.bn.", training, stats_out)
        h = sim.act(_lrelu(h, f"encoder{i}", act_masks))
        if taps is not None:
            taps[f"encoder{i}"] = h
    q = h
    for i in range(sz["n"]):
        q = complex_conv_transpose2d(q, p, f"decoder{i}.transconv.", sz["dec_s"][i], sz["dec_p"][i], sim)
        q = complex_batchnorm2d(q, p, f"decoder{i}.bn.", training, stats_out)
        q = sim.act(_lrelu(q, f"decoder{i}", act_masks))
        if taps is not None:
            taps[f"decoder{i}"] = q
        if i == sz["n"] - 1:
            break
        q = torch.cat([q, xs[sz["n"] - 1 - i]], dim=1)
    mask = torch.tanh(complex_conv2d(q, p, "linear.", 1, 0)).transpose(2, 3)
    mr, mi = mask[..., 0], mask[..., 1]
    if masking_mode == "E":   # polar mask, as DCCRN's (src/model/dcunet.py:136-155)
        x_mag = torch.sqrt(real ** 2 + imag ** 2 + 1e-8)
        x_phase = torch.atan2(imag, real)
        mm = (mr ** 2 + mi ** 2) ** 0.5
        mask_phase = torch.atan2(mi / (mm + 1e-8), mr / (mm + 1e-8))
        est_mags = torch.tanh(mm) * x_mag
        est_phase = x_phase + mask_phase
        real, imag = est_mags * torch.cos(est_phase), est_mags * torch.sin(est_phase)
    elif masking_mode == "C":
        real, imag = real * mr - imag * mi, real * mi + imag * mr
    elif masking_mode == "R":
        real, imag = real * mr, imag * mi
    return torch.stack([real, imag], dim=-1)


def is_trainable(key: str) -> bool:
    return key.endswith((".weight", ".bias")) and not key.startswith(("encoders.", "decoders."))
