"""CPU oracle for ConvTasNet forward / loss / gradients (BASELINE config C4) -- TEST INFRASTRUCTURE ONLY.

Functional fp32 PyTorch-CPU restatement of the reference's ConvTasNet with the shipped options (``skip=False``, ``norm_type
="gLN"``, non-causal, ``mask_nonlinear="relu"``): reference src/model/conv_tasnet.py:34-154 (ConvTasNet), :157-176 (Encoder),
:179-204 + :11-31 (Decoder, overlap_and_add), :209-304 (TemporalConvNet), :307-349 (TemporalBlock), :352-402
(DepthwiseSeparableConv), :439-487 (cLN / gLN).  Parameters live in one dict keyed by the reference's state_dict names
(``separator.network.2.{r}.{x}.net.0.weight`` ...).  It shares no code with the reference.  Only tests/ may import it.

Parity pinning: tests/test_oracle_golden.py::test_convtasnet_oracle_matches_reference checks block activations, the
separated sources, an SI-SNR loss and every parameter gradient against tests/golden/convtasnet_tiny.npz, which
oracle/gen_golden_convtasnet.py produced by importing the real reference in the build container.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .dccrn_oracle import Bf16Sim, NoSim   # bf16 round-trips at the HIP path's storage points (tests only)

EPS = 1e-8   # src/model/conv_tasnet.py:207


def cln(y, gamma, beta):
    """Channel-wise layer norm over the channel dim of [M, N, K] (src/model/conv_tasnet.py:439-462)."""
    mean = y.mean(dim=1, keepdim=True)
    var = y.var(dim=1, keepdim=True, unbiased=False)
    return gamma * (y - mean) / torch.pow(var + EPS, 0.5) + beta


def gln(y, gamma, beta):
    """Global layer norm over (channel, time) of [M, N, K] (src/model/conv_tasnet.py:465-487)."""
    mean = y.mean(dim=(1, 2), keepdim=True)
    var = ((y - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return gamma * (y - mean) / torch.pow(var + EPS, 0.5) + beta


def overlap_and_add(frames, step):
    """[..., K, L] frames at hop `step` -> [..., step*(K-1)+L] (src/model/conv_tasnet.py:11-31, written as a fold)."""
    lead = frames.shape[:-2]
    k, l = frames.shape[-2:]
    x = frames.reshape(-1, k, l).transpose(1, 2)                       # [*, L, K]
    out = F.fold(x, output_size=(1, step * (k - 1) + l), kernel_size=(1, l), stride=(1, step))
    return out.reshape(*lead, -1)


def _prelu(h, a, mask):
    """nn.PReLU() (one slope).  With ``mask`` (tests only: a boolean tensor, True = the identity branch) the branch of every element is
    GIVEN instead of taken from the sign of h, so that a backward pass can be compared through the same branches as the run the mask
    came from (near-zero pre-activations change sign under a 1 % forward error of a bf16 path)."""
    if mask is None:
        return F.prelu(h, a)
    return h * torch.where(mask, torch.ones((), dtype=h.dtype), a.reshape(()))


def temporal_block(x, p, pre, dilation, P, masks=None, sim=NoSim):
    """1x1 B->H, PReLU, gLN, depthwise dilated conv (groups=H, 'same' padding), PReLU, gLN, 1x1 H->B, + residual
    (src/model/conv_tasnet.py:307-402 with skip=False).  masks: (mask1, mask2) for the two PReLUs, see _prelu.
    sim=Bf16Sim: bf16 round-trips of the five tensors the HIP path stores per block (1x1 output, PReLU + gLN output, depthwise
    output, PReLU + gLN output, block output) and of the two 1x1 weights."""
    h = sim.act(F.conv1d(x, sim.weight(p[pre + "net.0.weight"])))
    h = _prelu(h, p[pre + "net.1.weight"], None if masks is None else masks[0])
    h = sim.act(gln(h, p[pre + "net.2.gamma"], p[pre + "net.2.beta"]))
    q = pre + "net.3."
    pad = (P - 1) * dilation // 2
    h = sim.act(F.conv1d(h, p[q + "net.0.weight"], padding=pad, dilation=dilation, groups=h.shape[1]))
    h = _prelu(h, p[q + "net.1.weight"], None if masks is None else masks[1])
    h = sim.act(gln(h, p[q + "net.2.gamma"], p[q + "net.2.beta"]))
    return sim.act(F.conv1d(h, sim.weight(p[q + "pointwise_conv.weight"])) + x)


def convtasnet_forward(p, mixture, C=2, N=128, L=40, B=128, H=256, P=3, X=7, R=2, audio_channels=1, taps=None, act_masks=None,
                       sim=NoSim, mask_nonlinear="relu"):
    """mixture [M, ac, T] -> separated sources [M, C, ac, T] (src/model/conv_tasnet.py:136-154).
    act_masks (tests only): {"block{r}.{i}": (mask1, mask2), "mask": mask} -- given branches of the PReLUs / the mask ReLU.
    sim=Bf16Sim (tests only): bf16 storage at the HIP path's layer boundaries (cLN output, bottleneck, the five tensors of every
    temporal block, the mask scores) and bf16 1x1 weights; the encoder output w and the decoder stay fp32 on both sides."""
    w = F.relu(F.conv1d(mixture, p["encoder.conv1d_U.weight"], stride=L // 2))           # [M, N, K]
    net = "separator.network."
    x = sim.act(cln(w, p[net + "0.gamma"], p[net + "0.beta"]))
    x = sim.act(F.conv1d(x, sim.weight(p[net + "1.weight"])))
    if taps is not None:
        taps["bottleneck"] = x
    for r in range(R):
        for i in range(X):
            x = temporal_block(x, p, f"{net}2.{r}.{i}.", 2 ** i, P, None if act_masks is None else act_masks[f"block{r}.{i}"], sim)
            if taps is not None:
                taps[f"block{r}.{i}"] = x
    m, n, k = w.shape
    score = sim.act(F.conv1d(x, sim.weight(p[net + "3.weight"]))).view(m, C, n, k)
    if mask_nonlinear == "softmax":         # src/model/conv_tasnet.py:298-299: over the sources (the HIP path stores it in bf16)
        mask = sim.act(F.softmax(score, dim=1))
    else:
        mask = F.relu(score) if act_masks is None else score * act_masks["mask"].to(score.dtype)
    src_w = (w.unsqueeze(1) * mask).transpose(2, 3)                                        # [M, C, K, N]
    est = F.linear(src_w, p["decoder.basis_signals.weight"])                               # [M, C, K, ac*L]
    est = est.view(m, C, k, audio_channels, L).transpose(2, 3)
    est = overlap_and_add(est, L // 2)                                                     # [M, C, ac, T']
    return F.pad(est, (0, mixture.shape[-1] - est.shape[-1]))


def is_trainable(key: str) -> bool:
    return True   # ConvTasNet has parameters only (gLN / cLN: no running statistics)
