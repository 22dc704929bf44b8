#!/usr/bin/env python3
"""tests/golden/convtasnet_tiny.npz from the IMPORTED reference (build container only).

A small ConvTasNet (N=16, L=8, B=16, H=32, P=3, X=3, R=2, two speakers, mono, the shipped skip=False / gLN / relu options,
src/model/conv_tasnet.py:34-154) on a [2, 1, 404] mixture: state_dict, input / targets, the bottleneck and every temporal
block's output, the separated sources [2, 2, 1, 404], the reference's SI-SNR loss (src/loss.py:14-29) and every parameter
gradient.  convtasnet_tiny_softmax.npz: the same with mask_nonlinear='softmax' (:298-299: F.softmax over the sources).
Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_convtasnet.py"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
KW = dict(N=16, L=8, B=16, H=32, P=3, X=3, R=2, audio_channels=1)

from src.model.conv_tasnet import ConvTasNet  # noqa: E402
from src.loss import loss_sisdr  # noqa: E402



def build(out_path, extra):
    torch.manual_seed(7)
    model = ConvTasNet(sources=["None", "None"], **KW, **extra)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():   # non-trivial norm affine terms and PReLU slopes
        for name, prm in model.named_parameters():
            if name.endswith("gamma"):
                prm.copy_(1 + 0.2 * torch.randn(prm.shape, generator=g))
            if name.endswith("beta"):
                prm.copy_(0.1 * torch.randn(prm.shape, generator=g))
            if prm.numel() == 1:
                prm.copy_(0.25 + 0.1 * torch.randn(prm.shape, generator=g))
    mix = 0.3 * torch.randn(2, 1, 404, generator=g)
    # targets = the untrained network's own output + 30 % noise: SI-SNR around +10 dB.  (With targets that are uncorrelated with
    # the output of a randomly initialised network, <est, target> is a cancellation-dominated number -- SI-SNR near -35 dB -- and
    # the loss and every gradient are ill-conditioned: a 1 % change of est moves the loss by ~1 dB.  Useless for a parity check.)
    with torch.no_grad():
        e0 = model(mix)
    tgt = e0 + 0.3 * e0.std() * torch.randn(e0.shape, generator=g)
    out = {"sd." + k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    taps = {}
    net = model.separator.network
    hooks = [net[1].register_forward_hook(lambda m, a, o: taps.__setitem__("bottleneck", o.detach().clone()))]
    for r in range(KW["R"]):
        for i in range(KW["X"]):
            hooks.append(net[2][r][i].register_forward_hook(lambda m, a, o, r=r, i=i: taps.__setitem__(f"block{r}.{i}", o.detach().clone())))
    est = model(mix)
    loss = loss_sisdr(est, tgt)
    loss.backward()
    for k, v in taps.items():
        out["tap." + k] = v.numpy()
    out.update(mix=mix.numpy(), target=tgt.numpy(), est=est.detach().numpy(), loss=np.float32(loss.item()))
    for k, prm in model.named_parameters():
        out["grad." + k] = prm.grad.numpy()
    np.savez_compressed(out_path, **out)
    print("convtasnet golden:", len(out), "entries; est", tuple(est.shape), "loss", loss.item(), os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    build(os.path.join(GOLDEN, "convtasnet_tiny.npz"), {})
    build(os.path.join(GOLDEN, "convtasnet_tiny_softmax.npz"), dict(mask_nonlinear="softmax"))
