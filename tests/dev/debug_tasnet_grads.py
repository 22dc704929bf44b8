#!/usr/bin/env python3
"""Where does the ConvTasNet full-width gradient error come from?  Per residual block: relative error of the residual-stream
gradient dx_b (HIP bf16 buffer vs the oracle's autograd through its taps) and of the block's parameter gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from oracle import convtasnet_oracle as CT
from sehip.model import ConvTasNet

torch.manual_seed(5)
model = ConvTasNet(sources=["None", "None"], audio_channels=1).cuda()
p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(6)
mix = 0.3 * torch.randn(2, 1, 8000, generator=g)
leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
taps = {}
ref = CT.convtasnet_forward(leaves, mix, audio_channels=1, taps=taps)
G = torch.randn(ref.shape, generator=g) / ref.numel() ** 0.5
keys = [k for k in taps]
for k in keys:
    taps[k].retain_grad()
(ref * G).sum().backward()
est = model(mix.cuda())
est.backward(G.cuda())
torch.cuda.synchronize()
ws = model.workspace(2, 8000)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
names = ["bottleneck"] + [f"block{r}.{i}" for r in range(2) for i in range(7)]
for b_, nm in enumerate(names):
    ref_dx = taps[nm].grad            # [M, B, K]
    buf = ws.bufs.get(f"dx{b_}")
    if buf is None:
        continue
    got = buf.t.float().cpu().reshape(2, -1, ref_dx.shape[1]).transpose(1, 2)
    print(f"dx{b_:2d} ({nm:12s}) rel err {rel(got, ref_dx):.3e}  |dx| {float(ref_dx.norm()):.3e}")
got = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
for k in sorted(leaves):
    if leaves[k].grad is None: continue
    print(f"{k:60s} {rel(got[k], leaves[k].grad):.3e}  |g| {float(leaves[k].grad.norm()):.3e}")
