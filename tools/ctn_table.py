#!/usr/bin/env python3
"""Standalone time and kernel of every product launch of the ConvTasNet step (C4 shape): python tools/ctn_table.py [fwd|wg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import torch, ctypes as C
from sehip.model import ConvTasNet
from sehip._lib import call, stream, lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ConvTasNet(sources=["a", "b"], N=128, L=40, B=128, H=256, P=3, X=7, R=2, audio_channels=1, norm_type="gLN", causal=False,
                   mask_nonlinear="relu").to(dev).train()
x = (0.1 * torch.randn(32, 1, 32000)).to(dev)
out = model(x); out.backward(torch.randn_like(out) * 1e-3); torch.cuda.synchronize()
ws = model.workspace(32, 32000)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
L = lib(); L.sehip_last_kernel.restype = C.c_char_p
tot, seen = 0.0, {}
for name, d in ws.desc.items():
    isw = name.endswith(".wg")
    if (which == "wg") != isw: continue
    fn = "sehip_wgrad" if isw else "sehip_gemm"
    key = (d.M, d.N, d.K, bool(d.res), fn)
    for _ in range(3): call(fn, C.byref(d), stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call(fn, C.byref(d), stream())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    if key not in seen:
        seen[key] = True
        mb = (d.M * d.K + d.M * d.N * (2 if d.res else 1)) * 2 / 1e6
        print(f"{name:14s} {us:7.1f} us  M={d.M} N={d.N} K={d.K} res={bool(d.res)}  {mb:6.1f} MB ({mb / us * 1e3 / 1e3:5.2f} TB/s)  {L.sehip_last_kernel().decode()}")
print("total us", round(tot, 1), "launches", sum(1 for n in ws.desc if n.endswith(".wg") == (which == "wg")))
