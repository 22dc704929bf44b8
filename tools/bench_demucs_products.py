#!/usr/bin/env python3
"""Times every product of the Demucs step (C3 shape by default) singly: name, kernel, M x N x K, microseconds, TFLOP/s and the
bytes-per-second the launch would need if it only moved its operands once (to spot launches far from both roofs).
    python tools/bench_demucs_products.py [--batch 16] [--samples 96000] [--top 40]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
from sehip._lib import call, stream, lib  # noqa: E402
from sehip.model import Demucs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--samples", type=int, default=96000)
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
model = Demucs(sources=["clean"], audio_channels=2).cuda().train()
x = 0.1 * torch.randn(a.batch, 2, a.samples, device="cuda")
y = model(x)
y.backward(torch.randn_like(y) * 1e-3)
torch.cuda.synchronize()
ws = model.workspace(a.batch, a.samples)
flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device="cuda")
rows = []
for name, d in ws.desc.items():
    wg = name.endswith(".wg")
    fn = "sehip_wgrad" if wg else "sehip_gemm"
    p = ws.st.prods[name[:-3] if wg else name]
    ts = []
    for _ in range(3):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if wg:
            ws._launch_wgrad(name[:-3], stream())      # the streaming dense-row kernel where the workspace bound one
        else:
            call(fn, C.byref(d), stream())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts)[1]
    kern = lib().sehip_last_kernel().decode()
    flops = 2.0 * d.M * p.N * p.K
    byts = 2.0 * d.M * (d.src[0].C if not p.src[1] else d.src[0].C / 4) + (4.0 if d.dst[0].is_f32 else 2.0) * d.M * p.N + 2.0 * p.N * p.K
    if wg:
        byts += 4.0 * p.N * p.K
    rows.append((us, name, kern, d.M, p.N, p.K, flops / us / 1e6, byts / us / 1e6))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(rows)} products, {tot / 1e3:.2f} ms in all (single launches, caches flushed)")
for us, name, kern, M, N, K, tf, tb in rows[:a.top]:
    print(f"{us:8.1f} us  {name:22s} {kern:34s} M {M:7d} N {N:5d} K {K:6d}  {tf:7.1f} TFLOP/s  {tb:5.2f} TB/s")
