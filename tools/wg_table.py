#!/usr/bin/env python3
"""Standalone time, kernel and HBM-ideal time of every weight-gradient launch of the B=32 step (side-stream budget)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import torch, ctypes as C
from sehip.model import DCCRN
from sehip._lib import call, stream, lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = DCCRN(length=32000).to(dev).train()
x = (0.1 * torch.randn(32, 1, 32000)).to(dev)
out = model(x); out.backward(torch.randn_like(out) * 1e-3); torch.cuda.synchronize()
ws = model.workspace(32, 32000)
which = sys.argv[1] if len(sys.argv) > 1 else "wg"
tot = 0.0
L = lib()
L.sehip_last_kernel.restype = C.c_char_p
for name, d in ws.desc.items():
    isw = name.endswith(".wg")
    if (which == "wg") != isw: continue
    if not isw and not d.W: continue
    fn = "sehip_wgrad" if isw else "sehip_gemm"
    for _ in range(3): call(fn, C.byref(d), stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call(fn, C.byref(d), stream())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    k = L.sehip_last_kernel().decode()
    srcC = d.src[0].C + (d.src[1].C if d.src[1].ptr else 0)
    # bytes: dOut/out (M x N bf16) + input rows (M rows x srcC x fmul-ish; approximation: M*srcC*2/ (1 if conv stride... ))
    out_b = d.M * d.N * 2
    gf = 2.0 * d.M * d.N * d.K / 1e9
    tot += us
    print(f"{name:16s} {us:7.1f} us  M={d.M:8d} N={d.N:4d} K={d.K:5d} srcC={srcC:4d} J={d.J:3d} {gf:6.1f} GF {gf/us*1e3:6.0f} TF/s  out {out_b/1e6:6.1f} MB ({out_b/us/1e3:5.0f} GB/s)  {k}")
print("total us", tot)
