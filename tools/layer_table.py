#!/usr/bin/env python3
"""Per-product table of the DCCRN step (B=32, 2-s clips): kernel, us (single launches after a cache flush, as bench.py's roofline
pass), algorithmic GFLOP and TFLOP/s, operand bytes (sources + destinations once) and the HBM time those bytes need at 5 TB/s."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from sehip._lib import call, stream, lib
solver, model, mixture, sources = bench.build_for_profile()
solver.train_step(mixture, sources); torch.cuda.synchronize()
ws = model.workspace(32, 32000)
flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device="cuda")
rows = []
for name, d in ws.desc.items():
    if not name.endswith(".wg") and not d.W:
        continue
    fn = "sehip_wgrad" if name.endswith(".wg") else "sehip_gemm"
    call(fn, C.byref(d), stream()); k = lib().sehip_last_kernel().decode(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(fn, C.byref(d), stream()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts)[1]
    gf = 2.0 * d.M * bench._weight_entries(ws, name) / 1e9
    B = d.M // (d.TT * d.J)
    by = 0
    for q in range(2):
        if d.src[q].ptr: by += B * d.src[q].T * d.src[q].F * d.src[q].C * 2
        if d.dst[q].ptr: by += B * d.dst[q].T * d.dst[q].F * d.dst[q].C * (4 if d.dst[q].is_f32 else 2)
    rows.append((us, name, k, gf, by))
tot = 0
for us, name, k, gf, by in sorted(rows, key=lambda r: -r[0]):
    tot += us
    print(f"{name:14s} {k[:44]:44s} {us:7.1f} us {gf:7.1f} GF {gf / us * 1e-3 * 1e3:7.1f} TF/s  {by / 1e6:7.1f} MB  hbm@5TB/s {by / 5e6:6.1f} us  M={d.M if False else ''}")
print("total us", tot)
