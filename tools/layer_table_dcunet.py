#!/usr/bin/env python3
"""Per-product table of the DCUnet-10 step (C2: B = 64, 257 x 257 spectra): kernel, us alone after a cache flush, algorithmic GFLOP and
TFLOP/s, operand bytes (sources + destinations once) and the time those bytes need at 5 TB/s.  The DCCRN twin is tools/layer_table.py."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from sehip import distrib
from sehip.solver import Solver, ScalarLog
from sehip._lib import call, stream, lib
B = int(os.environ.get("B", "64"))
cfg = bench.dcunet_config()
torch.manual_seed(cfg.seed)
model = distrib.get_model(cfg.model)
solver = Solver(cfg, model, distrib.get_optimizer(cfg.optim, model), distrib.get_loss_function(cfg.optim), device="gpu", writer=ScalarLog())
noisy, clean = bench.workload_batch("dcunet", B, 0, solver.device)
solver.train_step(*solver._prepare_batch(noisy, clean)); torch.cuda.synchronize()
ws = model.workspace(B, 257, 257)
flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device="cuda")
rows = []
for name, d in ws.desc.items():
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
    by = 0
    for q in range(2):
        if d.src[q].ptr: by += B * d.src[q].T * d.src[q].F * d.src[q].C * 2
        if d.dst[q].ptr: by += B * d.dst[q].T * d.dst[q].F * d.dst[q].C * (4 if d.dst[q].is_f32 else 2)
    rows.append((us, name, k, gf, by, d.M, d.N, d.K))
tot = {"fwd/dgrad": 0.0, "wgrad": 0.0}
for us, name, k, gf, by, M, N, K in sorted(rows, key=lambda r: -r[0]):
    tot["wgrad" if name.endswith(".wg") else "fwd/dgrad"] += us
    print(f"{name:16s} {k[:40]:40s} {us:8.1f} us {gf:7.1f} GF {gf / us:7.1f} TF/s {by / 1e6:7.1f} MB hbm@5TB/s {by / 5e6:6.1f} us  M={M} N={N} K={K}")
print("total us", tot)
