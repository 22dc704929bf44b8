#!/usr/bin/env python3
"""Debug: sehip_cbn_bwd_fused (reduce into replica rows + finalize inside the apply pass) against the three separate launches on the
same inputs, repeated; reports NaNs / differences per layer shape."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
from sehip._lib import call, ptr, stream
dev = torch.device("cuda:0")
B, T = 32, 323
NREP = int(os.environ.get("SEHIP_BWD_REPLICAS", "8"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
side = torch.cuda.Stream(device=dev)
noise = torch.randn(64 * 1024 * 1024, device=dev)
for F, cr in [(128, 8), (64, 16), (32, 32), (16, 64), (8, 128), (4, 128)]:
    rows, C = B * T * F, 2 * cr
    torch.manual_seed(cr + F)
    y = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    dz = (0.01 * torch.randn(rows, C, device=dev)).to(torch.bfloat16)
    dy0 = torch.empty_like(y); dy1 = torch.empty_like(y)
    acc = torch.zeros(512 * (6 * cr + 1), device=dev)
    coef = torch.zeros(cr, 16, device=dev); bcoef = torch.zeros(cr, 16, device=dev)
    w = [torch.full((cr,), v, device=dev) for v in (0.7071, 0.0, 0.7071, 0.0, 0.0)]
    run = [torch.zeros(cr, device=dev), torch.zeros(cr, device=dev), torch.full((cr,), 0.7071, device=dev), torch.zeros(cr, device=dev),
           torch.full((cr,), 0.7071, device=dev)]
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    g0 = [torch.zeros(cr, device=dev) for _ in range(5)] + [torch.zeros(1, device=dev)]
    g1 = [torch.zeros(cr, device=dev) for _ in range(5)] + [torch.zeros(1, device=dev)]
    slope = torch.full((1,), 0.25, device=dev)
    s = stream()
    call("sehip_cbn_stats", ptr(y), rows, cr, ptr(acc), s)
    call("sehip_cbn_finalize", ptr(acc), *[ptr(t) for t in w], *[ptr(t) for t in run], ptr(nbt), rows, cr, 1e-5, 0.1, 1, ptr(coef), s)
    call("sehip_cbn_bwd_reduce", ptr(dz), None, ptr(y), ptr(coef), ptr(slope), rows, cr, F, T, 0, ptr(acc), s)
    call("sehip_cbn_bwd_finalize", ptr(acc), ptr(coef), ptr(w[0]), ptr(w[1]), ptr(w[2]), rows, cr, *[ptr(t) for t in g0], ptr(bcoef), s)
    call("sehip_cbn_bwd_apply", ptr(dz), None, ptr(y), ptr(coef), ptr(bcoef), ptr(slope), rows, cr, F, T, 0, ptr(dy0), s)
    torch.cuda.synchronize()
    rep = torch.zeros(2, NREP * (6 * cr + 1), device=dev)
    bad = 0
    worst = 0.0
    for it in range(reps):
        turn = it & 1
        if it % 3 == 0:      # noise on a second stream (HBM / L2 pressure beside the passes)
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        call("sehip_cbn_bwd_fused", ptr(dz), None, ptr(y), ptr(coef), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(slope), rows, cr, F, T, 0,
             ptr(rep[turn]), ptr(rep[turn ^ 1]), NREP, *[ptr(t) for t in g1], ptr(dy1), s)
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(dy1.float()).all()) and all(bool(torch.isfinite(t).all()) for t in g1)
        d = float((dy1.float() - dy0.float()).norm() / dy0.float().norm()) if fin else float("nan")
        gd = max(float((a - b_).norm() / (b_.norm() + 1e-20)) for a, b_ in zip(g1, g0)) if fin else float("nan")
        if not fin or d > 1e-2 or gd > 1e-3:
            bad += 1
            if bad <= 3:
                print(f"  F={F} Cr={cr} it={it} turn={turn}: finite={fin} dy rel {d:.3e} grads rel {gd:.3e}; rep used finite {bool(torch.isfinite(rep[turn]).all())} "
                      f"|rep used| {float(rep[turn].abs().sum()):.3e} |rep next| {float(rep[turn ^ 1].abs().sum()):.3e}")
        else:
            worst = max(worst, d, gd)
    print(f"F={F} Cr={cr} rows={rows}: {bad} bad of {reps}; worst good rel {worst:.2e}")
