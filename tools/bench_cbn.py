"""Stand-alone timing of the four streaming ComplexBatchNorm passes at the C1 layer shapes (GPU box only).

    python tools/bench_cbn.py [--reps 30]

Knobs are read by libsehip from the environment (SEHIP_CBN_CH, SEHIP_CBN_U, SEHIP_CBN_BLOCKS, SEHIP_CBN_APPLY_ROWS ...).
Buffers rotate through enough copies that no launch finds its input in L2 / MALL.  Prints one JSON line.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
from sehip._lib import call, ptr, stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=323)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = [(128, 8), (64, 16), (32, 32), (16, 64), (8, 128), (4, 128)]
    out = {"env": {k: v for k, v in os.environ.items() if k.startswith("SEHIP_CBN")}, "layers": []}
    tot = dict(stats=0.0, apply=0.0, bwd_reduce=0.0, bwd_apply=0.0)
    for F, cr in shapes:
        rows, C = a.B * a.T * F, 2 * cr
        nbytes = rows * C * 2
        ncopy = max(2, int(600e6 // (3 * nbytes)) + 1)
        ys = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(ncopy)]
        dzs = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(ncopy)]
        zs = [torch.empty(rows, C, device=dev, dtype=torch.bfloat16) for _ in range(ncopy)]
        need = call("sehip_cbn_scratch_floats", rows, cr) if False else 512 * (6 * cr + 1)
        acc = torch.zeros(need, device=dev)
        coef = torch.zeros(cr, 16, device=dev)
        bcoef = torch.zeros(cr, 16, device=dev)
        w = [torch.full((cr,), v, device=dev) for v in (0.7071, 0.0, 0.7071, 0.0, 0.0)]
        run = [torch.zeros(cr, device=dev), torch.zeros(cr, device=dev), torch.full((cr,), 0.7071, device=dev),
               torch.zeros(cr, device=dev), torch.full((cr,), 0.7071, device=dev)]
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        g = [torch.zeros(cr, device=dev) for _ in range(5)] + [torch.zeros(1, device=dev)]
        slope = torch.full((1,), 0.25, device=dev)
        s = stream()
        call("sehip_cbn_stats", ptr(ys[0]), rows, cr, ptr(acc), s)
        call("sehip_cbn_finalize", ptr(acc), *[ptr(t) for t in w], *[ptr(t) for t in run], ptr(nbt), rows, cr, 1e-5, 0.1, 1,
             ptr(coef), s)
        call("sehip_cbn_bwd_reduce", ptr(dzs[0]), None, ptr(ys[0]), ptr(coef), ptr(slope), rows, cr, F, a.T, 0, ptr(acc), s)
        call("sehip_cbn_bwd_finalize", ptr(acc), ptr(coef), ptr(w[0]), ptr(w[1]), ptr(w[2]), rows, cr, *[ptr(t) for t in g],
             ptr(bcoef), s)
        ops = {
            "stats": (lambda i: call("sehip_cbn_stats", ptr(ys[i]), rows, cr, ptr(acc), s), nbytes),
            "apply": (lambda i: call("sehip_cbn_apply", ptr(ys[i]), ptr(coef), ptr(slope), rows, cr, ptr(zs[i]), s), 2 * nbytes),
            "bwd_reduce": (lambda i: call("sehip_cbn_bwd_reduce", ptr(dzs[i]), None, ptr(ys[i]), ptr(coef), ptr(slope), rows, cr,
                                          F, a.T, 0, ptr(acc), s), 2 * nbytes),
            "bwd_apply": (lambda i: call("sehip_cbn_bwd_apply", ptr(dzs[i]), None, ptr(ys[i]), ptr(coef), ptr(bcoef), ptr(slope),
                                         rows, cr, F, a.T, 0, ptr(zs[i]), s), 3 * nbytes),
        }
        rec = {"F": F, "Cr": cr, "MB": round(nbytes / 1e6, 1)}
        for name, (fn, traffic) in ops.items():
            for i in range(3):
                fn(i % ncopy)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps):
                fn(i % ncopy)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            rec[name] = [round(us, 1), round(traffic / us / 1e6, 2)]   # us, TB/s
            tot[name] += us
        out["layers"].append(rec)
        del ys, dzs, zs
    out["sum_us"] = {k: round(v, 1) for k, v in tot.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
