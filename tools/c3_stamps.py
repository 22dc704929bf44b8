#!/usr/bin/env python3
"""Where a conv_gemm_v3 launch spends its cycles: runs single launches of named DCCRN products (B = 32, 2-s clips) on a library built
with -DC3_STAMPS (python tools/build_variant.py stamps conv3.hip -DC3_STAMPS) and summarises the per-wave stamp records:
prologue / K loop / epilogue cycles per workgroup, the K step split into issue (fragment reads + DMA issue + 28 MFMAs), DMA wait,
barrier wait and tail (4 MFMAs + loop overhead), workgroup start / end times on the 100 MHz clock, co-residency per CU.

    SEHIP_LIB=tools/_var_stamps.so python tools/c3_stamps.py enc4.fwd dec0.dg ...
"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SEHIP_LIB", os.path.join(ROOT, "tools", "_var_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from sehip.model import DCCRN
from sehip._lib import call, stream, lib, LIB_PATH

names = sys.argv[1:] or ["enc3.fwd", "enc4.fwd", "enc5.fwd", "dec0.dg", "dec0.fwd0", "enc3.dg0"]
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = DCCRN(length=32000).to(dev).train()
x = (0.1 * torch.randn(32, 1, 32000)).to(dev)
out = model(x); out.backward(torch.randn_like(out) * 1e-3); torch.cuda.synchronize()
ws = model.workspace(32, 32000)
raw = C.CDLL(LIB_PATH)
NW = 8192 * 8
buf = torch.zeros(2 * NW * 16, dtype=torch.int32, device=dev)
assert raw.sehip_c3_set_stamps(C.c_void_p(buf.data_ptr())) == 0
for name in names:
    d = ws.desc[name]
    for _ in range(3): call("sehip_gemm", C.byref(d), stream())
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call("sehip_gemm", C.byref(d), stream()); e1.record(); torch.cuda.synchronize()
    k = lib().sehip_last_kernel().decode()
    rall = buf.cpu().numpy().view(np.uint32).reshape(-1, 16)
    keep = rall[:NW, 15] == 0x5eed
    r, r2 = rall[:NW][keep], rall[NW:][keep]
    if not len(r):
        print(name, k, "no stamps"); continue
    nwv = 4
    w0 = r[::1]
    t0 = int(r[:, 3].min())
    start = (r[:, 3].astype(np.int64) - t0) / 100.0      # us
    end = (r[:, 4].astype(np.int64) - t0) / 100.0
    nwg = len(np.unique(r[:, 0]))
    cu = (r[:, 2].astype(np.int64) << 16) | (r[:, 1] & 0xff00) | ((r[:, 1] >> 13) & 7) << 20       # xcc | cu_id,sh | se
    ncu = len(np.unique(cu))
    pro, loop, epi, n = r[:, 5].astype(float), r[:, 6].astype(float), r[:, 7].astype(float), r[:, 12].astype(float)
    wall_us = end - start
    clk = (pro + loop + epi) / np.maximum(wall_us, 1e-3) / 1e3     # GHz
    print(f"== {name}  {k}  event {e0.elapsed_time(e1) * 1e3:.1f} us, stamped span {end.max():.1f} us, {nwg} workgroups on {ncu} CUs, "
          f"{int(n[0])} K steps, clock {np.median(clk):.2f} GHz")
    first = start < np.percentile(start, 40)
    for tag, m in (("first-round workgroups", first), ("late workgroups", ~first)):
        if not m.any(): continue
        q = lambda a: f"{np.median(a[m]):8.0f}"
        print(f"   {tag:24s} n={m.sum() // nwv:4d}  start {np.median(start[m]):6.1f} us  life {np.median(wall_us[m]):6.1f} us | cycles: prologue{q(pro)} "
              f"loop{q(loop)} epilogue{q(epi)} | per step: issue{q(r[:, 8] / n)} dma-wait{q(r[:, 9] / n)} barrier{q(r[:, 10] / n)} "
              f"tail{q(r[:, 11] / n)}  (max issue{q(r[:, 13])} max barrier{q(r[:, 14])})")
    med = lambda a: f"{np.median(a.astype(float)):7.0f}"
    print(f"   prologue: table{med(r2[:, 0])} offsets{med(r2[:, 1])} dma issue{med(r2[:, 2])} first wait{med(r2[:, 3])} | epilogue: staging+sums{med(r2[:, 4])} "
          f"store loop{med(r2[:, 5])} drain{med(r2[:, 6])}")
    # timeline: how many workgroups are alive over time
    edges = np.linspace(0, end.max(), 9)
    alive = [int(((start <= t) & (end > t)).sum() // nwv) for t in edges[:-1]]
    print("   workgroups alive at", " ".join(f"{t:.0f}us:{a}" for t, a in zip(edges[:-1], alive)))
