#!/usr/bin/env python3
"""GroupNorm + GLU / GELU passes of Demucs (csrc/demucs.hip: dmx_gn_stats, dmx_act_fwd, dmx_act_bwd reduce + apply) alone at the C3
shapes (B = 16): microseconds after a cache flush and the bytes each pass has to move at least (bf16 tensors once), to see which of
them are away from the HBM roof on their own and which only in the step."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
import torch
from sehip import _lib
L = _lib
dev = "cuda"
BF = torch.bfloat16
flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device=dev)


def t_us(fn, reps=3):
    ts = []
    for _ in range(reps):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


B = 16
# (level, T, C of the normalised tensor, mode 1 = GLU (C -> C/2), G groups (0 = no GroupNorm))
shapes = [("e0.n4 GLU", 48468, 128, 1, 0), ("e1.n4 GLU", 12116, 256, 1, 0), ("e2.n4 GLU", 3028, 512, 1, 0), ("e3.n4 GLU", 756, 1024, 1, 0),
          ("e4.n4 GLU gn", 188, 2048, 1, 4), ("e5.n4 GLU gn", 46, 4096, 1, 4), ("e0.d0.n1 GELU gn1", 48468, 16, 0, 1),
          ("e2.d0.n1 GELU gn1", 3028, 64, 0, 1), ("e0.d0.n2 GLU gn1 scale", 48468, 128, 1, 1), ("e3.d0.n2 GLU gn1 scale", 756, 1024, 1, 1)]
p = lambda t: None if t is None else t.data_ptr()
for name, T, Cc, mode, G in shapes:
    Co = Cc // 2 if mode else Cc
    y = (torch.randn(B, T, Cc, device=dev) * 1.5).to(BF)
    dz = torch.randn(B, T, Co, device=dev).to(BF)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    scaled = "scale" in name
    scale = torch.full((Co,), 0.3, device=dev) if scaled else None
    resid = torch.randn(B, T, Co, device=dev).to(BF) if scaled else None
    stats = torch.zeros(B, 8, 2, dtype=torch.float64, device=dev)
    sums = torch.zeros(B, 8, 2, dtype=torch.float64, device=dev)
    out = torch.zeros(B, T, Co, dtype=BF, device=dev)
    dy = torch.zeros(B, T, Cc, dtype=BF, device=dev)
    gch = torch.zeros(2 * Cc + Co, device=dev)
    st = L.stream
    tb = y.numel() * 2
    if G:
        L.call("sehip_dmx_gn_stats", y.data_ptr(), B, T, Cc, G, stats.data_ptr(), None)
        us = t_us(lambda: L.call("sehip_dmx_gn_stats", y.data_ptr(), B, T, Cc, G, stats.data_ptr(), None))
        print(f"{name:26s} gn_stats   {us:7.1f} us  {tb / 1e6:7.1f} MB  {tb / us / 1e6:5.2f} TB/s")
    fwd = lambda: L.call("sehip_dmx_act_fwd", y.data_ptr(), p(stats) if G else None, p(gamma) if G else None, p(beta) if G else None, max(G, 1), 1e-5,
                         mode, p(scale), p(resid), None, B, T, Cc, out.data_ptr(), None)
    fwd()
    by = tb + out.numel() * 2 * (2 if scaled else 1)
    us = t_us(fwd)
    print(f"{name:26s} act_fwd    {us:7.1f} us  {by / 1e6:7.1f} MB  {by / us / 1e6:5.2f} TB/s")
    bwd = lambda: L.call("sehip_dmx_act_bwd", dz.data_ptr(), y.data_ptr(), p(stats) if G else None, p(gamma) if G else None, p(beta) if G else None,
                         max(G, 1), 1e-5, mode, p(scale), B, T, Cc, p(sums) if G else None, p(gch) if G else None, dy.data_ptr(), None)
    bwd()
    by = (tb + dz.numel() * 2) * (2 if G else 1) + dy.numel() * 2
    us = t_us(bwd)
    print(f"{name:26s} act_bwd    {us:7.1f} us  {by / 1e6:7.1f} MB  {by / us / 1e6:5.2f} TB/s   (reduce + apply)" if G else
          f"{name:26s} act_bwd    {us:7.1f} us  {by / 1e6:7.1f} MB  {by / us / 1e6:5.2f} TB/s")
