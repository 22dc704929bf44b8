#!/usr/bin/env python3
"""Times the LocalState attention kernels at the two C3 shapes (B=16: T=187/hid=256 and T=47/hid=512)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
from sehip import _lib as L  # noqa: E402
BF = torch.bfloat16
for B, T, hid in ((16, 187, 256), (16, 47, 512)):
    nq = 3 * hid + 16
    qkv = torch.randn(B, T, nq, device="cuda").to(BF)
    dres = torch.randn(B, T, hid, device="cuda").to(BF)
    out = torch.zeros(B, T, hid, dtype=BF, device="cuda")
    dq = torch.zeros(B, T, nq, dtype=BF, device="cuda")
    slabs = torch.empty(L.lib().sehip_dmx_attn_bwd_scratch_floats(B, T, hid), device="cuda")
    for name, fn in (("fwd", lambda: L.call("sehip_dmx_attn_fwd", qkv.data_ptr(), B, T, hid, 4, 4, nq, out.data_ptr(), None)),
                     ("bwd", lambda: L.call("sehip_dmx_attn_bwd", qkv.data_ptr(), dres.data_ptr(), B, T, hid, 4, 4, nq, slabs.data_ptr(), dq.data_ptr(), None))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B {B} T {T} hid {hid} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
