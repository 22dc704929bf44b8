#!/usr/bin/env python3
"""Runs selected launches of the DCCRN step in isolation (for rocprofv3 --pmc / --kernel-trace):
    python3 tools/prof_one.py dec1.dg dec0.fwd0.wg ...   [--reps 5] [--batch 32]
Names are keys of DCCRNWorkspace.desc ('.wg' suffix = weight-gradient launch), or 'lstm_fwd' / 'lstm_bwd' / 'bn'."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sehip.model import DCCRN  # noqa: E402


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = DCCRN(length=32000).to(dev).train()
    g = torch.Generator().manual_seed(0)
    clean = 0.1 * torch.randn(batch, 1, 32000, generator=g)
    noisy = (clean + 0.05 * torch.randn(batch, 1, 32000, generator=g)).to(dev)
    # one full step so that every buffer holds realistic (random-data) values
    out = model(noisy)
    out.backward(torch.randn_like(out) * 1e-3)
    torch.cuda.synchronize()
    ws = model.workspace(batch, 32000)
    for name in names:
        for _ in range(reps):
            if name.endswith(".wg"):
                ws.wgrad(name[:-3])
            else:
                ws.gemm(name)
        torch.cuda.synchronize()
    print("done", names)


if __name__ == "__main__":
    main()
