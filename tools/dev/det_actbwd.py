"""sehip_dmx_act_bwd under the deterministic schedule, the same call repeated beside traffic on a second stream: are `sums` / dy / gch
bit-stable?   python tools/dev/det_actbwd.py [calls] [noside]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speech-enhancement-pytorch_amd"))
from sehip import _lib  # noqa: E402
from sehip.utils import set_deterministic  # noqa: E402

BF = torch.bfloat16


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    noside = len(sys.argv) > 2
    set_deterministic(True)
    L = _lib
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    for (B, T, C, G, mode, scaled) in ((2, 12007, 128, 1, 1, True), (2, 12007, 16, 1, 0, False), (2, 3001, 256, 1, 1, True), (16, 24000, 128, 1, 1, True)):
        Co = C // 2 if mode else C
        y = (torch.randn(B, T, C, generator=g) * 1.5 + 0.3).to(BF).to(dev)
        dz = (1e-3 * torch.randn(B, T, Co, generator=g)).to(BF).to(dev)
        gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
        scale = (0.3 + 0.2 * torch.randn(Co, generator=g)).to(dev) if scaled else None
        stats = torch.zeros(B, 8, 2, dtype=torch.float64, device=dev)
        L.call("sehip_dmx_gn_stats", y.data_ptr(), B, T, C, G, stats.data_ptr(), None)
        side = torch.cuda.Stream()
        big = torch.randn(64 * 1024 * 1024, device=dev)
        big2 = torch.empty_like(big)
        ref = None
        nbad = 0
        for it in range(calls):
            sums = torch.zeros(B, 8, 2, dtype=torch.float64, device=dev)
            gch = torch.zeros(2 * C + Co, device=dev)
            dy = torch.zeros(B, T, C, dtype=BF, device=dev)
            if not noside:
                with torch.cuda.stream(side):
                    for _ in range(2):
                        big2.copy_(big)
            L.call("sehip_dmx_act_bwd", dz.data_ptr(), y.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), G, 1e-5, mode,
                   scale.data_ptr() if scaled else None, B, T, C, sums.data_ptr(), gch.data_ptr(), dy.data_ptr(), None)
            torch.cuda.synchronize()
            cur = (sums.clone(), gch.clone(), dy.clone())
            if ref is None:
                ref = cur
                continue
            eq = [torch.equal(a, b) for a, b in zip(cur, ref)]
            if not all(eq):
                nbad += 1
                if nbad <= 3:
                    d = (cur[0] != ref[0]).nonzero().tolist()
                    print(f"  call {it}: sums / gch / dy equal: {eq}; sums entries {d[:4]}: "
                          f"{[(cur[0][tuple(i)].item(), ref[0][tuple(i)].item()) for i in d[:2]]}")
        print(f"B={B} T={T} C={C} G={G} mode={mode} scaled={scaled}: {nbad} of {calls - 1} calls differ from the first")


main()
