"""Which parameter gradients differ between two identical forward + backward passes under the deterministic schedule?
    python tools/dev/det_diff.py demucs|convtasnet [repeats]
Prints, per repeat, the names of the tensors whose gradient bits differ from the first pass (none = bit-identical)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speech-enhancement-pytorch_amd"))
from sehip.loss import loss_sisdr  # noqa: E402
from sehip.utils import set_deterministic  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "demucs"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    set_deterministic(True)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    if which == "demucs":
        from sehip.model import Demucs
        model = Demucs(sources=["clean"], audio_channels=2).cuda().train()
        clean = 0.1 * torch.randn(2, 1, 2, 24000, generator=g)
        mix = clean[:, 0] + 0.05 * torch.randn(2, 2, 24000, generator=g)
    else:
        from sehip.model import ConvTasNet
        model = ConvTasNet(sources=["a", "b"], audio_channels=1).cuda().train()
        clean = 0.1 * torch.randn(4, 2, 1, 16000, generator=g)
        mix = clean.sum(1)
    mix, clean = mix.cuda(), clean.cuda()
    names = [n for n, _ in model._params]
    first = None
    prev = None
    for r in range(reps):
        model.flat_grads.zero_()
        est = model(mix)
        loss = loss_sisdr(est, clean)
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in model._params}
        out = est.detach().clone()
        ws = model.workspace(mix.shape[0], mix.shape[-1])
        snap = {k: v.t.detach().clone() for k, v in ws.bufs.items()}
        for k in ("gpack", "sums", "stats", "bsums"):
            if hasattr(ws, k):
                snap["ws." + k] = getattr(ws, k).detach().clone()
        if first is None:
            first = (grads, out, float(loss), snap)
            continue
        badb = [k for k in snap if not torch.equal(snap[k], first[3][k])]
        print(f"pass {r}: buffers that differ: {badb}")
        if "ws.sums" in badb:
            d = (snap["ws.sums"] != first[3]["ws.sums"]).nonzero()
            names_n = list(ws.norm_idx) if hasattr(ws, "norm_idx") else []
            for ix in d[:12].tolist():
                a_, b_ = snap["ws.sums"][tuple(ix)].item(), first[3]["ws.sums"][tuple(ix)].item()
                print(f"pass {r}:    sums{ix} ({names_n[ix[0]] if names_n else ''}) {a_!r} vs {b_!r}  rel {abs(a_ - b_) / (abs(b_) + 1e-300):.2e}")
        if prev is not None:
            print(f"pass {r}: buffers that differ from the previous pass: {[k for k in snap if not torch.equal(snap[k], prev[k])]}")
        prev = snap
        bad = [n for n in names if not torch.equal(grads[n], first[0][n])]
        print(f"pass {r}: loss {float(loss)!r} vs {first[2]!r}; output equal {torch.equal(out, first[1])}; {len(bad)} of {len(names)} gradients differ")
        for n in bad[:40]:
            d = (grads[n] - first[0][n]).abs().max().item()
            print(f"   {n:60s} max |delta| {d:.3e} of {first[0][n].abs().max().item():.3e}")


main()
