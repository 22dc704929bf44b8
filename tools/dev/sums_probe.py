"""Is anything added to a norm launch's `sums` entries BEFORE its own backward reduce runs?  Stream-ordered 256-byte copies of the entry
right before and right after every sehip_dmx_act_bwd of one full-width Demucs step; compared with a repeat of the same call afterwards.
    python tools/dev/sums_probe.py [passes]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speech-enhancement-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))


def main():
    from sehip.model import Demucs
    from sehip.utils import set_deterministic
    set_deterministic(True)            # (fixed-order sums: an in-step result that differs from the repeat IS a perturbation; SEHIP_DET_FORCE_SIDE=1 keeps the second stream)
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    B, T = 2, 24000
    torch.manual_seed(41)
    model = Demucs(sources=["clean"], audio_channels=2).cuda().train()
    g = torch.Generator().manual_seed(42)
    ws = model.workspace(B, T)
    nb = ws._norm_bwd
    rec = []

    def norm_bwd(key, params, dz, dy):
        n, j = ws.st.gch[key], ws.norm_idx[key]
        if not n["G"]:
            return nb(key, params, dz, dy)
        pre = ws.sums[j].clone()
        r = nb(key, params, dz, dy)
        post = ws.sums[j].clone()
        rec.append((key, j, pre, post, dz, dy))
        return r

    ws._norm_bwd = norm_bwd
    mix = (0.3 * torch.randn(B, 2, T, generator=g) + 0.05).cuda()
    for p in range(passes):
        rec.clear()
        est = model(mix)
        G = (torch.randn(est.shape, generator=g) / est.numel() ** 0.5).cuda()
        est.backward(G)
        torch.cuda.synchronize()
        bad_pre = [(k, pre.flatten()[:4].tolist()) for k, j, pre, post, dz, dy in rec if float(pre.abs().max()) != 0.0]
        # repeat every reduce now, alone, on the same operands
        ws._norm_bwd = nb
        diffs = []
        for key, j, pre, post, dz, dy in rec:
            ws.sums[j].zero_()
            gp = ws.gpack.clone()
            nb(key, model.flat_params, dz, dy)
            torch.cuda.synchronize()
            ws.gpack.copy_(gp)
            if not torch.equal(ws.sums[j], post):
                d = (ws.sums[j] - post).flatten()
                i = int(d.abs().argmax())
                diffs.append((key, i, float(post.flatten()[i]), float(ws.sums[j].flatten()[i])))
        ws._norm_bwd = norm_bwd
        print(f"pass {p}: entries that were not zero before their reduce: {bad_pre}; in-step result differs from the repeat alone: {diffs}", flush=True)


main()
