#!/usr/bin/env python3
"""tests/golden/evaluate_cases.npz from the IMPORTED reference's evaluate() (src/evaluate.py:10-98; build container only).

evaluate() = per-utterance normalisation, segmentation with stride win_length, the model on the stacked segments (in two
halves), [STFT-domain models: stft_custom before / istft_custom after], stitching (first segment whole, then the last
`stride` samples of every further one), crop, de-normalisation.  The model is a fixed toy module (y = 0.5 x + 0.1 x^2, shape
preserving; for the two-speaker case it stacks (y, -y)) so that the fixture pins the segment / stitch / normalise logic and
the STFT round trip, not a network.  Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_evaluate.py"""
import os, sys, types
import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "evaluate_cases.npz")
from src.evaluate import evaluate  # noqa: E402


class Toy(torch.nn.Module):
    def __init__(self, nspk=0):
        super().__init__()
        self.nspk = nspk

    def forward(self, x):
        y = 0.5 * x + 0.1 * x * x
        return torch.stack([y, -y], dim=1) if self.nspk == 2 else y


def cfg(name, norm, segment, win=512, sources=None):
    m = types.SimpleNamespace(name=name, win_length=win, n_fft=512, hop_length=128, center=True, segment=segment, sources=sources)
    d = types.SimpleNamespace(norm=norm, sample_rate=16000)
    return types.SimpleNamespace(model=m, dset=d)


CASES = {   # name: (model name, norm, segment seconds, samples, batch, channels, speakers)
    "dccrn_z": ("dccrn", "z-score", 0.25, 4000 + 3 * 512 + 77, 2, 1, 0),       # padded tail
    "dccrn_none_exact": ("dccrn", None, 0.25, 4000 + 2 * 512, 1, 1, 0),          # no padding needed
    "dcunet_z": ("dcunet", "z-score", 0.256, 4096 + 2 * 512 + 5, 2, 1, 0),       # STFT-domain: stft_custom -> model -> istft_custom
    # batch 1 (the reference's test loader, src/solver.py:552): its de-normalisation broadcasts mean/std [B, C, 1] against
    # [B, S, C, N] from the right, i.e. pairs the BATCH index of the statistics with the SPEAKER index -- only meaningful at B = 1
    "convtasnet_2spk": ("conv-tasnet", "z-score", 0.25, 4000 + 512 + 1, 1, 1, 2),
}

out = {}
g = torch.Generator().manual_seed(0)
for key, (name, norm, seg, n, b, c, spk) in CASES.items():
    x = 0.3 * torch.randn(b, c, n, generator=g) + 0.05
    y = evaluate(x, Toy(spk), "cpu", cfg(name, norm, seg, sources=["None"] * spk if spk else None))
    out[key + ".x"], out[key + ".y"] = x.numpy(), y.numpy()
    print(key, tuple(x.shape), "->", tuple(y.shape))
np.savez_compressed(OUT, **out)
print(os.path.getsize(OUT), "bytes")
