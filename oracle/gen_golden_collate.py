#!/usr/bin/env python3
"""tests/golden/collate_cases.npz from the IMPORTED reference's collate_fn_pad (src/distrib.py:38-98) and
sample_fixed_length_data_aligned (src/utils.py:63-87); build container only.  src/distrib.py imports packages that are absent here
(omegaconf, soundfile, librosa, julius, metric libraries): they are satisfied with EMPTY modules exactly as oracle/gen_golden.py does
for the Solver fixture -- none of them is called by the two functions.  WavDataset.__getitem__ itself needs soundfile, so its three
z-score lines (src/dataset.py:147-152) are applied here verbatim to the synthetic utterances before the real crop / collate.
Cases: ragged lengths (shorter than the sample, shorter than a segment, not a multiple of it), mono / stereo, one / two sources,
drop_last True / False, with / without crop.  Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_collate.py"""
import os, sys, types
import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "collate_cases.npz")
from gen_golden import install_stubs  # noqa: E402
install_stubs()
from src.distrib import collate_fn_pad  # noqa: E402
from src.utils import sample_fixed_length_data_aligned  # noqa: E402

CASES = {   # name: (C, S, lengths, sample_length, segment seconds, sample_rate, drop_last, z-score?)
    "crop_zs": (1, 1, [9000, 2500, 4000, 4001], 4000, 0.125, 16000, True, True),       # seg 2000: two segments per utterance
    "crop_stereo_2spk": (2, 2, [5000, 7000], 3000, 0.125, 16000, True, True),            # 3000 = 1.5 segments: drop_last cuts to 1
    "crop_pad_last": (1, 1, [5000, 7000, 900], 3000, 0.125, 16000, False, True),          # ... drop_last False pads to 2
    "nocrop_ragged": (1, 1, [1500, 4100, 6000], 0, 0.125, 16000, True, False),            # shorter than a segment / ragged
    "nocrop_pad": (2, 1, [1500, 4100], 0, 0.125, 16000, False, True),
}
out = {}
g = torch.Generator().manual_seed(0)
for key, (C, S, lengths, sl, segs, sr, drop_last, zs) in CASES.items():
    cfg = types.SimpleNamespace(segment=segs, sample_rate=sr)
    np.random.seed(7)
    items, starts = [], []
    for i, n in enumerate(lengths):
        mixture = 0.2 * torch.randn(C, n, generator=g) + 0.03
        sources = 0.2 * torch.randn(S, C, n, generator=g) - 0.02
        out[f"{key}.mix{i}"], out[f"{key}.src{i}"] = mixture.numpy().copy(), sources.numpy().copy()
        if zs:      # src/dataset.py:147-152
            eps = 1e-6
            mixture = (mixture - torch.mean(mixture, axis=-1, keepdims=True)) / (torch.std(mixture, axis=-1, keepdims=True) + eps)
            sources = (sources - torch.mean(sources, axis=-1, keepdims=True)) / (torch.std(sources, axis=-1, keepdims=True) + eps)
        if sl:
            state = np.random.get_state()
            st = int(np.random.randint(max(n, sl) - sl + 1))       # what the reference draws next ...
            np.random.set_state(state)
            mixture, sources = sample_fixed_length_data_aligned([mixture, sources], sl)   # ... and here it draws it
            starts.append(st)
        else:
            starts.append(0)
        items.append((mixture, sources, {}, {}, f"utt{i}"))
    bm, bs, _, _, names, index_batch = collate_fn_pad(cfg, drop_last=drop_last)(items)
    out[key + ".mixture"], out[key + ".sources"] = bm.contiguous().numpy(), bs.contiguous().numpy()
    out[key + ".index_batch"], out[key + ".starts"] = np.asarray(index_batch), np.asarray(starts)
    out[key + ".cfg"] = np.asarray([C, S, sl, int(segs * sr), int(drop_last), int(zs), len(lengths)])
    print(key, tuple(bm.shape), tuple(bs.shape), index_batch, starts)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT))
