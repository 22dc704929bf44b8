#!/usr/bin/env python3
"""Generate tests/golden/stft_custom.npz by IMPORTING the real reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_stft.py
Needs /root/reference (read-only).  The fixture holds inputs and the reference's numeric outputs of
src/evaluate.py stft_custom / istft_custom for the two shipped STFT configurations (src/conf/config.yaml:38-41:
512/128/512, test/conf/config.yaml:38-41: 512/256/512), a shorter window, a 4-D (speaker) input, an odd length, and (round 6) four
configurations with n_fft other than 512: 256, 1024 with an 800-sample window, 400 (not a power of two), 255 (odd).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

CASES = [
    # name, shape, n_fft, hop, win, length for the inverse
    ("cfg_512_128_512", (2, 1, 2000), 512, 128, 512, 2000),
    ("cfg_512_256_512", (2, 1, 2048), 512, 256, 512, 2048),
    ("win_400", (1, 2, 1777), 512, 100, 400, 1777),
    ("spk_4d", (2, 2, 1, 1500), 512, 128, 512, 1500),
    ("longer_length", (1, 1, 1024), 512, 128, 512, 1100),
    # round 6: n_fft other than 512 (src/evaluate.py:101-162 passes config.n_fft to torch.stft as it is); appended, so the cases above
    # keep their random draws
    ("nfft_256", (2, 1, 3000), 256, 64, 256, 3000),
    ("nfft_1024_win_800", (1, 2, 5000), 1024, 256, 800, 5000),
    ("nfft_400", (2, 1, 2500), 400, 100, 400, 2500),
    ("nfft_255_odd", (1, 1, 1500), 255, 60, 255, 1500),
]


def main():
    from src.evaluate import stft_custom, istft_custom
    out = {}
    g = torch.Generator().manual_seed(7)
    for name, shape, n_fft, hop, win, length in CASES:
        cfg = types.SimpleNamespace(n_fft=n_fft, hop_length=hop, win_length=win, center=True)
        x = 0.3 * torch.randn(*shape, generator=g)
        s = stft_custom(x.clone(), cfg)
        # a spectrum that is NOT the transform of a real signal (imaginary DC / Nyquist, random bins) for the inverse
        z = 0.1 * torch.randn(*s.shape, generator=g)
        y = istft_custom(z.clone(), length, cfg)
        y_rt = istft_custom(s.clone(), length, cfg)
        out[name + ".cfg"] = np.array([n_fft, hop, win, length])
        out[name + ".x"] = x.numpy()
        out[name + ".stft"] = s.numpy()
        out[name + ".z"] = z.numpy()
        out[name + ".istft"] = y.numpy()
        out[name + ".roundtrip"] = y_rt.numpy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "stft_custom.npz"), **out)
    print("wrote", os.path.join(OUT, "stft_custom.npz"), {k: v.shape for k, v in out.items() if k.endswith(".stft")})


if __name__ == "__main__":
    main()
