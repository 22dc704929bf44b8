#!/usr/bin/env python3
"""tests/golden/psa_loss.npz from the IMPORTED reference (build container only; nothing of it is copied): the phase-sensitive spectral
approximation loss of src/loss.py:32-56 on seeded [3, 1, 17, 9, 2] spectra (one case with the speaker axis the Solver adds for
separation models, src/solver.py:477-478): the loss value and its gradient w.r.t. the enhanced spectrum.
Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_psa.py"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (puts /root/reference on sys.path, stubs the absent third-party imports)

G.install_stubs()
from src.loss import loss_phase_sensitive_spectral_approximation as psa  # noqa: E402

out = {}
for name, shape, seed in (("a", (3, 1, 17, 9, 2), 51), ("b", (2, 2, 1, 17, 9, 2), 52)):
    g = torch.Generator().manual_seed(seed)
    enh = torch.randn(shape, generator=g).requires_grad_(True)
    tgt = torch.randn(shape, generator=g)
    mix = torch.randn(shape, generator=g)
    with torch.no_grad():                     # a few awkward elements: tiny real parts (large ratios), a negative-real target
        tgt[..., 0, 0, 0] = 1e-6
        mix[..., 1, 1, 0] = -1e-6
    loss = psa(enh, tgt, mix)
    loss.backward()
    out.update({f"{name}/enh": enh.detach().numpy(), f"{name}/tgt": tgt.numpy(), f"{name}/mix": mix.numpy(),
                f"{name}/loss": np.float32(loss.item()), f"{name}/denh": enh.grad.numpy()})
    print(name, shape, "loss", loss.item())
np.savez_compressed(os.path.join(G.OUT, "psa_loss.npz"), **out)
