#!/usr/bin/env python3
"""tests/golden/pit_cases.npz from the IMPORTED reference's UtterenceBaasedPermutationInvariantTraining (src/loss.py:58-100)
around loss_sisdr (src/loss.py:14-29) and torch l1 / mse (src/distrib.py:263-268); build container only.
Cases: 2 and 3 speakers, targets = a permutation of (estimates + noise) so that the winning permutation is not the identity,
one case with a channel axis; stored: inputs, loss, the chosen (ienhance, itarget) pairs, d loss / d enhance.
Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_pit.py"""
import os, sys
import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pit_cases.npz")
from src.loss import UtterenceBaasedPermutationInvariantTraining as ref_pit, loss_sisdr  # noqa: E402

LOSSES = {"sisdr": loss_sisdr, "l1": torch.nn.functional.l1_loss, "mse": torch.nn.functional.mse_loss}
CASES = {   # name: (B, S, C, n, order of the targets relative to the estimates, loss)
    "s2_swap": (3, 2, 1, 1000, (1, 0), "sisdr"),
    "s2_id": (2, 2, 1, 777, (0, 1), "sisdr"),
    "s3_rot": (2, 3, 1, 640, (2, 0, 1), "sisdr"),
    "s2_c2": (2, 2, 2, 512, (1, 0), "sisdr"),
    "s2_l1": (2, 2, 1, 300, (1, 0), "l1"),
    "s3_mse": (2, 3, 1, 300, (1, 2, 0), "mse"),
}
out = {}
g = torch.Generator().manual_seed(0)
for key, (b, s, c, n, order, lname) in CASES.items():
    est = (0.3 * torch.randn(b, s, c, n, generator=g)).requires_grad_(True)
    tgt = torch.stack([est.detach()[:, o] for o in order], dim=1) + 0.1 * torch.randn(b, s, c, n, generator=g)
    loss, comb = ref_pit(enhance=est, target=tgt, loss_function=LOSSES[lname], return_comb=True)
    loss.sum().backward()
    out[key + ".est"], out[key + ".tgt"] = est.detach().numpy(), tgt.numpy()
    out[key + ".loss"] = loss.detach().numpy().reshape(1)
    out[key + ".comb"] = np.asarray(comb, dtype=np.int64)
    out[key + ".grad"] = est.grad.numpy()
    out[key + ".lname"] = np.asarray(lname)
    print(key, float(loss), comb)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT))
