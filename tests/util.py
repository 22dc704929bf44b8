"""Shared helpers for the tests (golden loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    """Entries of ``d`` under ``prefix/`` as torch tensors."""
    return {k[len(prefix) + 1:]: torch.from_numpy(np.asarray(v)) for k, v in d.items()
            if k.startswith(prefix + "/")}


def json_entry(d, key):
    return json.loads(bytes(d[key]).decode())


def rel_err(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def golden_grads(g):
    """Gradients stored by oracle/gen_golden.py:_pack_grads -> ({name: fp32 tensor} for the fully stored ones, {name: norm})."""
    names = json_entry(g, "grad_names_json")
    norms = dict(zip(names, [float(v) for v in g["grad_norms"]]))
    full = {k[len("grad16/"):]: torch.from_numpy(g[k].astype(np.float32)) * float(g["gscale/" + k[len("grad16/"):]])
            for k in g if k.startswith("grad16/")}
    return full, norms


def make_batch(seed, b, n):
    """The synthetic (noisy [B,1,N], clean [B,1,1,N]) pair used by every DCCRN fixture (SURVEY section 8d recipe)."""
    gen = torch.Generator().manual_seed(seed)
    clean = 0.1 * torch.randn(b, 1, 1, n, generator=gen)
    noisy = clean[:, 0] + 0.05 * torch.randn(b, 1, n, generator=gen)
    return noisy, clean
