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
