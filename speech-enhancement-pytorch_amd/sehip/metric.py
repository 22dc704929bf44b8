"""On-device validation metric: SI_SDR of src/metric.py:92-123 without the round trip through numpy.

Only SI-SDR is rebuilt: PESQ / STOI / HASPI of src/metric.py wrap third-party CPU libraries and are out of scope."""
import torch

from ._lib import call, ptr, require_gpu, stream


def SI_SDR(reference, estimation, sr=16000):
    """reference / estimation: device tensors [..., T] of the same shape -> 0-dim device tensor (float32).

    Same arithmetic as the reference (projection on the reference signal, mean of the per-row energy ratios, THEN the
    logarithm; eps = float32 machine epsilon); `sr` is accepted and unused, as in the reference."""
    require_gpu(reference, "SI_SDR")
    require_gpu(estimation, "SI_SDR")
    if reference.shape != estimation.shape:
        raise ValueError(f"SI_SDR: shapes differ: {tuple(reference.shape)} vs {tuple(estimation.shape)}")
    n = reference.shape[-1]
    r = reference.reshape(-1, n).contiguous().float()
    e = estimation.reshape(-1, n).contiguous().float()
    ratios = torch.empty(r.shape[0], device=r.device, dtype=torch.float32)
    out = torch.empty((), device=r.device, dtype=torch.float32)
    call("sehip_sisdr_metric", ptr(r), ptr(e), r.shape[0], n, ptr(ratios), ptr(out), stream())
    return out
