#!/usr/bin/env python3
"""tests/golden/demucs_tiny.npz from the IMPORTED reference (build container only).

Two small Demucs models (src/model/demucs.py:272-501) with ``resample=False``:
  a: channels=4, depth=5, GroupNorm(4) and DConv with BLSTM + LocalState from layer 3 (the default structure one level
     shallower, to keep the fixture small), GLU rewrite, normalize, stereo, 2 sources, [2, 2, 9000] mixture;
  b: channels=8, depth=3, dconv_lstm=1, dconv_attn=2, norm_starts=1, mono, 1 source, [1, 1, 18000] mixture: layer 1 has
     T = 1123 > max_steps = 200, i.e. the BLSTM runs on overlapping chunks (:91-117).
Per case: state_dict, input / target, every encoder / decoder output, the separated sources, the reference's SI-SNR loss
(src/loss.py:14-29) and every parameter gradient.

src/model/demucs.py:12 imports julius (absent here) at module level.  It is only CALLED when resample=True (:469-470,
:485-486), so the import is satisfied with an empty module whose resample_frac raises: no arithmetic is stood in for, and the
resampler stays "parity unpinned" (oracle/demucs_oracle.py).
Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_demucs.py"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demucs_tiny.npz")


def _absent(*a, **k):
    raise RuntimeError("julius is not installed: the golden vectors are generated with resample=False")


sys.modules["julius"] = types.ModuleType("julius")
sys.modules["julius"].resample_frac = _absent

from src.model.demucs import Demucs  # noqa: E402
from src.loss import loss_sisdr  # noqa: E402

CASES = {
    "a": (dict(sources=["s0", "s1"], audio_channels=2, channels=4, depth=5, norm_starts=3, dconv_lstm=3, dconv_attn=3, resample=False), (2, 2, 9000), 11),
    "b": (dict(sources=["s0"], audio_channels=1, channels=8, depth=3, dconv_lstm=1, dconv_attn=2, norm_starts=1, resample=False),
          (1, 1, 18000), 12),
}
out = {}
for tag, (kw, shape, seed) in CASES.items():
    torch.manual_seed(seed)
    model = Demucs(**kw)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():   # non-trivial norm affine terms and layer scales (1e-4 at init would hide the DConv branches)
        for name, prm in model.named_parameters():
            if name.endswith(".scale"):
                prm.copy_(0.3 + 0.1 * torch.randn(prm.shape, generator=g))
            elif prm.dim() == 1 and "lstm" not in name and name.endswith("weight"):
                prm.copy_(1 + 0.2 * torch.randn(prm.shape, generator=g))
            elif prm.dim() == 1 and "lstm" not in name and "query_decay" not in name:
                prm.copy_(0.1 * torch.randn(prm.shape, generator=g))
    mix = 0.3 * torch.randn(*shape, generator=g) + 0.05
    with torch.no_grad():
        e0 = model(mix)
    tgt = e0 + 0.3 * e0.std() * torch.randn(e0.shape, generator=g)
    taps = {}
    hooks = [m.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"enc{i}", o.detach().clone())) for i, m in enumerate(model.encoder)]
    hooks += [m.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"dec{i}", o.detach().clone())) for i, m in enumerate(model.decoder)]
    est = model(mix)
    loss = loss_sisdr(est, tgt)
    loss.backward()
    pre = tag + "."
    for k, v in model.state_dict().items():
        out[pre + "sd." + k] = v.detach().clone().numpy()
    for k, v in taps.items():
        out[pre + "tap." + k] = v.numpy()
    out[pre + "mix"], out[pre + "target"], out[pre + "est"] = mix.numpy(), tgt.numpy(), est.detach().numpy()
    out[pre + "loss"] = np.float32(loss.item())
    out[pre + "names"] = np.array([k for k, _ in model.named_parameters()])
    for k, prm in model.named_parameters():
        out[pre + "grad." + k] = prm.grad.numpy()
    print(tag, "params", sum(p.numel() for p in model.parameters()), "est", tuple(est.shape), "loss", loss.item(),
          {k: tuple(v.shape) for k, v in taps.items()})
np.savez_compressed(OUT, **out)
print("demucs golden:", len(out), "entries,", os.path.getsize(OUT), "bytes")
