"""Does the non-finite-gradient failure of the fused BatchNorm backward finalize (SEHIP_FUSE_BWD_FINALIZE=1) follow
the same rule as the two-queue disturbance of the Demucs step -- gone when the streaming dense-row weight-gradient kernel takes a CU's
whole LDS (SEHIP_DTW_LDS_ALL=1)?  Runs child processes: (fused, default), (fused, LDS_ALL), (fused, no side stream), each 12 headline
train steps at B = 32, and reports whether the loss / parameters stay finite.      python tools/dev/nan_lds_all.py"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "speech-enhancement-pytorch_amd"))
from sehip.model import DCCRN
from sehip.optim import FlatOptimizer
from sehip.loss import loss_sisdr
torch.manual_seed(0)
model = DCCRN(length=32000).cuda().train()
opt = FlatOptimizer(model, lr=3e-4)
g = torch.Generator().manual_seed(1)
clean = 0.1 * torch.randn(32, 1, 32000, generator=g).cuda()
noisy = clean + 0.05 * torch.randn(32, 1, 32000, generator=g).cuda()
bad = None
for s in range(12):
    opt.zero_grad()
    loss = loss_sisdr(model(noisy), clean)
    loss.backward()
    opt.clip_grad_norm_(5.0)
    opt.step()
    torch.cuda.synchronize()
    if not (torch.isfinite(loss).item() and torch.isfinite(model.flat_params).all().item()):
        bad = s
        break
print("RESULT", "non-finite at step %d" % bad if bad is not None else "finite (loss %.3f)" % float(loss))
'''


def run(env_extra):
    env = dict(os.environ, SEHIP_FUSE_BWD_FINALIZE="1", **env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    return lines[-1] if lines else "no result: " + out.stderr[-300:]


for name, extra in (("fused finalize, two queues", {}), ("fused finalize, two queues, dense-row kernel with the whole LDS", {"SEHIP_DTW_LDS_ALL": "1"}),
                    ("fused finalize, one queue", {"SEHIP_NO_SIDE_STREAM": "1"})):
    for rep in range(3):
        print(f"{name} [{rep}]: {run(extra)}", flush=True)
