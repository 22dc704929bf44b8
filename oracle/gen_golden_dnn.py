#!/usr/bin/env python3
"""tests/golden/dnn_c0.npz from the IMPORTED reference (build container only): the DNN magnitude-mask model of BASELINE
config C0 (src/model/dnn.py:65-142 with the ExponentialMovingAverage of src/model/ema.py), a small instance.

Stored: the reference's state_dict (the product module must load it by key), an STFT-domain input, the train-mode output
(drop_out 0: deterministic), the mse loss, every parameter gradient, the BatchNorm running statistics afterwards and the
eval-mode output.  Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_dnn.py"""
import io, os, sys, contextlib
import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dnn_c0.npz")
KW = dict(n_fft=64, nfft=64, hidden_layer=48, bias=True, activation="leaky-relu", drop_out=0.0, dnn_method="mask", dnn_ema=True)

from src.model.dnn import DeepNeuralNetwork  # noqa: E402

torch.manual_seed(3)
with contextlib.redirect_stdout(io.StringIO()):
    model = DeepNeuralNetwork(**KW)
g = torch.Generator().manual_seed(4)
with torch.no_grad():
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g)); m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
x = torch.randn(3, 1, 33, 17, 2, generator=g)
tgt = torch.randn(3, 1, 33, 17, 2, generator=g)
out = {"sd." + k: v.clone().numpy() for k, v in model.state_dict().items()}
model.train()
y = model(x)
loss = torch.nn.functional.mse_loss(y, tgt)
loss.backward()
out.update(x=x.numpy(), target=tgt.numpy(), train_out=y.detach().numpy(), loss=np.float32(loss.item()))
for k, p in model.named_parameters():
    if p.grad is not None:
        out["grad." + k] = p.grad.numpy()
for k, v in model.state_dict().items():
    if "running" in k or "num_batches" in k:
        out["stat." + k] = v.numpy()
model.eval()
with torch.no_grad():
    out["eval_out"] = model(x).numpy()
np.savez_compressed(OUT, **out)
print("dnn golden:", len(out), "entries, loss", loss.item(), os.path.getsize(OUT), "bytes")
