#!/usr/bin/env python3
"""Generate tests/golden/dcunet_tiny.npz by IMPORTING the real reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_dcunet.py
Tiny complex DCUNet-10 (model_complexity 8 -> 5 complex channels, src/model/dcunet.py:64-65) on a [2, 1, 257, 33, 2]
spectrum (T = 33 = 1 mod 32, the depth-10 constraint): state_dict, input / target, the output of every encoder / decoder
block, the enhanced spectrum in train and eval mode, an mse loss, every parameter gradient and the updated running stats.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def main():
    from src.model.dcunet import DCUnet
    torch.manual_seed(10)
    model = DCUnet(audio_channels=1, data_type=True, model_complexity=8, model_depth=10, masking_mode="E")
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():   # non-trivial affine terms / running stats
        for name, prm in model.named_parameters():
            if ".bn." in name:
                prm.copy_((1.0 if name.endswith("weight") else 0.0) + 0.2 * torch.randn(prm.shape, generator=g))
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
            if name.endswith("running_var"):
                buf.copy_(1.0 + 0.3 * torch.rand(buf.shape, generator=g))
    x = 0.5 * torch.randn(2, 1, 257, 33, 2, generator=g)
    tgt = 0.5 * torch.randn(2, 1, 257, 33, 2, generator=g)
    out = {}
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        if not k.startswith(("encoders.", "decoders.")):   # the reference registers every block twice: same tensors
            out["sd." + k] = v.numpy()
    out["n_state_dict_keys"] = np.array(len(sd0))
    model.eval()
    with torch.no_grad():
        out["eval_out"] = model(x).numpy()
    model.train()
    taps = {}
    hooks = []
    for i in range(5):
        hooks.append(getattr(model, f"encoder{i}").register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"encoder{i}", o.detach().clone())))
        hooks.append(getattr(model, f"decoder{i}").register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"decoder{i}", o.detach().clone())))
    est = model(x)
    loss = torch.nn.functional.mse_loss(est, tgt)
    loss.backward()
    for h in hooks:
        h.remove()
    for k, v in taps.items():
        out["tap." + k] = v.numpy()
    out["x"], out["target"], out["train_out"], out["loss"] = x.numpy(), tgt.numpy(), est.detach().numpy(), np.array(loss.item())
    seen = set()
    for name, prm in model.named_parameters():           # named_parameters de-duplicates the encoders./decoders. aliases
        out["grad." + name] = prm.grad.numpy()
        seen.add(name)
    for k, v in model.state_dict().items():
        if k.endswith(("running_mean", "running_var")) and not k.startswith(("encoders.", "decoders.")):
            out["stat." + k] = v.numpy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "dcunet_tiny.npz"), **out)
    print("wrote dcunet_tiny.npz:", len(out), "arrays, loss", loss.item(), "grads", len(seen))


def main20():
    """tests/golden/dcunet20_tiny.npz: the depth-20 tables (src/model/dcunet.py:215-305) on the one spectrum shape the reference
    network accepts at that depth, [1, 1, 257, 257, 2]: state_dict, input / target, train and eval output, mse loss, every
    parameter gradient, updated running statistics."""
    from src.model.dcunet import DCUnet
    torch.manual_seed(20)
    model = DCUnet(audio_channels=1, data_type=True, model_complexity=8, model_depth=20, masking_mode="E")
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if ".bn." in name:
                prm.copy_((1.0 if name.endswith("weight") else 0.0) + 0.2 * torch.randn(prm.shape, generator=g))
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
            if name.endswith("running_var"):
                buf.copy_(1.0 + 0.3 * torch.rand(buf.shape, generator=g))
    x = (0.5 * torch.randn(1, 1, 257, 257, 2, generator=g)).half().float()      # fp16-representable values: stored as fp16, exact
    tgt = (0.5 * torch.randn(1, 1, 257, 257, 2, generator=g)).half().float()
    out = {}
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        if not k.startswith(("encoders.", "decoders.")):
            out["sd." + k] = v.numpy()
    out["n_state_dict_keys"] = np.array(len(sd0))
    model.eval()
    with torch.no_grad():
        out["eval_out"] = model(x).numpy()
    model.train()
    est = model(x)
    loss = torch.nn.functional.mse_loss(est, tgt)
    loss.backward()
    out["x"], out["target"] = x.numpy().astype(np.float16), tgt.numpy().astype(np.float16)
    out["train_out"], out["loss"] = est.detach().numpy(), np.array(loss.item())
    n = 0
    for name, prm in model.named_parameters():
        out["grad." + name] = prm.grad.numpy()
        n += 1
    for k, v in model.state_dict().items():
        if k.endswith(("running_mean", "running_var")) and not k.startswith(("encoders.", "decoders.")):
            out["stat." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "dcunet20_tiny.npz"), **out)
    print("wrote dcunet20_tiny.npz:", len(out), "arrays, loss", loss.item(), "grads", n)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "20":
        main20()
    else:
        main()
