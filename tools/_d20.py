import sys, os
sys.path.insert(0, "speech-enhancement-pytorch_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from oracle import dcunet_oracle as D
from sehip.model import DCUnet
torch.manual_seed(3)
for mc, B, T in ((8, 2, 33), (45, 1, 65)):
    model = DCUnet(data_type=True, model_complexity=mc, model_depth=20)
    p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("encoders.", "decoders."))}
    model = model.cuda().train()
    g = torch.Generator().manual_seed(4)
    x = 0.5 * torch.randn(B, 1, 257, T, 2, generator=g)
    names = sorted(k for k in p if D.is_trainable(k))
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    ref = D.dcunet_forward(work, x, model_complexity=mc, model_depth=20, training=True)
    G = torch.randn(ref.shape, generator=g) / ref.numel() ** 0.5
    grads = torch.autograd.grad((ref * G).sum(), [leaves[k] for k in names])
    est = model(x.cuda())
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    print("mc", mc, "out rel", rel(est.detach().cpu(), ref.detach()))
    est.backward(G.cuda()); torch.cuda.synchronize()
    got = {k: v.grad.detach().cpu() for k, v in model.named_parameters() if not k.startswith(("encoders.", "decoders."))}
    num = sum(float(((got[k].double() - gr.double()) ** 2).sum()) for k, gr in zip(names, grads))
    den = sum(float((gr.double() ** 2).sum()) for gr in grads)
    print("   global grad rel (plain oracle)", (num / den) ** 0.5)
