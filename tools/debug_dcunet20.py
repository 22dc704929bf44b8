"""Development aid: DCUnet depth 20 (SEHIP_DCUNET20=1) against the oracle at [1, 1, 257, 257, 2]: output error, global gradient
error, per-tensor gradient errors.  State at round 3: forward 1.3e-2 (complexity 8), encoder 0-2 weight gradients wrong, complexity 45 faults."""
import sys, os
sys.path.insert(0, "speech-enhancement-pytorch_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from oracle import dcunet_oracle as D
from sehip.model import DCUnet
torch.manual_seed(3)
for mc, B, T in ((8, 1, 257),):
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
    import os; os.environ['X']='1'
    est = model(x.cuda()); torch.cuda.synchronize(); print('forward done', flush=True)
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    print("mc", mc, "out rel", rel(est.detach().cpu(), ref.detach()))
    est.backward(G.cuda()); torch.cuda.synchronize()
    got = {k: v.grad.detach().cpu() for k, v in model.named_parameters() if not k.startswith(("encoders.", "decoders."))}
    num = sum(float(((got[k].double() - gr.double()) ** 2).sum()) for k, gr in zip(names, grads))
    den = sum(float((gr.double() ** 2).sum()) for gr in grads)
    print("   global grad rel (plain oracle)", (num / den) ** 0.5)
    rows = sorted(((float((got[k].double() - gr.double()).norm() / (gr.double().norm() + 1e-30)), float(gr.norm()), k) for k, gr in zip(names, grads)), reverse=True)
    for r in rows[:60]:
        if r[1] > 1e-7: print("   %.3e |g| %.2e %s" % r)
