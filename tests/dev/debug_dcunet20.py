"""Development aid: DCUnet (DEPTH=10|20, MC=complexity, BB=batch, TT=frames) against the oracle: output error, global gradient error, the worst
per-tensor gradient errors and the norms of the chain's gradient tensors per encoder level.  (Round 3: this is how the depth-20 failure was
found -- sehip_rbn_bwd_finalize's [2 cs][4] coefficient block was allocated for cs <= 64 and the 128-channel bottleneck of depth 20 wrote
2 KB past it, into the chunk tables of the first three encoder products.)"""
import sys, os
sys.path.insert(0, "speech-enhancement-pytorch_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from oracle import dcunet_oracle as D
from sehip.model import DCUnet
torch.manual_seed(3)
DEPTH = int(os.environ.get('DEPTH', '20'))
for mc, B, T in ((int(os.environ.get('MC', '8')), int(os.environ.get('BB', '1')), int(os.environ.get('TT', '257'))),):
    model = DCUnet(data_type=True, model_complexity=mc, model_depth=DEPTH)
    p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("encoders.", "decoders."))}
    model = model.cuda().train()
    g = torch.Generator().manual_seed(4)
    x = 0.5 * torch.randn(B, 1, 257, T, 2, generator=g)
    names = sorted(k for k in p if D.is_trainable(k))
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    ref = D.dcunet_forward(work, x, model_complexity=mc, model_depth=DEPTH, training=True)
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
    # where the encoder gradients stop: norms of the chain's gradient tensors and the kernels the weight gradients took
    import ctypes as C
    from sehip._lib import lib
    ws = model.workspace(B, 257, T)
    L = lib(); L.sehip_last_kernel.restype = C.c_char_p
    for i in range(DEPTH // 2):
        nm = lambda k: float(ws.bufs[k].t.float().norm()) if k in ws.bufs else None
        print(f"   enc{i}: |ze| {nm(f'ze{i}')} |dze| {nm(f'dze{i}')} |dye| {nm(f'dye{i}')}  dg specs {[k for k in ws.pl.specs if k.startswith(f'enc{i}.dg')]}")
        d = ws.desc[f"enc{i}.fwd.wg"]
        print(f"        wg: M {d.M} N {d.N} Npad {d.Npad} K {d.K} TT {d.TT} J {d.J} tmul {d.tmul} fmul {d.fmul} cv_nf {d.cv_nf} cv2 {d.cv2_nkt} {d.cv2_nf} dW {d.dW is not None}")
