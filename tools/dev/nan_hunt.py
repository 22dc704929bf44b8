#!/usr/bin/env python3
"""Debug: N train steps of the headline workload, after each the parameter tensors / gpack segments that hold non-finite values."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
solver, model, mixture, sources = bench.build_for_profile()
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
model.train()
with torch.no_grad():
    model(mixture)
L = model.static.layout
bad_any = False
for it in range(nsteps):
    loss, metric = solver.train_step(mixture, sources)
    torch.cuda.synchronize()
    g = model.flat_grads
    ws = model.workspace(mixture.shape[0], mixture.shape[-1])
    gp = ws.gpack
    nf = ~torch.isfinite(g)
    if bool(nf.any()) or not bool(torch.isfinite(gp).all()):
        bad_any = True
        names = []
        for name in L.param_names:
            off, shape = L.param_off[name]
            n = 1
            for s in shape: n *= s
            if bool(nf[off:off + n].any()):
                names.append(name)
        print(f"step {it}: loss {float(loss):.4f} NON-FINITE grads in {len(names)} tensors: {names[:12]}")
        bg = (~torch.isfinite(gp)).nonzero().flatten()
        print("   gpack non-finite count", int(bg.numel()), "first idx", bg[:8].tolist())
        for pre, cr in model.static.bn:
            offs = model.static.bn_g_off[pre]
            for k, o in offs.items():
                seg = gp[o:o + (1 if k == "slope" else cr)]
                if not bool(torch.isfinite(seg).all()):
                    print("   BN grad non-finite:", pre, k)
        for pre, rep in ws.bn_brep.items():
            if not bool(torch.isfinite(rep).all()):
                print("   rep rows non-finite:", pre)
    else:
        print(f"step {it}: loss {float(loss):.4f} metric {float(metric[0]):.3f} ok")
print("RESULT", "BAD" if bad_any else "clean")
# ---- after the run: which BatchNorm backward inputs / outputs are non-finite in the LAST step, and do the kernels reproduce it alone
import ctypes as C
from sehip._lib import call, ptr, stream
st = model.static
b = ws.bufs
params = model.flat_params
Lh = st.layout
cfg = st.cfg
for i in range(5, -1, -1):
    pre = f"encoder.{i}."
    cr = cfg.kernel_num[i + 1] // 2
    dz = b["dz5l"] if i == 5 else b[f"dz{i}"]
    y, dy = b[f"y{i}"], b[f"dye{i}"]
    rows = y.t.numel() // (2 * cr)
    coef = ws.bn_coef[pre]
    fin = lambda t: bool(torch.isfinite(t.float()).all())
    k = coef.view(cr, 16)
    delta = k[:, 11] * k[:, 13] - k[:, 12] ** 2
    pp = lambda kk: params.data_ptr() + 4 * Lh.param_off[pre + kk][0]
    print(f"{pre} Cr={cr} rows={rows}: dz finite {fin(dz.t)} y finite {fin(y.t)} dy finite {fin(dy.t)} coef finite {fin(coef)} min delta {float(delta.min()):.3e} "
          f"rep finite {fin(ws.bn_brep[pre])} |rep0| {float(ws.bn_brep[pre][0].abs().sum()):.3e} |rep1| {float(ws.bn_brep[pre][1].abs().sum()):.3e} turn {ws._brep_turn[pre]}")
    # the fused launch again on the same buffers, into scratch outputs
    rep = torch.zeros(2, 8 * (6 * cr + 1), device=params.device)
    gs = [torch.zeros(cr, device=params.device) for _ in range(5)] + [torch.zeros(1, device=params.device)]
    out = torch.empty_like(dy.t)
    call("sehip_cbn_bwd_fused", dz.ptr, None, y.ptr, ptr(coef), pp("1.Wrr"), pp("1.Wri"), pp("1.Wii"), pp("2.weight"), rows, cr, y.F, y.Tst, 0,
         ptr(rep[0]), ptr(rep[1]), 8, *[ptr(t) for t in gs], ptr(out), stream())
    torch.cuda.synchronize()
    print(f"     fused again: dy finite {fin(out)} grads finite {all(fin(t) for t in gs)}  (params finite {fin(params)})")
# ---- pattern of the damage in encoder.5's dy against the separate launches on the same inputs
i = 5
pre = f"encoder.{i}."; cr = cfg.kernel_num[i + 1] // 2
dz, y, dy = b["dz5l"], b[f"y{i}"], b[f"dye{i}"]
rows = y.t.numel() // (2 * cr)
coef = ws.bn_coef[pre]
pp = lambda kk: params.data_ptr() + 4 * Lh.param_off[pre + kk][0]
acc = torch.zeros(512 * (6 * cr + 1), device=params.device)
bco = torch.zeros(cr, 16, device=params.device)
gs = [torch.zeros(cr, device=params.device) for _ in range(5)] + [torch.zeros(1, device=params.device)]
ref = torch.empty_like(dy.t)
if bool(torch.isfinite(params).all()) or True:
    # (the parameters may be NaN after a bad optimizer step: use finite stand-ins for Wrr / Wri / Wii / slope, the comparison is of structure)
    W = [torch.full((cr,), v, device=params.device) for v in (0.7071, 0.0, 0.7071)]
    sl = torch.full((1,), 0.25, device=params.device)
    call("sehip_cbn_bwd_reduce", dz.ptr, None, y.ptr, ptr(coef), ptr(sl), rows, cr, y.F, y.Tst, 0, ptr(acc), stream())
    call("sehip_cbn_bwd_finalize", ptr(acc), ptr(coef), ptr(W[0]), ptr(W[1]), ptr(W[2]), rows, cr, *[ptr(t) for t in gs], ptr(bco), stream())
    call("sehip_cbn_bwd_apply", dz.ptr, None, y.ptr, ptr(coef), ptr(bco), ptr(sl), rows, cr, y.F, y.Tst, 0, ptr(ref), stream())
    torch.cuda.synchronize()
d = dy.t.float().reshape(rows, 2 * cr)
nanmask = ~torch.isfinite(d)
print("enc5 dy: non-finite elements", int(nanmask.sum()), "of", d.numel(), "| rows with any", int(nanmask.any(1).sum()), "| channels with any", int(nanmask.any(0).sum()))
rws = nanmask.any(1).nonzero().flatten()
print("   first bad rows", rws[:12].tolist(), "row mod 8 hist", torch.bincount(rws % 8, minlength=8).tolist())
print("   bad channels", nanmask.any(0).nonzero().flatten().tolist()[:40])
