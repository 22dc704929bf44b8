"""DCUnet at FULL WIDTH (model_complexity 45: 31 / 62 complex channels stored as 32 / 64 -- the shapes gemm_kernel<192, 128>,
conv_wgrad2_kernel's tap-parity classes and the fused tail were written for), every kernel of the backward pass OP-LOCALLY against
float64 arithmetic on THE SAME OPERANDS the HIP path read (its own stored bf16 activations / gradients, bf16-rounded weights).

Why this file exists.  The whole-chain gradient gate of tests/test_gpu_dcunet.py must allow what bf16 STORAGE
does to a non-smooth network (sign flips of near-zero LeakyReLU pre-activations: 13 % of the gradient norm, reproduced on the CPU
with no HIP code in the loop, tests/test_bf16_storage_oracles.py), so a 3-5 % regression of one backward kernel would pass it.
Storing the activations in fp32 would not remove that: the MFMA operands are still bf16 (2^-9 per element, ~2e-3 per product,
~0.5 % of the branches still flip).  What does remove it is taking the non-smooth chain out of the comparison: every product /
normalisation kernel is checked alone, from the operands it actually read, against exact arithmetic -- the only differences left
are fp32 accumulation order (weight gradients, BatchNorm parameter gradients: <= 1e-4 measured, bound 3e-4) and ONE bf16 rounding of
a stored output (<= 1.2e-3 measured, bound 2e-3).  A kernel that is off by 0.3 % fails these bounds; the chain tests keep checking
that the kernels are composed as the reference composes them."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import dcunet_oracle as D
from util import rel_err
from test_gpu_dcunet import to_ref, eff_bias, bf

pytestmark = pytest.mark.gpu

# Measured on MI355X (round 6), every layer: stored outputs 1.62e-3 ... 1.69e-3 -- exactly the rms of ONE round-to-nearest bf16
# rounding (unit roundoff 2^-8, relative rms 2^-8 / sqrt(3) * E[1 / mantissa] = 1.65e-3), weight / parameter gradients 6e-8 ... 2.1e-7.
OUT_TOL = 1.8e-3    # rms bound of a stored tensor: a systematic 0.1 % error on top of the rounding (1.94e-3) fails
ULP_TOL = 2.0 ** -8 * 1.02   # ... and no element further than one bf16 ulp from the exact value (a tie may round the other way)
ACC_TOL = 2e-5      # fp32 accumulation order of a sum over 1e4 ... 1e6 products of exact operands (100 x the measured figure)
FRAMES, BATCH, CPLX = 65, 2, 45


def d64(t):
    return t.double()


def out_err(got, want):
    """(rms relative error, worst element in units of |want| + 1e-3 rms(want)) of a bf16 tensor against its exact value"""
    got, want = got.double(), want.double()
    floor = 1e-3 * float(want.pow(2).mean().sqrt())
    return rel_err(got, want), float(((got - want).abs() / (want.abs() + floor)).max())


@pytest.fixture(scope="module", params=[45, 90], ids=["complexity45", "complexity90"])
def full(request):
    """(complexity 90 = the "Large" DCUnet of the paper: 63 / 126 complex channels stored as 64 / 128 -- the widest the plan accepts;
    the same gates.)  One forward / backward of the full-width DCUnet-10 through libsehip with the DEFAULT plan (fused tail, the 192-column tile of
    the last decoder's input gradient, the encoders' weight gradients by tap-parity class -- here from 256 rows per utterance so
    that encoder 1 AND 2 take them at 65 frames)."""
    from sehip.model import DCUnet
    from sehip.loss import mse_loss
    old = os.environ.get("SEHIP_DCUNET_ENC_WG_MIN")
    os.environ["SEHIP_DCUNET_ENC_WG_MIN"] = "256"
    try:
        torch.manual_seed(21)
        model = DCUnet(data_type=True, model_complexity=request.param, model_depth=10)
        p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("encoders.", "decoders."))}
        model = model.cuda().train()
        g = torch.Generator().manual_seed(22)
        x = 0.5 * torch.randn(BATCH, 1, 257, FRAMES, 2, generator=g)
        tgt = 0.5 * torch.randn(BATCH, 1, 257, FRAMES, 2, generator=g)
        est = model(x.cuda())
        loss = mse_loss(est, tgt.cuda())
        loss.backward()
        torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop("SEHIP_DCUNET_ENC_WG_MIN", None)
        else:
            os.environ["SEHIP_DCUNET_ENC_WG_MIN"] = old
    ws = model.workspace(BATCH, 257, FRAMES)
    assert ws.fused_tail, "this file checks the default plan (run it without SEHIP_DCUNET_NO_TAIL)"
    kinds = {name: ws.last_kernel.get(name, "") for name in getattr(ws, "last_kernel", {})} if hasattr(ws, "last_kernel") else {}
    grads = {k: v.grad.detach().cpu().clone() for k, v in model.named_parameters()}
    return dict(model=model, ws=ws, est=est.detach().cpu(), grads=grads, p=p, x=x, tgt=tgt, sz=D.dcunet_sizes(request.param, 10, 1), kinds=kinds, cplx=request.param)


def test_the_default_plan_is_the_one_under_test(full):
    """The products this file is about are really in the plan: the 192-column input gradient of the last decoder and the tap-parity
    weight-gradient classes of encoder 1 and 2."""
    if full["cplx"] != 45:
        pytest.skip("the 192-column tile and the class counts below are the complexity-45 shapes")
    pl = full["ws"].pl
    assert any(n.startswith("enc1.wg") for n in pl.enc_wg[1]) and len(pl.enc_wg[1]) == 4, pl.enc_wg[1]
    assert any(n.startswith("enc2.wg") for n in pl.enc_wg[2]) and len(pl.enc_wg[2]) == 4, pl.enc_wg[2]
    assert int(full["ws"].desc["dec4.dg"].Npad) == 192


@pytest.mark.parametrize("i", range(5))
def test_encoder_products_full_width(full, i):
    ws, p, sz = full["ws"], full["p"], full["sz"]
    b = ws.bufs
    cin, cout = sz["enc_ch"][i], sz["enc_ch"][i + 1]
    x = d64(to_ref(b["x0"], 1) if i == 0 else to_ref(b[f"ze{i - 1}"], cin)).requires_grad_(True)
    q = {k: d64(bf(v) if k.endswith("weight") else v).clone().requires_grad_(True) for k, v in p.items() if k.startswith(f"encoder{i}.conv.")}
    y = D.complex_conv2d(x, q, f"encoder{i}.conv.", sz["enc_s"][i], sz["enc_p"][i])
    e_fwd, u_fwd = out_err(to_ref(b[f"ye{i}"], cout), y.detach() - d64(eff_bias(q, f"encoder{i}.conv.conv")))   # stored without the bias
    dy = d64(to_ref(b[f"dye{i}"], cout))
    names = sorted(k for k in q if k.endswith("weight"))
    outs = torch.autograd.grad((y * dy).sum(), [x] + [q[k] for k in names])
    e_dg = u_dg = 0.0
    if i > 0:
        want = outs[0] + (d64(to_ref(b[f"dskip{i - 1}"], cin)) if i - 1 < 4 else 0)   # the input gradient adds the skip connection's (res)
        e_dg, u_dg = out_err(to_ref(b[f"dze{i - 1}"], cin), want)
    e_wg = max(rel_err(d64(full["grads"][k]), gref) for k, gref in zip(names, outs[1:]))
    print(f"DCUnet full width encoder {i}: forward {e_fwd:.2e} (worst element {u_fwd / 2 ** -8:.2f} ulp)  input gradient {e_dg:.2e} ({u_dg / 2 ** -8:.2f} ulp)  "
          f"weight gradient {e_wg:.2e}")
    assert e_fwd < OUT_TOL and e_dg < OUT_TOL and e_wg < ACC_TOL and max(u_fwd, u_dg) < ULP_TOL, (i, e_fwd, e_dg, e_wg, u_fwd, u_dg)


@pytest.mark.parametrize("j", range(5))
def test_decoder_products_full_width(full, j):
    ws, p, sz = full["ws"], full["p"], full["sz"]
    b, n = ws.bufs, 5
    c1, c2, cout = sz["dec_ch"][j], sz["enc_ch"][n - j], sz["dec_ch"][j + 1]
    skip = d64(to_ref(b[f"ze{n - 1 - j}"], c2)).requires_grad_(True)
    if j == 0:
        leaves, cat = [skip], skip
    else:
        a = d64(to_ref(b[f"zd{j - 1}"], c1)).requires_grad_(True)
        leaves, cat = [a, skip], torch.cat([a, skip], dim=1)
    q = {k: d64(bf(v) if k.endswith("weight") else v).clone().requires_grad_(True) for k, v in p.items() if k.startswith(f"decoder{j}.transconv.")}
    y = D.complex_conv_transpose2d(cat, q, f"decoder{j}.transconv.", sz["dec_s"][j], sz["dec_p"][j])
    e_fwd, u_fwd = out_err(to_ref(b[f"yd{j}"], cout), y.detach() - d64(eff_bias(q, f"decoder{j}.transconv.tconv")))
    dy = d64(to_ref(b[f"dyd{j}"], cout))              # (j = 4: written by the fused tail)
    names = sorted(k for k in q if k.endswith("weight"))
    outs = torch.autograd.grad((y * dy).sum(), leaves + [q[k] for k in names])
    if j == 0:
        e_dg, u_dg = out_err(to_ref(b["dze4"], c2), outs[0])
    else:
        (e1, u1), (e2, u2) = out_err(to_ref(b[f"dzd{j - 1}"], c1), outs[0]), out_err(to_ref(b[f"dskip{n - 1 - j}"], c2), outs[1])
        e_dg, u_dg = max(e1, e2), max(u1, u2)
    e_wg = max(rel_err(d64(full["grads"][k]), gref) for k, gref in zip(names, outs[len(leaves):]))
    print(f"DCUnet full width decoder {j}: forward {e_fwd:.2e} (worst element {u_fwd / 2 ** -8:.2f} ulp)  input gradients {e_dg:.2e} ({u_dg / 2 ** -8:.2f} ulp)  "
          f"weight gradient {e_wg:.2e}")
    assert e_fwd < OUT_TOL and e_dg < OUT_TOL and e_wg < ACC_TOL and max(u_fwd, u_dg) < ULP_TOL, (j, e_fwd, e_dg, e_wg, u_fwd, u_dg)


@pytest.mark.parametrize("tag", [f"e{i}" for i in range(5)] + [f"d{j}" for j in range(4)])
def test_batchnorm_leakyrelu_full_width(full, tag):
    """(decoder 4's BatchNorm lives in the fused tail: test_fused_tail_full_width)"""
    ws, p, sz = full["ws"], full["p"], full["sz"]
    b = ws.bufs
    enc, idx = tag[0] == "e", int(tag[1])
    cr = sz["enc_ch"][idx + 1] if enc else sz["dec_ch"][idx + 1]
    pre = f"encoder{idx}.bn." if enc else f"decoder{idx}.bn."
    y = d64(to_ref(b[("ye" if enc else "yd") + str(idx)], cr)).requires_grad_(True)
    q = {k: d64(v).clone() for k, v in p.items() if k.startswith(pre)}
    leaves = {k: q[k].requires_grad_(True) for k in q if k.endswith((".weight", ".bias"))}
    z = F.leaky_relu(D.complex_batchnorm2d(y, q, pre, True), 0.01)
    e_fwd = rel_err(d64(to_ref(b[("ze" if enc else "zd") + str(idx)], cr)), z.detach())
    dz = d64(to_ref(b[("dze" if enc else "dzd") + str(idx)], cr))
    names = sorted(leaves)
    outs = torch.autograd.grad((z * dz).sum(), [y] + [leaves[k] for k in names])
    e_dy = rel_err(d64(to_ref(b[("dye" if enc else "dyd") + str(idx)], cr)), outs[0])
    e_pg = max(rel_err(d64(full["grads"][k]), gref) for k, gref in zip(names, outs[1:]))
    print(f"DCUnet full width BatchNorm + LeakyReLU {tag}: forward {e_fwd:.2e}  dy {e_dy:.2e}  weight / bias gradients {e_pg:.2e}")
    assert e_fwd < OUT_TOL and e_dy < OUT_TOL and e_pg < ACC_TOL, (tag, e_fwd, e_dy, e_pg)


def test_fused_tail_full_width(full):
    """csrc/dcunet.hip: last BatchNorm + LeakyReLU + 1 x 1 complex conv + tanh + mask (src/model/dcunet.py:40-50, :93-95, :131-159),
    forward and backward, from the stored pre-BatchNorm tensor: the 62-channel tensor the fused tail never writes."""
    t = full
    ws, p, sz = t["ws"], t["p"], t["sz"]
    b = ws.bufs
    cr = sz["dec_ch"][-1]
    pre = "decoder4.bn."
    y = d64(to_ref(b["yd4"], cr)).requires_grad_(True)
    q = {k: d64(v).clone() for k, v in p.items() if k.startswith((pre, "linear."))}
    leaves = {k: q[k].requires_grad_(True) for k in q if k.endswith((".weight", ".bias"))}
    z = F.leaky_relu(D.complex_batchnorm2d(y, q, pre, True), 0.01)
    mask = torch.tanh(D.complex_conv2d(z, q, "linear.", 1, 0)).transpose(2, 3)
    x = d64(t["x"])
    real, imag = x[..., 0], x[..., 1]
    mr, mi = mask[..., 0], mask[..., 1]
    x_mag, x_phase = torch.sqrt(real ** 2 + imag ** 2 + 1e-8), torch.atan2(imag, real)
    mm = (mr ** 2 + mi ** 2) ** 0.5
    ph = x_phase + torch.atan2(mi / (mm + 1e-8), mr / (mm + 1e-8))
    est = torch.stack([torch.tanh(mm) * x_mag * torch.cos(ph), torch.tanh(mm) * x_mag * torch.sin(ph)], dim=-1)
    e_est = rel_err(d64(t["est"]), est.detach())
    dout = 2.0 * (d64(t["est"]) - d64(t["tgt"])) / t["tgt"].numel()
    names = sorted(leaves)
    outs = torch.autograd.grad((est * dout).sum(), [y] + [leaves[k] for k in names])
    e_dy = rel_err(d64(to_ref(b["dyd4"], cr)), outs[0])
    e_pg = {k: rel_err(d64(t["grads"][k]), gref) for k, gref in zip(names, outs[1:])}
    print(f"DCUnet full width fused tail: output {e_est:.2e}  dy {e_dy:.2e}  parameter gradients {max(e_pg.values()):.2e} ({max(e_pg, key=e_pg.get)})")
    assert e_est < 1e-4 and e_dy < OUT_TOL and max(e_pg.values()) < ACC_TOL, (e_est, e_dy, e_pg)
