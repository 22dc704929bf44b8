"""ConvTasNet at FULL WIDTH (N128 L40 B128 H256 P3 X7 R2, two speakers: the C4 network), every product and normalisation kernel of
the forward and backward pass OP-LOCALLY, for ALL 14 temporal blocks, against float64 arithmetic on the operands the HIP path read
(its own stored bf16 tensors, bf16-rounded 1x1 weights).  The counterpart of tests/test_gpu_dcunet_fullwidth.py (VERDICT r5 weak #1:
the whole-chain gradient gate must allow the 5 % that bf16 STORAGE does to a network with 28 PReLUs and a ReLU mask; here nothing
non-smooth is chained, so a stored tensor may differ by ONE bf16 rounding and an fp32-accumulated gradient by its summation order).
SEHIP_CTN_KEEP_GRADS=1 gives every block its own du / dh2 (the plan otherwise shares one pair): same kernels, same launches."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import convtasnet_oracle as CT
from util import rel_err

pytestmark = pytest.mark.gpu

OUT_TOL = 1.8e-3              # rms of ONE round-to-nearest bf16 rounding is 1.65e-3 (tests/test_gpu_dcunet_fullwidth.py)
ULP_TOL = 2.0 ** -8 * 1.02
ACC_TOL = 2e-5                # products' weight gradients: fp32 accumulation order
NORM_TOL = 2e-4               # gLN / PReLU / depthwise parameter gradients: per-workgroup fp32 partial rows of 1e5 ... 1e6 addends
M, T = 2, 8000


def bf(x):
    return x.to(torch.bfloat16).float()


def out_err(got, want):
    got, want = got.double(), want.double()
    floor = 1e-3 * float(want.pow(2).mean().sqrt())
    return rel_err(got, want), float(((got - want).abs() / (want.abs() + floor)).max())


@pytest.fixture(scope="module")
def full():
    from sehip.model import ConvTasNet
    old = os.environ.get("SEHIP_CTN_KEEP_GRADS")
    os.environ["SEHIP_CTN_KEEP_GRADS"] = "1"
    try:
        torch.manual_seed(15)
        model = ConvTasNet(sources=["None", "None"], audio_channels=1).cuda().train()
        p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(16)
        mix = 0.3 * torch.randn(M, 1, T, generator=g)
        est = model(mix.cuda())
        G = torch.randn(est.shape, generator=g) / est.numel() ** 0.5
        est.backward(G.cuda())
        torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop("SEHIP_CTN_KEEP_GRADS", None)
        else:
            os.environ["SEHIP_CTN_KEEP_GRADS"] = old
    ws = model.workspace(M, T)
    assert ws.st.keep_grads
    grads = {k: v.grad.detach().cpu().clone() for k, v in model.named_parameters()}
    tr = lambda name: ws.bufs[name].t.float().cpu()[:, :, 0].transpose(1, 2).contiguous().double()      # [M, K, 1, C] -> [M, C, K]
    return dict(model=model, ws=ws, p=p, grads=grads, tr=tr)


def _product(full, wname, src, dst, dout, dsrc, res_fwd=None, res_bwd=None):
    """One 1x1 convolution: forward dst = W src (+ res_fwd), weight gradient, input gradient dsrc = W^T dout (+ res_bwd)."""
    tr, p, G = full["tr"], full["p"], full["grads"]
    x = tr(src).requires_grad_(True)
    w = bf(p[wname]).double().requires_grad_(True)
    y = F.conv1d(x, w)
    want = y.detach() + (tr(res_fwd) if res_fwd else 0)
    e_fwd, u_fwd = out_err(tr(dst), want)
    dy = tr(dout)
    dx, dw = torch.autograd.grad((y * dy).sum(), [x, w])
    e_wg = rel_err(G[wname].double(), dw)
    e_dg = u_dg = 0.0
    if dsrc is not None:
        e_dg, u_dg = out_err(tr(dsrc), dx + (tr(res_bwd) if res_bwd else 0))
    return e_fwd, u_fwd, e_wg, e_dg, u_dg


def test_bottleneck_and_mask_products(full):
    nb = len(full["ws"].st.blocks)
    net = "separator.network."
    for tag, args in (("bottleneck", (net + "1.weight", "cln", "x0", "dx0", "dcln")),
                      ("mask", (net + "3.weight", f"x{nb}", "mlin", "dmlin", f"dx{nb}"))):
        e_fwd, u_fwd, e_wg, e_dg, u_dg = _product(full, *args)
        print(f"ConvTasNet full width {tag}: forward {e_fwd:.2e} ({u_fwd / 2 ** -8:.2f} ulp)  weight gradient {e_wg:.2e}  input gradient {e_dg:.2e} ({u_dg / 2 ** -8:.2f} ulp)")
        assert e_fwd < OUT_TOL and e_dg < OUT_TOL and max(u_fwd, u_dg) < ULP_TOL and e_wg < ACC_TOL, (tag, e_fwd, u_fwd, e_wg, e_dg, u_dg)


@pytest.mark.parametrize("i", range(14))
def test_block_products(full, i):
    """the two 1x1 convolutions of temporal block i (src/model/conv_tasnet.py:307-402): forward (the second with the residual), weight
    gradients, input gradients (the first adds the gradient that arrives over the residual connection)"""
    st = full["ws"].st
    r, x = st.blocks[i]
    q = f"separator.network.2.{r}.{x}.net."
    a = _product(full, q + "0.weight", f"x{i}", f"h1_{i}", f"dh1_{i}", f"dx{i}", res_bwd=f"dx{i + 1}")
    b = _product(full, q + "3.pointwise_conv.weight", f"u{i}", f"x{i + 1}", f"dx{i + 1}", st.du_name(i), res_fwd=f"x{i}")
    for tag, (e_fwd, u_fwd, e_wg, e_dg, u_dg) in (("1x1 in", a), ("1x1 out", b)):
        print(f"ConvTasNet full width block {i} {tag}: forward {e_fwd:.2e} ({u_fwd / 2 ** -8:.2f} ulp)  weight gradient {e_wg:.2e}  input gradient {e_dg:.2e} ({u_dg / 2 ** -8:.2f} ulp)")
        assert e_fwd < OUT_TOL and e_dg < OUT_TOL and max(u_fwd, u_dg) < ULP_TOL and e_wg < ACC_TOL, (i, tag, e_fwd, u_fwd, e_wg, e_dg, u_dg)


@pytest.mark.parametrize("i", range(14))
def test_block_streams(full, i):
    """PReLU + gLN + depthwise dilated convolution + PReLU + gLN of block i (csrc/tasnet.hip), forward and backward, from the stored
    h1 / h2 / du / dh2 of THIS block"""
    ws, p, G, tr = full["ws"], full["p"], full["grads"], full["tr"]
    st = ws.st
    r, x = st.blocks[i]
    q = f"separator.network.2.{r}.{x}.net."
    keys = ("1.weight", "2.gamma", "2.beta", "3.net.0.weight", "3.net.1.weight", "3.net.2.gamma", "3.net.2.beta")
    leaves = {k: p[q + k].double().clone().requires_grad_(True) for k in keys}
    h1 = tr(f"h1_{i}").requires_grad_(True)
    n1 = CT.gln(F.prelu(h1, leaves["1.weight"]), leaves["2.gamma"], leaves["2.beta"])
    h2 = F.conv1d(n1, leaves["3.net.0.weight"], padding=2 ** x, dilation=2 ** x, groups=h1.shape[1])
    e_h2, u_h2 = out_err(tr(f"h2_{i}"), h2.detach())
    h2s = tr(f"h2_{i}").requires_grad_(True)                 # continue from the stored tensor
    u = CT.gln(F.prelu(h2s, leaves["3.net.1.weight"]), leaves["3.net.2.gamma"], leaves["3.net.2.beta"])
    e_u, u_u = out_err(tr(f"u{i}"), u.detach())
    outs2 = torch.autograd.grad((u * tr(st.du_name(i))).sum(), [h2s, leaves["3.net.1.weight"], leaves["3.net.2.gamma"], leaves["3.net.2.beta"]])
    e_dh2, u_dh2 = out_err(tr(st.dh2_name(i)), outs2[0])
    e_p2 = max(rel_err(G[q + k].double(), gref) for k, gref in zip(("3.net.1.weight", "3.net.2.gamma", "3.net.2.beta"), outs2[1:]))
    outs1 = torch.autograd.grad((h2 * tr(st.dh2_name(i))).sum(), [h1, leaves["1.weight"], leaves["2.gamma"], leaves["2.beta"], leaves["3.net.0.weight"]])
    e_dh1, u_dh1 = out_err(tr(f"dh1_{i}"), outs1[0])
    e_p1 = max(rel_err(G[q + k].double(), gref) for k, gref in zip(("1.weight", "2.gamma", "2.beta", "3.net.0.weight"), outs1[1:]))
    print(f"ConvTasNet full width block {i} streams: h2 {e_h2:.2e} ({u_h2 / 2 ** -8:.2f} ulp)  u {e_u:.2e} ({u_u / 2 ** -8:.2f})  dh2 {e_dh2:.2e} ({u_dh2 / 2 ** -8:.2f})  "
          f"dh1 {e_dh1:.2e} ({u_dh1 / 2 ** -8:.2f})  parameter gradients {e_p2:.2e} / {e_p1:.2e}")
    assert max(e_h2, e_u, e_dh2, e_dh1) < OUT_TOL and max(u_h2, u_u, u_dh2, u_dh1) < ULP_TOL, (i, e_h2, e_u, e_dh2, e_dh1, u_h2, u_u, u_dh2, u_dh1)
    assert max(e_p1, e_p2) < NORM_TOL, (i, e_p1, e_p2)
