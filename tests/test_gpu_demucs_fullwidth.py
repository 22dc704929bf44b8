"""Demucs at FULL WIDTH (channels 64, depth 6, every default: the C3 network, 133.7 M parameters), EVERY product launch of the forward and
backward pass -- 219 descriptors: the strided / transposed convolutions in their quad views, the dilated DConv convolutions, the 1x1
products, the BLSTM input / recurrent / projection products, LocalState's projections; forward, input gradients, weight and bias
gradients (the streaming dense-row kernel of csrc/dtw.hip, the generic kernel, the split-K launches) -- and EVERY GroupNorm / GELU / GLU /
LayerScale kernel, forward and backward, OP-LOCALLY against float64 arithmetic on the operands the HIP path itself read (its own stored
bf16 tensors, its packed bf16 weights).  The counterpart of tests/test_gpu_dcunet_fullwidth.py / test_gpu_convtasnet_fullwidth.py
(the Demucs full-width gate was a whole-chain bound of 6e-2 / 0.15 and the streaming weight gradients were checked
against another HIP kernel): here nothing non-smooth is chained, so a stored tensor may differ from the float64 result by ONE bf16
rounding and an fp32-accumulated gradient by its summation order.

The products are checked through the library's own descriptor semantics (include/sehip.h sehip_gemm_desc: the row (b, t) of the
implicit matrix A gathers 8-element chunks at (source, frame offset, element offset) with zeros outside the valid frame range; column n
goes to the destination column of its chunk) -- i.e. the kernels against the arithmetic they are asked for; that the descriptors ask for
the reference's convolutions is what the whole-chain oracle tests pin (tests/test_gpu_demucs.py::test_full_width_default_network)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

OUT_TOL = 1.8e-3              # rms of ONE round-to-nearest bf16 rounding is 1.65e-3 of the value (tests/test_gpu_dcunet_fullwidth.py)
ULP_TOL = 2.0 ** -8 * 1.02
F32_TOL = 2e-5                # fp32 destinations (the LSTM gate pre-activations) and fp32-accumulated weight gradients
NORM_TOL = 3e-4               # GroupNorm / LayerScale parameter gradients: per-workgroup fp32 partial rows of 1e4 ... 1e6 addends
B, T = 2, 24000
BF = torch.bfloat16


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-300))


def out_err(got, want):
    """(rms, max) of the ELEMENTWISE relative error |got - want| / (|want| + floor).  Elementwise, not norm-wise: LocalState's 16
    query_decay columns sit at -2 + small = the bottom of a binade with values ~30 x the other columns', where one correct rounding is
    2.2e-3 rms -- as a ratio of norms the product e4.d0.qkv read 2.4e-3 with every element inside one rounding (0.99)."""
    got, want = got.double(), want.double()
    floor = 1e-3 * float(want.pow(2).mean().sqrt())
    r = (got - want).abs() / (want.abs() + floor)
    return float(r.pow(2).mean().sqrt()), float(r.max())


def build_full():
    """One full-width step on ONE queue (SEHIP_NO_SIDE_STREAM, read when the workspace is built): these are checks of every kernel's
    arithmetic, not of the two-stream schedule -- beside the weight-gradient stream one of the 47 backward launches (e0.d0.n2) came out a
    few bf16 ulp off at a handful of near-cancelling elements in about half of the runs (its per-item sums 1e-4 ... 2e-3 off; DESIGN
    section 7, tools/dev/det_diff.py), which is a finding about the schedule, recorded there."""
    import os
    from sehip.model import Demucs
    from test_gpu_demucs import randomise_small_terms
    old = os.environ.get("SEHIP_NO_SIDE_STREAM")
    os.environ["SEHIP_NO_SIDE_STREAM"] = "1"
    try:
        return _build_full(Demucs, randomise_small_terms)
    finally:
        if old is None:
            os.environ.pop("SEHIP_NO_SIDE_STREAM", None)
        else:
            os.environ["SEHIP_NO_SIDE_STREAM"] = old


@pytest.fixture(scope="module")
def full():
    return build_full()


def _build_full(Demucs, randomise_small_terms):
    torch.manual_seed(41)
    model = Demucs(sources=["clean"], audio_channels=2)
    g = torch.Generator().manual_seed(42)
    randomise_small_terms(model, g)
    model = model.cuda().train()
    ws = model.workspace(B, T)
    calls = {"fwd": [], "bwd": []}
    nf, nb = ws._norm_fwd, ws._norm_bwd

    def norm_fwd(key, params, out, resid=None, add=None):
        calls["fwd"].append((key, out, resid, add))
        return nf(key, params, out, resid=resid, add=add)

    def norm_bwd(key, params, dz, dy):
        calls["bwd"].append((key, dz, dy))
        return nb(key, params, dz, dy)

    ws._norm_fwd, ws._norm_bwd = norm_fwd, norm_bwd
    mix = 0.3 * torch.randn(B, 2, T, generator=g) + 0.05
    est = model(mix.cuda())
    assert model.workspace(B, T) is ws
    G = torch.randn(est.shape, generator=g) / est.numel() ** 0.5
    est.backward(G.cuda())
    torch.cuda.synchronize()
    ws.check_lstm_handoffs()
    ws._norm_fwd, ws._norm_bwd = nf, nb
    flat = {b_.t.data_ptr(): b_.t for b_ in ws.bufs.values()}
    return dict(model=model, ws=ws, calls=calls, flat=flat, params=model.flat_params.detach())


# ---- the descriptor's arithmetic in float64 (torch on the GPU: none of libsehip) ----------------------------------------------------
def _table(base_tensor, ptr, n, width=4):
    off = (ptr - base_tensor.data_ptr()) // (4 * width)
    return base_tensor.view(-1, width)[off:off + n].to(torch.int64)


def gather_A(full, d):
    """[M, K] float64: row (b, t) of the implicit matrix (sehip_kchunk: source, (frame offset << 16) | row offset, element delta)."""
    ws, flat = full["ws"], full["flat"]
    dev = ws.gpack.device
    kt = _table(ws.ktab_dev, d.ktab, d.K // 8)
    m = torch.arange(d.M, device=dev)
    b, t = m // d.TT, m % d.TT
    tm = max(int(d.tmul), 1)
    A = torch.zeros(d.M, d.K // 8, 8, dtype=torch.float64, device=dev)
    eight = torch.arange(8, device=dev)
    for s in range(2):
        sel = (kt[:, 0] == s).nonzero().flatten()
        if sel.numel() == 0:
            continue
        src = d.src[s]
        x = flat[src.ptr].reshape(-1)
        foff = kt[sel, 1] >> 16                       # (arithmetic shift: negative frame offsets stay negative)
        fadd = kt[sel, 2]
        frame = (t * tm)[:, None] + foff[None, :]     # [M, chunks]
        ok = (frame >= src.tlo) & (frame < src.thi)
        base = ((b * src.T + t * tm) * src.F * src.C)[:, None] + fadd[None, :]
        idx = (base.clamp(0, x.numel() - 8)[:, :, None] + eight[None, None, :])
        A[:, sel] = x[idx].double() * ok[:, :, None]
    return A.reshape(d.M, d.K)


def dst_index(full, d):
    """element index of (row m, column n < N) in destination 0, and the tensor"""
    ws, flat = full["ws"], full["flat"]
    dev = ws.gpack.device
    nt = _table(ws.tb.ntab, d.ntab, d.Npad // 4)
    assert bool((nt[: (d.N + 3) // 4, 0] == 0).all())
    n = torch.arange(d.N, device=dev)
    col = nt[n // 4, 1] + (n % 4)
    m = torch.arange(d.M, device=dev)
    b, t = m // d.TT, m % d.TT
    ds = d.dst[0]
    row = ((b * ds.T + t * max(int(ds.tmul), 1) + ds.toff) * ds.F + ds.fadd) * ds.C
    return flat[ds.ptr].reshape(-1), row[:, None] + col[None, :]


def weights_of(full, d):
    tb = full["ws"].tb
    off = (d.W - tb.wpack.data_ptr()) // 2
    W = tb.wpack[off:off + d.Npad * d.K].view(d.Npad, d.K)[:d.N].double()
    bias = None
    if d.bias:
        boff = (d.bias - tb.bpack.data_ptr()) // 4
        bias = tb.bpack[boff:boff + d.N].double()
    return W, bias


def product_names(full):
    ws = full["ws"]
    fwd = sorted(k for k in ws.desc if not k.endswith(".wg"))
    wg = sorted(k for k in ws.desc if k.endswith(".wg"))
    return fwd, wg


def test_every_forward_and_input_gradient_product(full):
    """out = A W^T + bias (+ res): every forward product and every input-gradient product of the step, element by element"""
    ws, flat = full["ws"], full["flat"]
    fwd, _ = product_names(full)
    assert len(fwd) > 120
    worst = {"bf16": (0.0, 0.0, ""), "f32": (0.0, "")}
    skipped = []
    for name in fwd:
        d = ws.desc[name]
        # the LSTM kernels overwrite the gate pre-activations with the gates (csrc/demucs.hip dmx_lstm_seq_fwd_kernel): the product's
        # output no longer exists; tests/test_gpu_demucs.py::test_bidirectional_lstm_layer pins that path
        if ".ih" in name and not name.endswith(".dg"):
            skipped.append(name)
            continue
        A = gather_A(full, d)
        W, bias = weights_of(full, d)
        want = A @ W.t()
        if bias is not None:
            want = want + bias[None, :]
        out, idx = dst_index(full, d)
        if d.res:
            want = want + flat[d.res].reshape(-1)[idx].double()
        got = out[idx]
        if d.dst[0].is_f32:
            e = rel(got, want)
            worst["f32"] = max(worst["f32"], (e, name))
            assert e < F32_TOL, (name, e)
        else:
            e, u = out_err(got, want)
            worst["bf16"] = max(worst["bf16"], (e, u, name))
            assert e < OUT_TOL and u < ULP_TOL, (name, e, u / 2 ** -8)
    print(f"Demucs full width, {len(fwd) - len(skipped)} forward / input-gradient products: worst bf16 destination rms {worst['bf16'][0]:.2e} "
          f"({worst['bf16'][1] / 2 ** -8:.2f} ulp, {worst['bf16'][2]}), worst fp32 destination {worst['f32'][0]:.2e} ({worst['f32'][1]}); "
          f"{len(skipped)} LSTM input products overwritten by the recurrence, not compared")


def test_every_weight_and_bias_gradient(full):
    """dW = dOut^T A, dbias = column sums of dOut: every weight-gradient launch (streaming dense-row kernel, generic kernel) against
    float64 products of the tensors it read.  Products that share a bias (the phases of a transposed convolution) are summed."""
    ws = full["ws"]
    _, wg = product_names(full)
    assert len(wg) > 60
    gp = ws.gpack
    bias_sum, bias_names = {}, {}
    worst = (0.0, "")
    dense = 0
    for name in wg:
        d = ws.desc[name]
        A = gather_A(full, d)
        out, idx = dst_index(full, d)
        dO = out[idx].double()                                   # [M, N]
        dW = dO.t() @ A
        off = (d.dW - gp.data_ptr()) // 4
        got = gp[off:off + d.Npad * d.K].view(d.Npad, d.K)[:d.N].double()
        e = rel(got, dW)
        worst = max(worst, (e, name))
        dense += name in ws._dtw
        assert e < F32_TOL, (name, e, "streaming" if name in ws._dtw else "generic")
        if d.dbias:
            boff = (d.dbias - gp.data_ptr()) // 4
            bias_sum[boff] = bias_sum.get(boff, 0) + dO.sum(0)
            bias_names.setdefault(boff, []).append((name, d.N))
    for boff, s in bias_sum.items():
        n = bias_names[boff][0][1]
        e = rel(gp[boff:boff + n].double(), s)
        worst = max(worst, (e, "bias of " + "+".join(k for k, _ in bias_names[boff])))
        assert e < F32_TOL * 5, (bias_names[boff], e)
    print(f"Demucs full width, {len(wg)} weight gradients ({dense} on the streaming dense-row kernel) + {len(bias_sum)} bias gradients: worst {worst[0]:.2e} ({worst[1]})")


def _norm_ref(y, gamma, beta, G, mode, scale, resid, add):
    """[B, T, C] float64 -> GroupNorm(G) (or identity) + GELU / GLU (+ LayerScale + residual) (+ addend), src/model/demucs.py:139-207, :386-413"""
    x = y.transpose(1, 2)
    n = F.group_norm(x, G, gamma, beta, eps=1e-5) if G else x
    v = F.glu(n, dim=1) if mode else F.gelu(n)
    if scale is not None:
        v = resid.transpose(1, 2) + scale[:, None] * v
    if add is not None:
        v = v + add.transpose(1, 2)
    return v.transpose(1, 2)


def test_every_norm_and_activation_kernel(full):
    """every GroupNorm / GELU / GLU / LayerScale launch: forward output, input gradient, gamma / beta / scale gradients from the stored
    y, dz of THAT launch"""
    ws, calls, params = full["ws"], full["calls"], full["params"]
    st = ws.st
    fwd = {k: (out, resid, add) for k, out, resid, add in calls["fwd"]}
    bwd = {k: (dz, dy) for k, dz, dy in calls["bwd"]}
    assert len(fwd) >= 40 and set(bwd) <= set(fwd)
    pv = lambda name: params[st.layout.param_off[name][0]:st.layout.param_off[name][0] + int(np.prod(st.layout.param_off[name][1]))].double()
    tens = lambda name: ws.bufs[name].t[:, :, 0].double()
    worst_out, worst_dy, worst_g = (0.0, 0.0, ""), (0.0, 0.0, ""), (0.0, "")
    for key, (out, resid, add) in fwd.items():
        n = st.gch[key]
        G, mode = n["G"], n["mode"]
        leaves = {}
        if G:
            leaves["gamma"], leaves["beta"] = pv(n["gamma"]).requires_grad_(True), pv(n["beta"]).requires_grad_(True)
        if n["scale"]:
            leaves["scale"] = pv(n["scale"]).requires_grad_(True)
        y = tens(n["y"]).requires_grad_(True)
        v = _norm_ref(y, leaves.get("gamma"), leaves.get("beta"), G, mode, leaves.get("scale"), tens(resid) if resid else None,
                      tens(add) if add else None)
        e, u = out_err(tens(out), v.detach())
        worst_out = max(worst_out, (e, u, key))
        assert e < OUT_TOL and u < ULP_TOL, ("forward", key, e, u / 2 ** -8)
        if key not in bwd:
            continue
        dz, dy = bwd[key]
        outs = torch.autograd.grad((v * tens(dz)).sum(), [y] + list(leaves.values()))
        e, u = out_err(tens(dy), outs[0])
        worst_dy = max(worst_dy, (e, u, key))
        assert e < OUT_TOL and u < ULP_TOL, ("input gradient", key, e, u / 2 ** -8)
        if G:
            Cc, Co = n["C"], n["Co"]
            g = ws.gpack[n["off"]:n["off"] + 2 * Cc + Co].double()
            got = {"gamma": g[:Cc], "beta": g[Cc:2 * Cc], "scale": g[2 * Cc:]}
            for (nm, _), ref in zip(leaves.items(), outs[1:]):
                e = rel(got[nm], ref)
                worst_g = max(worst_g, (e, f"{key}.{nm}"))
                assert e < NORM_TOL, (key, nm, e)
    print(f"Demucs full width, {len(fwd)} norm / activation launches forward, {len(bwd)} backward: worst output {worst_out[0]:.2e} "
          f"({worst_out[1] / 2 ** -8:.2f} ulp, {worst_out[2]}), worst input gradient {worst_dy[0]:.2e} ({worst_dy[1] / 2 ** -8:.2f} ulp, {worst_dy[2]}), "
          f"worst parameter gradient {worst_g[0]:.2e} ({worst_g[1]})")
