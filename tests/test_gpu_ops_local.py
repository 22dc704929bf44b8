"""GPU, op-local ("teacher-forced") parity: after one full forward/backward through libsehip, every operator is
re-computed on the CPU with the oracle's functions FROM THE HIP PATH'S OWN INPUTS (its stored bf16 activations and
activation gradients, its weights) and compared with what the HIP kernels produced.  Because both sides start from
identical inputs, the tolerances are tight (one bf16 rounding of the output, fp32 accumulation order) -- unlike the
whole-chain tests, where rounding flips compound through 11 BatchNorms."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dccrn_oracle as O
from util import rel_err

pytestmark = pytest.mark.gpu
SMALL = dict(kernel_num=[16, 32, 64, 64, 128, 128], rnn_units=128, length=4000)


def to_ref(buf, t0=0):
    """channels-last [B,T(+1),F,C] bf16 device buffer -> oracle layout [B,C,F,T] fp32 (logical frames)."""
    x = buf.t.float().cpu()
    if t0:
        x = x[:, t0:]
    return x.permute(0, 3, 2, 1).contiguous()


def bf(x):
    return x.to(torch.bfloat16).float()


def build_run(kw, B, N, seed=4):
    """One forward/backward of DCCRN(**kw) on B clips of N samples through libsehip; returns the workspace (every stored
    activation / activation gradient), the weights and the parameter gradients.  tests/test_gpu_c1_fullsize.py reuses
    this and the tests below at the headline configuration."""
    from sehip.model import DCCRN
    from sehip import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    model = DCCRN(**kw).to(dev).train()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # non-trivial biases / affine terms
        for name, p in model.named_parameters():
            if name.endswith(".bias") or name.endswith((".Br", ".Bi")):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            if name.endswith("2.weight"):
                p.copy_(0.25 + 0.1 * torch.randn(p.shape, generator=g))
    clean = 0.1 * torch.randn(B, 1, N, generator=g)
    noisy = clean + 0.05 * torch.randn(B, 1, N, generator=g)
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith(("stft.", "istft."))}
    est = model(noisy.to(dev))
    loss, rowstat = ops.sisnr_fwd(est.reshape(B, -1).contiguous(), clean.reshape(B, -1).contiguous().to(dev))
    est.backward(ops.sisnr_bwd(est.detach().reshape(B, -1).contiguous(), clean.reshape(B, -1).contiguous().to(dev), rowstat).view_as(est))
    torch.cuda.synchronize()
    ws = model.workspace(B, N)
    grads = {}
    L = model.static.layout
    gflat = model.flat_grads.cpu()
    for name in L.param_names:
        off, shape = L.param_off[name]
        grads[name] = gflat[off:off + int(np.prod(shape))].reshape(shape)
    return dict(ws=ws, p=p, grads=grads, B=B, T=ws.T, cfg=O.DCCRNConfig(**kw), kn=[2] + list(kw["kernel_num"]), model=model, noisy=noisy, clean=clean)


@pytest.fixture(scope="module")
def run():
    return build_run(SMALL, 3, 4000)


def conv_params(p, pre):
    return (bf(p[pre + "0.real_conv.weight"]), p[pre + "0.real_conv.bias"], bf(p[pre + "0.imag_conv.weight"]),
            p[pre + "0.imag_conv.bias"])


@pytest.mark.parametrize("i", range(6))
def test_encoder_conv_forward_dgrad_wgrad(run, i):
    ws, p = run["ws"], run["p"]
    b = ws.bufs
    x = to_ref(b["enc_in"] if i == 0 else b[f"z{i - 1}"]).requires_grad_(True)
    wr, br, wi, bi = conv_params(p, f"encoder.{i}.")
    wr.requires_grad_(True); wi.requires_grad_(True); br = br.clone().requires_grad_(True); bi = bi.clone().requires_grad_(True)
    y = O.complex_conv2d(x, wr, br, wi, bi)
    assert rel_err(to_ref(b[f"y{i}"]), y.detach()) < 4e-3           # one bf16 rounding of the output
    dy = to_ref(b[f"dye{i}"])
    gx, gwr, gbr, gwi, gbi = torch.autograd.grad((y * dy).sum(), [x, wr, br, wi, bi])
    if i > 0:
        # the dgrad product also adds the gradient that arrived over the skip connection (descriptor field `res`)
        from sehip.plan import FUSE_SKIP_GRAD
        want = gx + to_ref(b[f"dskip{i - 1}"]) if FUSE_SKIP_GRAD else gx
        assert rel_err(to_ref(b[f"dz{i - 1}"]), want) < 6e-3
    G = run["grads"]
    pre = f"encoder.{i}.0."
    assert rel_err(G[pre + "real_conv.weight"], gwr) < 5e-3 and rel_err(G[pre + "imag_conv.weight"], gwi) < 5e-3
    assert float((G[pre + "real_conv.bias"] - gbr).norm()) < 5e-3 * float(gwr.norm()) + 1e-6


@pytest.mark.parametrize("j", range(6))
def test_decoder_deconv_forward_dgrad_wgrad(run, j):
    ws, p = run["ws"], run["p"]
    b = ws.bufs
    a = (to_ref(b["P"]) if j == 0 else to_ref(b[f"zd{j - 1}"], 1)).requires_grad_(True)
    skip = to_ref(b[f"z{5 - j}"]).requires_grad_(True)
    wr, br, wi, bi = conv_params(p, f"decoder.{j}.")
    wr.requires_grad_(True); wi.requires_grad_(True)
    full = O.complex_deconv2d(O.complex_cat(a, skip), wr, br, wi, bi)   # [B,C,2F,T+1]
    if j < 5:
        got = b[f"yd{j}"].t.float().cpu().permute(0, 3, 2, 1)
        assert rel_err(got, full.detach()) < 4e-3
        dfull = b[f"dyd{j}"].t.float().cpu().permute(0, 3, 2, 1)
    else:
        got = b["mask"].t.cpu().permute(0, 3, 2, 1)
        assert rel_err(got, full.detach()[..., 1:]) < 2e-5            # fp32 output
        dfull = torch.zeros_like(full)
        dfull[..., 1:] = b["dmask"].t.float().cpu().permute(0, 3, 2, 1)
    ga, gskip, gwr, gwi = torch.autograd.grad((full * dfull).sum(), [a, skip, wr, wi])
    d1 = to_ref(b["dP"]) if j == 0 else to_ref(b[f"dzd{j - 1}"], 1)
    assert rel_err(d1, ga) < 6e-3 and rel_err(to_ref(b[f"dskip{5 - j}"]), gskip) < 6e-3
    G = run["grads"]
    pre = f"decoder.{j}.0."
    assert rel_err(G[pre + "real_conv.weight"], gwr) < 5e-3 and rel_err(G[pre + "imag_conv.weight"], gwi) < 5e-3


@pytest.mark.parametrize("name", [f"encoder.{i}." for i in range(6)] + [f"decoder.{j}." for j in range(5)])
def test_complex_batchnorm_prelu_forward_backward(run, name):
    ws, p = run["ws"], run["p"]
    b = ws.bufs
    idx = name.split(".")[1]
    enc = name.startswith("encoder")
    y = (to_ref(b[f"y{idx}"]) if enc else b[f"yd{idx}"].t.float().cpu().permute(0, 3, 2, 1).contiguous()).requires_grad_(True)
    q = {k: v.clone() for k, v in p.items() if k.startswith(name)}
    for k in ("RMr", "RMi", "RVri"):
        q[name + "1." + k].zero_()
    leaves = {k: q[name + "1." + k].requires_grad_(True) for k in ("Wrr", "Wri", "Wii", "Br", "Bi")}
    slope = q[name + "2.weight"].requires_grad_(True)
    z = F.prelu(O.complex_batchnorm(y, q, name + "1.", True), slope)
    zh = to_ref(b[f"z{idx}"]) if enc else b[f"zd{idx}"].t.float().cpu().permute(0, 3, 2, 1)
    sl = slice(None) if enc else slice(1, None)  # the dropped decoder frame is never written
    assert rel_err(zh[..., sl], z.detach()[..., sl]) < 4e-3
    if enc:
        from sehip.plan import FUSE_SKIP_GRAD
        dz = to_ref(b["dz5l"] if idx == "5" else b[f"dz{idx}"])
        if not FUSE_SKIP_GRAD:                    # otherwise the skip gradient is already inside dz (descriptor field `res` of the producer)
            dz = dz + to_ref(b[f"dskip{idx}"])
        dyh = to_ref(b[f"dye{idx}"])
    else:
        dz = b[f"dzd{idx}"].t.float().cpu().permute(0, 3, 2, 1).clone()
        dz[..., 0] = 0
        dyh = b[f"dyd{idx}"].t.float().cpu().permute(0, 3, 2, 1)
    outs = torch.autograd.grad((z * dz).sum(), [y, slope] + list(leaves.values()))
    assert rel_err(dyh, outs[0]) < 8e-3
    G = run["grads"]
    assert rel_err(G[name + "2.weight"], outs[1]) < 5e-3
    for k, gref in zip(leaves, outs[2:]):
        assert rel_err(G[name + "1." + k], gref) < 5e-3, k


def test_complex_lstm_forward_backward(run):
    """Both layers: recurrent kernels + the input / projection products, against the oracle's explicit recurrence with the
    same bf16 rounding points, from the HIP path's z5 and dP."""
    _check_complex_lstm(run)


@pytest.mark.parametrize("layers,units", [(1, 128), (3, 128), (2, 256), (3, 256), (2, 64), (2, 192)])
def test_complex_lstm_other_depths_and_widths(layers, units):
    """rnn_layers / rnn_units of the reference constructor (src/model/dccrn.py:12-13, the stack :84-96): the same op-local gate for the
    per-layer launches (csrc/lstm.hip at hidden 64 / 128) and the products between the layers."""
    _check_complex_lstm(build_run(dict(SMALL, rnn_layers=layers, rnn_units=units), 3, 4000))


@pytest.mark.parametrize("units", [128, 64, 32, 96])
def test_real_lstm_forward_backward(units):
    """DCCRN(use_clstm=False) (src/model/dccrn.py:98-106, :184-189): one real two-layer nn.LSTM over all channels + the `tranform`
    Linear -- sehip_rlstm_fwd / _bwd and the products around them against the oracle's explicit recurrence with the same bf16 rounding
    points, from the HIP path's z5 and dP."""
    import torch.nn.functional as F
    run = build_run(dict(SMALL, use_clstm=False, rnn_units=units), 3, 4000)
    ws, p = run["ws"], run["p"]
    b, B, T = ws.bufs, run["B"], run["T"]
    z5 = to_ref(b["z5"])                                  # [B,C,4,T]
    ch = z5.shape[1]
    x = z5.permute(3, 0, 1, 2).reshape(T, B, -1).requires_grad_(True)
    names = [k for k in p if k.startswith(("enhance.", "tranform."))]
    assert len(names) == 10
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    y = x
    for layer in range(2):
        y = O.lstm_single(y, leaves[f"enhance.weight_ih_l{layer}"], leaves[f"enhance.weight_hh_l{layer}"],
                          leaves[f"enhance.bias_ih_l{layer}"], leaves[f"enhance.bias_hh_l{layer}"], O.Bf16Sim)
    y = F.linear(y, O.Bf16Sim.weight(leaves["tranform.weight"]), leaves["tranform.bias"])
    out = y.reshape(T, B, ch, 4).permute(1, 2, 3, 0)      # [B,C,4,T]
    assert rel_err(to_ref(b["P"]), out.detach()) < 1e-2
    dP = to_ref(b["dP"])
    outs = torch.autograd.grad((out * dP).sum(), [x] + [leaves[k] for k in names])
    from sehip.plan import FUSE_SKIP_GRAD
    dz5, floor = to_ref(b["dz5l"]), 0.0
    if FUSE_SKIP_GRAD:
        floor = 2.0 ** -9 * float(dz5.norm() + to_ref(b["dskip5"]).norm()) / 2 ** 0.5
        dz5 = dz5 - to_ref(b["dskip5"])
    e = float((dz5.permute(3, 0, 1, 2).reshape(T, B, -1) - outs[0]).norm())
    assert e < 3e-2 * float(outs[0].norm()) + floor, (e, float(outs[0].norm()), floor)
    G = run["grads"]
    for k, gref in zip(names, outs[1:]):
        assert rel_err(G[k], gref) < 3e-2, k


def _check_complex_lstm(run):
    ws, p, cfg = run["ws"], run["p"], run["cfg"]
    b, B, T = ws.bufs, run["B"], run["T"]
    z5 = to_ref(b["z5"])                                  # [B,C,4,T]
    ch = z5.shape[1]
    seq = z5.permute(3, 0, 1, 2)
    r_in = seq[:, :, : ch // 2].reshape(T, B, -1).requires_grad_(True)
    i_in = seq[:, :, ch // 2:].reshape(T, B, -1).requires_grad_(True)
    names = [k for k in p if k.startswith("enhance.")]
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    r2, i2 = r_in, i_in
    for layer in range(cfg.rnn_layers):
        r2, i2 = O.complex_lstm(r2, i2, leaves, f"enhance.{layer}.", layer == cfg.rnn_layers - 1, sim=O.Bf16Sim)
    out = torch.cat([r2.reshape(T, B, ch // 2, 4), i2.reshape(T, B, ch // 2, 4)], 2).permute(1, 2, 3, 0)  # [B,C,4,T]
    assert rel_err(to_ref(b["P"]), out.detach()) < 1e-2
    dP = to_ref(b["dP"])
    outs = torch.autograd.grad((out * dP).sum(), [r_in, i_in] + [leaves[k] for k in names])
    from sehip.plan import FUSE_SKIP_GRAD
    dz5 = to_ref(b["dz5l"])
    floor = 0.0
    if FUSE_SKIP_GRAD:        # the dx1 products store the LSTM's input gradient + the innermost skip connection's (descriptor field `res`)
        # (both tensors are bf16: the difference carries their rounding, 2^-9 of each at most -- what is left of a small input gradient
        #  under a large skip gradient; a deeper stack at initialisation has exactly that)
        floor = 2.0 ** -9 * float(dz5.norm() + to_ref(b["dskip5"]).norm()) / 2 ** 0.5
        dz5 = dz5 - to_ref(b["dskip5"])
    dz5 = dz5.permute(3, 0, 1, 2)
    for part, ref in ((dz5[:, :, : ch // 2], outs[0]), (dz5[:, :, ch // 2:], outs[1])):
        e = float((part.reshape(T, B, -1) - ref).norm())
        assert e < 3e-2 * float(ref.norm()) + floor, (e, float(ref.norm()), floor)
    G = run["grads"]
    for k, gref in zip(names, outs[2:]):
        assert rel_err(G[k], gref) < 3e-2, k
