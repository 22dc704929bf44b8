"""CPU: the host logic of sehip.plan_dcunet (strided chunk tables, parity classes of the transposed convolutions and of the
strided convolutions' input gradients, two-source skip concatenation, channel padding 5 -> 8 / 10 -> 16, packed-weight and
gradient un-packing tables) interpreted in numpy exactly as libsehip's kernels read them (include/sehip.h) and compared with
the oracle's complex convolutions (oracle/dcunet_oracle.py, pinned to the reference by tests/golden/dcunet_tiny.npz)."""
import numpy as np
import pytest
import torch

from oracle import dcunet_oracle as D
from util import load_golden

B, F0, T0 = 2, 257, 33


@pytest.fixture(scope="module")
def ctx():
    from sehip import plan_dcunet as P
    cfg = P.DCUNetConfig(data_type=True, model_complexity=8, model_depth=10)
    st = P.DCUNetStatic(cfg)
    pl = P.DCUNetPlan(st, F0, T0)
    g = load_golden("dcunet_tiny.npz")
    p = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    flat = np.zeros(st.layout.n_params, dtype=np.float64)
    for name in st.layout.param_names:
        off, shape = st.layout.param_off[name]
        flat[off:off + p[name].numel()] = p[name].reshape(-1).double().numpy()
    return dict(st=st, pl=pl, p=p, flat=flat, gen=torch.Generator().manual_seed(2), sz=D.dcunet_sizes(8, 10, 1))


def term(flat, e):
    e = np.asarray(e, dtype=np.int64)
    v = np.where(e >= 0, flat[np.maximum(e, 0) >> 1], 0.0)
    return np.where((e & 1) == 1, -v, v)


def run_spec(flat, spec, bufs):
    """One sehip_gemm_desc product in float64: row (b, t, j) reads source frame t*tmul + toff, row j*fmul + fadd; column
    groups go to ((b*T + t*dst_tmul + toff)*F + j*fmul + fadd)."""
    from sehip.plan import enc_entry
    K = spec.K
    W = term(flat, enc_entry(spec.widx, spec.wneg).reshape(-1)).reshape(spec.Npad, K)[:spec.N]
    M = B * spec.tt * spec.J
    m = np.arange(M)
    j, bt = m % spec.J, m // spec.J
    t, b = bt % spec.tt, bt // spec.tt
    A = np.zeros((M, K))
    for c, (src, toff, fadd, coff) in enumerate(spec.ktab):
        if src < 0:
            continue
        x = bufs[spec.srcs[src][0]]
        ts, f = t * spec.tmul + toff, j * spec.fmul + fadd
        if x.shape[3] == 2:
            for q in range(coff):
                ok = (ts >= 0) & (ts < x.shape[1]) & (f + q >= 0) & (f + q < x.shape[2])
                A[ok, c * 8 + 2 * q:c * 8 + 2 * q + 2] = x[b[ok], ts[ok], f[ok] + q]
        else:
            ok = (ts >= 0) & (ts < x.shape[1]) & (f >= 0) & (f < x.shape[2])
            A[ok, c * 8:c * 8 + 8] = x[b[ok], ts[ok], f[ok], coff:coff + 8]
    out = A @ W.T
    if spec.bias_pairs is not None:
        bp = spec.bias_pairs[:spec.N]
        out = out + (term(flat, bp[:, 0]) + term(flat, bp[:, 1]))[None, :]
    res = {}
    for q, (name, toff, fmul, fadd) in enumerate(spec.dsts):
        res[name] = dict(rows=(b, t * spec.dst_tmul[q] + toff, j * fmul + fadd), cols=[], vals=[])
    for n4 in range(spec.Npad // 4):
        dst, coff, nvalid, _ = spec.ntab[n4]
        for q in range(nvalid):
            res[spec.dsts[dst][0]]["cols"].append(coff + q)
            res[spec.dsts[dst][0]]["vals"].append(out[:, 4 * n4 + q])
    return res


def scatter_add(acc, res, name):
    r = res[name]
    b, t, f = r["rows"]
    for c, v in zip(r["cols"], r["vals"]):
        assert acc[name][b, t, f, c].max() == 0 and acc[name][b, t, f, c].min() == 0    # every element is written once
        acc[name][b, t, f, c] = v


def to_cl(x, cs):
    """oracle [B, C, T, F, 2] -> channels-last [B, T, F, 2*cs] (real half | imaginary half, zero padding channels)."""
    x = x.detach().double().numpy()
    b, c, t, f, _ = x.shape
    out = np.zeros((b, t, f, 2 * cs))
    out[..., :c] = x[..., 0].transpose(0, 2, 3, 1)
    out[..., cs:cs + c] = x[..., 1].transpose(0, 2, 3, 1)
    return out


def test_encoder_forward_and_input_gradient(ctx):
    pl, p, flat, sz, st, gen = ctx["pl"], ctx["p"], ctx["flat"], ctx["sz"], ctx["st"], ctx["gen"]
    dims = [(T0, F0)] + pl.enc_dims
    for i in range(5):
        cin, cout = sz["enc_ch"][i], sz["enc_ch"][i + 1]
        cs_in = 1 if i == 0 else st.enc_c[i - 1]
        x = torch.randn(B, cin, dims[i][0], dims[i][1], 2, generator=gen, requires_grad=True)
        y = D.complex_conv2d(x, p, f"encoder{i}.conv.", sz["enc_s"][i], sz["enc_p"][i])
        src = "x0" if i == 0 else f"ze{i - 1}"
        acc = {f"ye{i}": np.zeros((B,) + pl.enc_dims[i] + (2 * st.enc_c[i],))}
        scatter_add(acc, run_spec(flat, pl.specs[f"enc{i}.fwd"], {src: to_cl(x, cs_in)}), f"ye{i}")
        assert np.abs(acc[f"ye{i}"] - to_cl(y, st.enc_c[i])).max() < 1e-5, i
        if i == 0:
            continue
        dy = torch.randn(y.shape, generator=gen)
        (dx,) = torch.autograd.grad((y * dy).sum(), x)
        name = f"dze{i - 1}"
        acc = {name: np.zeros((B,) + dims[i] + (2 * cs_in,))}
        for k, s in pl.specs.items():
            if k.startswith(f"enc{i}.dg"):
                scatter_add(acc, run_spec(flat, s, {f"dye{i}": to_cl(dy, st.enc_c[i])}), name)
        assert np.abs(acc[name] - to_cl(dx, cs_in)).max() < 1e-5, i


def test_decoder_forward_skip_concat_and_input_gradients(ctx):
    pl, p, flat, sz, st, gen = ctx["pl"], ctx["p"], ctx["flat"], ctx["sz"], ctx["st"], ctx["gen"]
    n = 5
    for j in range(n):
        Tin, Fin = pl.enc_dims[4] if j == 0 else pl.dec_dims[j - 1]
        c1, c2 = sz["dec_ch"][j], sz["enc_ch"][n - j]
        skip = torch.randn(B, c2, Tin, Fin, 2, generator=gen, requires_grad=True)
        bufs = {}
        if j == 0:
            cat, leaves = skip, [skip]
            bufs["ze4"] = to_cl(skip, st.enc_c[4])
        else:
            a = torch.randn(B, c1, Tin, Fin, 2, generator=gen, requires_grad=True)
            cat, leaves = torch.cat([a, skip], dim=1), [a, skip]
            bufs[f"zd{j - 1}"], bufs[f"ze{n - 1 - j}"] = to_cl(a, st.dec_c[j - 1]), to_cl(skip, st.enc_c[n - 1 - j])
        y = D.complex_conv_transpose2d(cat, p, f"decoder{j}.transconv.", sz["dec_s"][j], sz["dec_p"][j])
        assert tuple(y.shape[2:4]) == pl.dec_dims[j]
        acc = {f"yd{j}": np.zeros((B,) + pl.dec_dims[j] + (2 * st.dec_c[j],))}
        for name in pl.bias_group[f"dec{j}"]:
            scatter_add(acc, run_spec(flat, pl.specs[name], bufs), f"yd{j}")
        assert np.abs(acc[f"yd{j}"] - to_cl(y, st.dec_c[j])).max() < 1e-5, j
        dy = torch.randn(y.shape, generator=gen)
        grads = torch.autograd.grad((y * dy).sum(), leaves)
        spec = pl.specs[f"dec{j}.dg"]
        inputs = ["ze4"] if j == 0 else [f"zd{j - 1}", f"ze{n - 1 - j}"]
        acc = {d[0]: np.zeros(bufs[name].shape) for d, name in zip(spec.dsts, inputs)}
        res = run_spec(flat, spec, {f"dyd{j}": to_cl(dy, st.dec_c[j])})
        for d in spec.dsts:
            scatter_add(acc, res, d[0])
        if j == 0:
            assert np.abs(acc["dze4"] - to_cl(grads[0], st.enc_c[4])).max() < 1e-5
        else:
            assert np.abs(acc[f"dzd{j - 1}"] - to_cl(grads[0], st.dec_c[j - 1])).max() < 1e-5, j
            assert np.abs(acc[f"dskip{n - 1 - j}"] - to_cl(grads[1], st.enc_c[n - 1 - j])).max() < 1e-5, j


def test_unpack_table_folds_every_packed_gradient(ctx):
    """grads = d/dparam of <gw, Wpacked(param)> + <gb, bias(param)> for random packed gradients of EVERY forward product
    (the parity classes of a transposed convolution share one bias-gradient region)."""
    from sehip.plan import enc_entry
    pl, st = ctx["pl"], ctx["st"]
    rng = np.random.default_rng(0)
    gpack = rng.standard_normal(pl.n_gpack)
    grads = np.zeros(st.layout.n_params)
    for q in range(4):
        e = pl.utab[:, q].astype(np.int64)
        ok = e >= 0
        grads[ok] += np.where((e[ok] & 1) == 1, -1.0, 1.0) * gpack[e[ok] >> 1]
    ref = np.zeros_like(grads)
    seen = set()
    for s in pl.specs.values():
        if s.dw_off is None:
            continue
        ent = enc_entry(s.widx, s.wneg).astype(np.int64)
        ok = ent >= 0
        gw = gpack[s.dw_off:s.dw_off + ent.size].reshape(ent.shape)
        np.add.at(ref, ent[ok] >> 1, np.where((ent[ok] & 1) == 1, -1.0, 1.0) * gw[ok])
        if s.db_off is not None and s.db_off not in seen:
            seen.add(s.db_off)
            for col in (0, 1):
                e = s.bias_pairs[:, col].astype(np.int64)
                ok = e >= 0
                np.add.at(ref, e[ok] >> 1, np.where((e[ok] & 1) == 1, -1.0, 1.0) * gpack[s.db_off + np.flatnonzero(ok)])
    L = st.layout
    for pre, tag, cs, cr in pl.bn:
        for k, leaf in (("w_re", "bn_re.weight"), ("b_re", "bn_re.bias"), ("w_im", "bn_im.weight"), ("b_im", "bn_im.bias")):
            ref[L.index_array(pre + leaf)] += gpack[pl.bn_g_off[pre][k] + np.arange(cr)]
    cs_l, cr_l = st.dec_c[-1], st.dec_cr[-1]
    ref[L.index_array("linear.conv_re.weight").reshape(-1)] += gpack[pl.lin_g_off + np.arange(cr_l)]
    ref[L.index_array("linear.conv_im.weight").reshape(-1)] += gpack[pl.lin_g_off + cs_l + np.arange(cr_l)]
    ref[L.index_array("linear.conv_re.bias")] += gpack[pl.lin_g_off + 2 * cs_l]
    ref[L.index_array("linear.conv_im.bias")] += gpack[pl.lin_g_off + 2 * cs_l + 1]
    assert np.abs(grads - ref).max() < 1e-12
    real = np.concatenate([L.index_array(n).reshape(-1) for n in L.param_names])          # (the flat layout pads tensors to 16 bytes)
    assert np.abs(grads[real]).min() > 0                                                    # every parameter receives a gradient


def test_module_schema_matches_reference_checkpoint():
    from sehip.model import DCUnet
    from sehip import SehipError
    g = load_golden("dcunet_tiny.npz")
    m = DCUnet(data_type=True, model_complexity=8, model_depth=10)
    sd = m.state_dict()
    assert len(sd) == int(g["n_state_dict_keys"])                  # incl. the encoders.* / decoders.* aliases of the reference
    ref = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    assert all(tuple(sd[k].shape) == tuple(v.shape) for k, v in ref.items())
    m.load_state_dict(ref, strict=False)                           # a checkpoint without the aliases fills both names
    assert torch.equal(m.state_dict()["encoder2.conv.conv_im.weight"], ref["encoder2.conv.conv_im.weight"])
    assert torch.equal(m.state_dict()["encoders.2.conv.conv_im.weight"], ref["encoder2.conv.conv_im.weight"])   # same tensor
    assert [n for n, _ in m.named_parameters()][:2] == ["encoder0.conv.conv_re.weight", "encoder0.conv.conv_re.bias"]
    with pytest.raises(SehipError):
        m(torch.zeros(1, 1, 257, 33, 2))                           # CPU tensor: no fallback
    with pytest.raises(SehipError):
        DCUnet(data_type=True, model_depth=12)                     # "Unknown model depth"


def test_depth_20_state_dict_matches_the_reference():
    """model_depth=20 (src/model/dcunet.py:215-305): names and shapes of the reference's state_dict (tests/golden/dcunet20_tiny.npz)."""
    from sehip.model import DCUnet
    g = load_golden("dcunet20_tiny.npz")
    m = DCUnet(data_type=True, model_complexity=8, model_depth=20)
    sd = m.state_dict()
    assert len(sd) == int(g["n_state_dict_keys"])
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    assert set(ref) <= set(sd) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in ref.items())
    assert tuple(sd["encoder9.conv.conv_re.weight"].shape) == (128, 10, 5, 3)      # the fixed 128-channel bottleneck (:226)
    assert tuple(sd["encoder0.conv.conv_re.weight"].shape[2:]) == (7, 1) and tuple(sd["decoder9.transconv.tconv_re.weight"].shape[2:]) == (7, 1)
