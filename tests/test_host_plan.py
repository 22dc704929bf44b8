"""CPU: the host logic of sehip.plan (chunk tables, packed-weight tables, two-source skip concatenation, parity split of
the transposed convolution, frame drop, LSTM permutations, gradient un-packing) interpreted in numpy and compared with
the oracle's convolutions -- no GPU needed."""
import numpy as np
import pytest
import torch

from oracle import dccrn_oracle as O

KW = dict(kernel_num=[16, 16, 16, 32, 32, 64], rnn_units=128, length=1200)


@pytest.fixture(scope="module")
def ctx():
    from sehip import plan
    cfg = plan.DCCRNConfig(**KW)
    st = plan.DCCRNStatic(cfg)
    p = O.init_params(O.DCCRNConfig(**KW), seed=1)
    g = torch.Generator().manual_seed(2)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g)
    flat = np.zeros(st.layout.n_params, dtype=np.float64)
    for name in st.layout.param_names:
        off, shape = st.layout.param_off[name]
        flat[off:off + p[name].numel()] = p[name].reshape(-1).double().numpy()
    return dict(st=st, p=p, flat=flat, g=g)


def term(flat, e):
    e = np.asarray(e, dtype=np.int64)
    v = np.where(e >= 0, flat[np.maximum(e, 0) >> 1], 0.0)
    return np.where((e & 1) == 1, -v, v)


def run_spec(st, flat, spec, bufs, T, B):
    """Evaluates one GemmSpec exactly as libsehip's kernels do (include/sehip.h), in float64."""
    tt = T if spec.tt == "T" else T + 1
    K = spec.K
    W = term(flat, plan_entry(spec)).reshape(spec.Npad, K)[:spec.N]
    M = B * tt * spec.J
    A = np.zeros((M, K))
    m = np.arange(M)
    j = m % spec.J
    bt = m // spec.J
    t, b = bt % tt, bt // tt
    for c, (src, toff, fadd, coff) in enumerate(spec.ktab):
        if src < 0:
            continue
        name, mode = spec.srcs[src]
        x = bufs[name]  # [B, Tst, F, C]
        tlo, thi = (1, T + 1) if mode == "drop1" else (0, x.shape[1])
        ts, f = t + toff, j * spec.fmul + fadd
        if x.shape[3] == 2:
            for q in range(coff):
                ok = (ts >= tlo) & (ts < thi) & (f + q >= 0) & (f + q < x.shape[2])
                A[ok, c * 8 + 2 * q:c * 8 + 2 * q + 2] = x[b[ok], ts[ok], f[ok] + q]
        else:
            ok = (ts >= tlo) & (ts < thi) & (f >= 0) & (f < x.shape[2])
            A[ok, c * 8:c * 8 + 8] = x[b[ok], ts[ok], f[ok], coff:coff + 8]
    out = A @ W.T
    if spec.bias_pairs is not None:
        bp = spec.bias_pairs[:spec.N]
        out = out + (term(flat, bp[:, 0]) + term(flat, bp[:, 1]))[None, :]
    res = {}
    for q, (name, toff, fmul, fadd) in enumerate(spec.dsts):
        res[name] = dict(rows=(b, t + toff, j * fmul + fadd), cols=[], vals=[])
    for n4 in range(spec.Npad // 4):
        dst, coff, nvalid, _ = spec.ntab[n4]
        name = spec.dsts[dst][0]
        for q in range(nvalid):
            res[name]["cols"].append(coff + q)
            res[name]["vals"].append(out[:, 4 * n4 + q])
    return res


def plan_entry(spec):
    from sehip.plan import enc_entry
    return enc_entry(spec.widx, spec.wneg).reshape(-1)


def scatter(res, name, shape):
    out = np.zeros(shape)
    r = res[name]
    b, t, f = r["rows"]
    for c, v in zip(r["cols"], r["vals"]):
        out[b, t, f, c] = v
    return out


def cl(x):
    return x.detach().permute(0, 3, 2, 1).double().numpy()  # [B,C,F,T] -> [B,T,F,C]


def test_encoder_conv_tables(ctx):
    st, p, flat, g = ctx["st"], ctx["p"], ctx["flat"], ctx["g"]
    B, T = 2, 7
    kn = st.cfg.kernel_num
    for i in (0, 1, 3):
        ci, co, fi = kn[i], kn[i + 1], 256 >> i
        x = torch.randn(B, ci, fi, T, generator=g)
        pre = f"encoder.{i}."
        ref = O.complex_conv2d(x, p[pre + "0.real_conv.weight"], p[pre + "0.real_conv.bias"],
                               p[pre + "0.imag_conv.weight"], p[pre + "0.imag_conv.bias"])
        src = "enc_in" if i == 0 else f"z{i - 1}"
        res = run_spec(st, flat, st.specs[f"enc{i}.fwd"], {src: cl(x)}, T, B)
        got = scatter(res, f"y{i}", (B, T, fi // 2, co))
        assert np.abs(got - cl(ref)).max() < 1e-5, i


def test_encoder_dgrad_tables(ctx):
    st, p, flat, g = ctx["st"], ctx["p"], ctx["flat"], ctx["g"]
    B, T, i = 2, 6, 2
    kn = st.cfg.kernel_num
    ci, co, fi = kn[i], kn[i + 1], 256 >> i
    x = torch.randn(B, ci, fi, T, generator=g, requires_grad=True)
    pre = f"encoder.{i}."
    y = O.complex_conv2d(x, p[pre + "0.real_conv.weight"], p[pre + "0.real_conv.bias"], p[pre + "0.imag_conv.weight"],
                         p[pre + "0.imag_conv.bias"])
    dy = torch.randn(y.shape, generator=g)
    (dx,) = torch.autograd.grad((y * dy).sum(), x)
    got = np.zeros((B, T, fi, ci))
    for par in (0, 1):
        res = run_spec(st, flat, st.specs[f"enc{i}.dg{par}"], {f"dye{i}": cl(dy)}, T, B)
        got += scatter(res, f"dz{i - 1}", (B, T, fi, ci))
    assert np.abs(got - cl(dx)).max() < 1e-5


@pytest.mark.parametrize("j", [0, 2, 5])
def test_decoder_tables_with_skip_concat_and_frame_drop(ctx, j):
    st, p, flat, g = ctx["st"], ctx["p"], ctx["flat"], ctx["g"]
    B, T = 2, 6
    kn = st.cfg.kernel_num
    idx = 6 - j
    c1 = kn[idx]
    co, f_in = kn[idx - 1], 256 >> idx
    a = torch.randn(B, c1, f_in, T, generator=g, requires_grad=True)     # previous decoder output (logical frames)
    skip = torch.randn(B, c1, f_in, T, generator=g, requires_grad=True)
    pre = f"decoder.{j}."
    full = O.complex_deconv2d(O.complex_cat(a, skip), p[pre + "0.real_conv.weight"], p[pre + "0.real_conv.bias"],
                              p[pre + "0.imag_conv.weight"], p[pre + "0.imag_conv.bias"])  # [B,co,2F,T+1]
    s1 = "P" if j == 0 else f"zd{j - 1}"
    a_cl = cl(a)
    if j > 0:  # stored with the dropped first frame in front
        a_cl = np.concatenate([np.full((B, 1, f_in, c1), 7.0), a_cl], 1)
    bufs = {s1: a_cl, f"z{5 - j}": cl(skip)}
    last = j == 5
    name = "mask" if last else f"yd{j}"
    got = np.zeros((B, T if last else T + 1, 2 * f_in, co))
    for par in (0, 1):
        res = run_spec(st, flat, st.specs[f"dec{j}.fwd{par}"], bufs, T, B)
        got += scatter(res, name, got.shape)
    ref = cl(full)[:, 1:] if last else cl(full)
    assert np.abs(got - ref).max() < 1e-5
    # dgrad into both sources
    dfull = torch.randn(full.shape, generator=g)
    if last:
        dfull[..., 0] = 0  # the dropped frame carries no gradient
    da, dskip = torch.autograd.grad((full * dfull).sum(), [a, skip])
    gname = "dmask" if last else f"dyd{j}"
    res = run_spec(st, flat, st.specs[f"dec{j}.dg"], {gname: cl(dfull)[:, 1:] if last else cl(dfull)}, T, B)
    d1 = "dP" if j == 0 else f"dzd{j - 1}"
    got1 = scatter(res, d1, (B, T if j == 0 else T + 1, f_in, c1))
    got2 = scatter(res, f"dskip{5 - j}", (B, T, f_in, c1))
    assert np.abs((got1 if j == 0 else got1[:, 1:]) - cl(da)).max() < 1e-5
    assert np.abs(got2 - cl(dskip)).max() < 1e-5


def test_split_decoder_products_add_up_to_the_two_source_product():
    """Round 6 (DCCRNStatic._maybe_split_decoder): the skip-connection half `dec{j}.fs{p}` and the main half `dec{j}.fm{p}` of a deep
    decoder's forward product, interpreted from their tables, add up to the two-source product `dec{j}.fwd{p}` -- i.e. to the
    reference's ComplexConvTranspose2d of complex_cat([main, skip]) (src/model/dccrn.py:186-197, :387-450); the bias rides in the
    main half only; both are forward-only (no weight-gradient twin, no rows in the un-packing table)."""
    import os
    from sehip import plan
    kw = dict(kernel_num=[16, 16, 32, 32, 64, 64], rnn_units=128, length=1200)      # (the shared fixture's widths have no 64-output decoder)
    old_env = os.environ.get("SEHIP_DEC_SPLIT")
    os.environ["SEHIP_DEC_SPLIT"] = "1"                 # (opt-in: measured slower in the step, see DCCRNStatic._maybe_split_decoder)
    try:
        st = plan.DCCRNStatic(plan.DCCRNConfig(**kw))
    finally:
        if old_env is None:
            os.environ.pop("SEHIP_DEC_SPLIT", None)
        else:
            os.environ["SEHIP_DEC_SPLIT"] = old_env
    p = O.init_params(O.DCCRNConfig(**kw), seed=3)
    g = torch.Generator().manual_seed(4)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g)
    flat = np.zeros(st.layout.n_params, dtype=np.float64)
    for name in st.layout.param_names:
        off, shape = st.layout.param_off[name]
        flat[off:off + p[name].numel()] = p[name].reshape(-1).double().numpy()
    assert st.dec_split, "no decoder layer of this configuration is split"
    B, T = 2, 6
    kn = st.cfg.kernel_num
    for j in st.dec_split:
        idx = 6 - j
        c1, co, f_in = kn[idx], kn[idx - 1], 256 >> idx
        a = torch.randn(B, c1, f_in, T, generator=g)
        skip = torch.randn(B, c1, f_in, T, generator=g)
        pre = f"decoder.{j}."
        full = O.complex_deconv2d(O.complex_cat(a, skip), p[pre + "0.real_conv.weight"], p[pre + "0.real_conv.bias"],
                                  p[pre + "0.imag_conv.weight"], p[pre + "0.imag_conv.bias"])
        s1 = "P" if j == 0 else f"zd{j - 1}"
        a_cl = cl(a)
        if j > 0:
            a_cl = np.concatenate([np.full((B, 1, f_in, c1), 7.0), a_cl], 1)
        bufs = {s1: a_cl, f"z{5 - j}": cl(skip)}
        shape = (B, T + 1, 2 * f_in, co)
        part, got = np.zeros(shape), np.zeros(shape)
        for par in (0, 1):
            fs, fm = st.specs[f"dec{j}.fs{par}"], st.specs[f"dec{j}.fm{par}"]
            assert fs.kind == fm.kind == "fwd_only" and fs.dw_off is None and fm.dw_off is None and fs.bias_pairs is None
            assert fm.res == f"pd{j}" and fm.stats_of == st.specs[f"dec{j}.fwd{par}"].stats_of
            part += scatter(run_spec(st, flat, fs, bufs, T, B), f"pd{j}", shape)
            got += scatter(run_spec(st, flat, fm, bufs, T, B), f"yd{j}", shape)
        assert np.abs(part + got - cl(full)).max() < 1e-5, j
        assert np.abs(part).max() > 0.1 and np.abs(got).max() > 0.1


def test_unpack_table_folds_block_gradients(ctx):
    """d(packed W) -> d(Wr), d(Wi): feed the packed-gradient buffer with the analytic block gradient of a linear probe."""
    st, flat = ctx["st"], ctx["flat"]
    spec = st.specs["enc3.fwd"]
    rng = np.random.default_rng(0)
    gpack = np.zeros(st.n_gpack)
    gw = rng.standard_normal((spec.Npad, spec.K))
    gpack[spec.dw_off:spec.dw_off + gw.size] = gw.reshape(-1)
    grads = np.zeros(st.layout.n_params)
    for q in range(4):
        e = st.utab[:, q].astype(np.int64)
        ok = e >= 0
        grads[ok] += np.where((e[ok] & 1) == 1, -1.0, 1.0) * gpack[e[ok] >> 1]
    # reference: d/dparam of sum(gw * Wpacked(param))
    ent = plan_entry(spec).astype(np.int64).reshape(spec.Npad, spec.K)
    ref = np.zeros_like(grads)
    ok = ent >= 0
    np.add.at(ref, ent[ok] >> 1, np.where((ent[ok] & 1) == 1, -1.0, 1.0) * gw[ok])
    assert np.abs(grads - ref).max() < 1e-12
    assert np.abs(ref).sum() > 0


def test_module_schema_and_optimizer_state_on_cpu():
    from sehip.model import DCCRN
    from sehip import distrib, utils, SehipError
    m = DCCRN(**KW)
    sd = m.state_dict()
    ref = O.init_params(O.DCCRNConfig(**KW))
    keys = [k for k in sd if not k.startswith(("stft.", "istft."))]
    assert set(keys) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in keys)
    a, s_, w = O.stft_bases(400, 512)
    assert torch.allclose(sd["stft.weight"][:, 0], a, atol=1e-6) and torch.allclose(sd["istft.weight"][:, 0], s_, atol=1e-6)
    # parameters are views of one flat buffer and survive load_state_dict / .to()
    m2 = DCCRN(**KW)
    m2.load_state_dict(sd)
    m2.to(torch.float32)
    assert torch.equal(m2.flat_params, m.flat_params)
    assert m2.encoder[0][0].real_conv.weight.data_ptr() == m2.flat_params.data_ptr()
    opt = distrib.get_optimizer(utils.dict2obj({"optim": "adam", "lr": 3e-4, "beta1": 0.9, "beta2": 0.999}), m2)
    st = opt.state_dict()
    assert len(st["state"]) == len(list(m2.parameters())) and set(st["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    with pytest.raises(SehipError):
        m2(torch.zeros(1, 1, 1200))  # CPU tensor: the HIP path has no fallback
    with pytest.raises(SehipError):
        distrib.get_model(utils.dict2obj({"name": "dcunet"}))


@pytest.mark.parametrize("case,extra", [("hamming", dict(win_type="hamming")), ("realbn", dict(use_cbn=False)), ("rnn1", dict(rnn_layers=1)),
                                        ("rnn3", dict(rnn_layers=3)), ("ru256", dict(rnn_units=256)), ("reallstm", dict(use_clstm=False))])
def test_constructor_variants_keep_the_reference_parameter_order(case, extra):
    """The module tree of every constructor variant against what the IMPORTED reference registered (tests/golden/dccrn_variants.npz keeps
    its named_parameters() order and every gradient's shape): same names, same order, same shapes -- optimizer state indices and
    checkpoints interchange.  CPU: the modules are built without a GPU."""
    from sehip.model import DCCRN
    from util import load_golden, json_entry
    g = {k[len(case) + 1:]: v for k, v in load_golden("dccrn_variants.npz").items() if k.startswith(case + "/")}
    names = json_entry(g, "grad_names_json")
    m = DCCRN(kernel_num=[16, 16, 32, 32, 64, 64], length=4000, **extra)
    mine = [(n, tuple(p.shape)) for n, p in m.named_parameters()]
    assert [n for n, _ in mine] == list(names)
    for n, shape in mine:
        assert tuple(g["grad16/" + n].shape) == shape, n
    for k in g:
        if k.startswith("state_after/"):
            assert k[len("state_after/"):] in m.state_dict(), k


def test_fused_statistics_column_tiles():
    """The forward products that also accumulate the ComplexBatchNorm sums (sehip_gemm_desc.stats): every 128-column tile
    must hold [64 re | 64 im] of the same 64 complex channels, and the re-ordered product is still the reference's conv."""
    from sehip import plan
    kw = dict(kernel_num=[16, 32, 64, 128, 256, 256], rnn_units=128, length=1200)
    st = plan.DCCRNStatic(plan.DCCRNConfig(**kw))
    # (round 3: conv_gemm_v3 also takes the sums of the two 64-output layers -- a single [32 re | 32 im] tile, no re-ordering)
    assert st.fused_stats == {"encoder.2.", "encoder.3.", "encoder.4.", "encoder.5.", "decoder.0.", "decoder.1.", "decoder.2."}
    for name in ("enc2.fwd", "dec2.fwd0", "dec2.fwd1"):
        sp = st.specs[name]
        assert sp.stats_of is not None and sp.N == sp.Npad == 64 and sp.v3_channels() is not None
        chan = np.concatenate([sp.ntab[q, 1] + np.arange(4) for q in range(16)])
        assert (chan == np.arange(64)).all()                    # natural column order: [32 re | 32 im]
    for name, co in (("enc3.fwd", 128), ("enc4.fwd", 256), ("enc5.fwd", 256), ("dec0.fwd0", 256), ("dec0.fwd1", 256), ("dec1.fwd1", 128)):
        sp = st.specs[name]
        assert sp.stats_of is not None and sp.N == co
        cr = co // 2
        chan = np.concatenate([sp.ntab[q, 1] + np.arange(4) for q in range(co // 4)])   # destination channel of every column
        for t in range(co // 128):
            tile = chan[128 * t:128 * t + 128]
            assert (tile[:64] == 64 * t + np.arange(64)).all() and (tile[64:] == cr + 64 * t + np.arange(64)).all(), (name, t)
    # numerics of a permuted product (enc4: 128 -> 256 channels) through the table interpreter
    p = O.init_params(O.DCCRNConfig(**kw), seed=3)
    flat = np.zeros(st.layout.n_params, dtype=np.float64)
    for nm in st.layout.param_names:
        off, _ = st.layout.param_off[nm]
        flat[off:off + p[nm].numel()] = p[nm].reshape(-1).double().numpy()
    g = torch.Generator().manual_seed(4)
    B, T, i = 1, 3, 4
    ci, co, fi = 128, 256, 256 >> i
    x = torch.randn(B, ci, fi, T, generator=g)
    pre = f"encoder.{i}."
    ref = O.complex_conv2d(x, p[pre + "0.real_conv.weight"], p[pre + "0.real_conv.bias"], p[pre + "0.imag_conv.weight"],
                           p[pre + "0.imag_conv.bias"])
    res = run_spec(st, flat, st.specs["enc4.fwd"], {"z3": cl(x)}, T, B)
    got = scatter(res, "y4", (B, T, fi // 2, co))
    assert np.abs(got - cl(ref)).max() < 1e-4
