"""Host-side plan of the Demucs path (no GPU): parameter order against the oracle's (which is pinned to the reference's
named_parameters() by tests/golden/demucs_tiny.npz), packing / un-packing table invariants, the restated resampling kernels,
length bookkeeping and the error behaviour for options that are not built."""
import numpy as np
import pytest
import torch

from oracle import demucs_oracle as DM

SMALL = dict(sources=["a", "b"], audio_channels=2, channels=32, depth=4, norm_starts=2, dconv_lstm=2, dconv_attn=2)


@pytest.fixture(scope="module")
def static():
    from sehip import plan_demucs as P
    cfg = P.DemucsConfig(**SMALL)
    return P, cfg, P.DemucsStatic(cfg)


def test_parameter_order_and_lengths(static):
    P, cfg, st = static
    ocfg = DM.DemucsConfig(**SMALL)
    assert [(n, s) for n, s, _ in cfg.param_specs()] == DM.param_shapes(ocfg)
    full = P.DemucsConfig(sources=["clean"], audio_channels=2)
    assert sum(int(np.prod(s)) for _, s, _ in full.param_specs()) == 133749986         # 133.7 M parameters
    for n in (1, 999, 6000, 96000):
        assert cfg.valid_length(n) == ocfg.valid_length(n)
    assert full.valid_length(96000) == DM.DemucsConfig(sources=["clean"]).valid_length(96000) == 96938


def test_every_parameter_is_packed_and_unpacked_once(static):
    P, cfg, st = static
    L = st.layout
    used = np.zeros(L.n_params, dtype=bool)
    for name in L.param_names:
        off, shape = L.param_off[name]
        used[off:off + int(np.prod(shape))] = True
    entries = st.unpack_entries
    assert (entries[used] >= 1).all() and (entries[~used] == 0).all()
    # weights and GroupNorm / LayerScale terms have one packed-gradient entry, the bias of a transposed convolution four (one per
    # output phase), everything else one
    four = np.zeros(L.n_params, dtype=bool)
    for name in L.param_names:
        if name.startswith("decoder.") and name.endswith(".3.bias"):
            off, shape = L.param_off[name]
            four[off:off + int(np.prod(shape))] = True
    assert (entries[four] == 4).all() and (entries[used & ~four] == 1).all()
    # compact tables: one entry per parameter + an index list for the four-entry ones
    assert st.utab1.shape == (L.n_params,) and (st.utab1[used] >= 0).all() and (st.utab1[~used] == -1).all()
    assert np.array_equal(st.ulist, np.flatnonzero(four)) and st.utab4.shape == (int(four.sum()), 4) and (st.utab4 >= 0).all()
    assert np.array_equal(st.utab4[:, 0], st.utab1[st.ulist])
    g = np.concatenate([st.utab1[st.utab1 >= 0], st.utab4[:, 1:].reshape(-1)]) >> 1
    assert g.max() < st.n_gpack
    # every weight element appears in exactly one forward product's packed operand
    fwd = np.zeros(L.n_params, dtype=np.int32)
    for p in st.prods.values():
        if p.kind == "fwd":
            w = st.wtab[p.w_off:p.w_off + p.Npad * p.K]
            np.add.at(fwd, w[w >= 0] >> 1, 1)
    for name in L.param_names:
        off, shape = L.param_off[name]
        if len(shape) >= 2:
            assert (fwd[off:off + int(np.prod(shape))] == 1).all(), name


def test_pack_run_descriptors_reproduce_the_index_table(static):
    """sehip_pack_bf16_runs' (base, stride) pairs / side entries decode to exactly the per-element table."""
    P, cfg, st = static
    n = st.n_wpack_dev
    want = st.wtab[:n].reshape(-1, 8)
    runs, side, dst = st.runs, st.side.reshape(-1, 8), st.pack_dst
    dec = np.full_like(want, -1)                              # entry i of the (sorted) table ...
    aff = runs[:, 0] >= 0
    dec[aff] = (runs[aff, :1].astype(np.int64) + np.arange(8)[None] * runs[aff, 1:].astype(np.int64)) << 1
    irr = runs[:, 0] <= -2
    dec[irr] = side[-2 - runs[irr, 0]]
    got = np.full_like(want, -1)
    got[dst] = dec                                            # ... produces run dst[i] (sehip_pack_bf16_runs_to)
    assert np.array_equal(got, want)
    assert np.array_equal(np.sort(dst), np.arange(dst.size))  # a permutation ...
    for a, b in zip([0, st.n_wpack_head // 8, st.n_wpack_fwd // 8, st.n_wpack_bwd_head // 8],
                    [st.n_wpack_head // 8, st.n_wpack_fwd // 8, st.n_wpack_bwd_head // 8, n // 8]):
        assert b == a or (dst[a:b].min() == a and dst[a:b].max() == b - 1)   # ... inside every separately packed segment
        key = np.where(runs[a:b, 0] >= 0, runs[a:b, 0].astype(np.int64), 1 << 40)
        assert (np.diff(key) >= 0).all()                      # sorted by base address: neighbouring lanes, neighbouring parameters
    assert aff.mean() > 0.9                                   # the layouts are long regular runs


def test_quad_view_weight_layouts(static):
    """The strided convolution as a two-tap convolution over frames of four samples, and its mirror images: the packed operands
    reproduce F.conv1d / F.conv_transpose1d on random data."""
    P, cfg, st = static
    L = st.layout
    rng = np.random.default_rng(0)
    flat = rng.standard_normal(L.n_params).astype(np.float32)
    get = lambda n: torch.from_numpy(flat[L.param_off[n][0]:L.param_off[n][0] + int(np.prod(L.param_off[n][1]))].reshape(L.param_off[n][1]))

    def operand(p):
        w = st.wtab[p.w_off:p.w_off + p.Npad * p.K].reshape(p.Npad, p.K)
        return torch.from_numpy(np.where(w >= 0, flat[np.maximum(w, 0) >> 1], 0.0).astype(np.float32))[:p.N]

    i, cin, ch, T = 1, 32, 64, 9
    x = torch.from_numpy(rng.standard_normal((1, cin, 4 * T + 4)).astype(np.float32))
    want = torch.nn.functional.conv1d(x, get("encoder.1.0.weight"), stride=4)                      # [1, ch, T]
    quad = x[0].t().reshape(T + 1, 4 * cin)                                                         # frames of four samples
    a = torch.cat([quad[:-1], quad[1:]], dim=1)                                                     # rows i: frames i, i+1
    got = a @ operand(st.prods["e1.conv"]).t()
    assert torch.allclose(got.t(), want[0], atol=1e-4)
    # input gradient: N = 4 cin columns = the quad view of dx
    dy = torch.from_numpy(rng.standard_normal((1, ch, T)).astype(np.float32))
    want = torch.nn.functional.conv_transpose1d(dy, get("encoder.1.0.weight"), stride=4)            # [1, cin, 4T+4]
    dyp = torch.cat([dy[0].t(), torch.zeros(1, ch)])                                                 # frame T is zero
    prev = torch.cat([torch.zeros(1, ch), dyp[:-1]])
    got = torch.cat([dyp, prev], dim=1) @ operand(st.prods["e1.conv.dg"]).t()                       # [T+1, 4 cin]
    assert torch.allclose(got.reshape(4 * T + 4, cin).t(), want[0], atol=1e-4)
    # transposed convolution of decoder index 1 (ch 64 -> 32)
    gq = torch.from_numpy(rng.standard_normal((1, ch, T)).astype(np.float32))
    wt = get("decoder.2.3.weight")
    want = torch.nn.functional.conv_transpose1d(gq, wt, stride=4)
    gp = torch.cat([gq[0].t(), torch.zeros(1, ch)])
    prev = torch.cat([torch.zeros(1, ch), gp[:-1]])
    got = torch.cat([gp, prev], dim=1) @ operand(st.prods["d1.ct"]).t()
    assert torch.allclose(got.reshape(4 * T + 4, cin).t(), want[0], atol=1e-4)
    dyt = torch.from_numpy(rng.standard_normal((1, cin, 4 * T + 4)).astype(np.float32))
    want = torch.nn.functional.conv1d(dyt, wt, stride=4)                                            # [1, ch, T]
    quad = dyt[0].t().reshape(T + 1, 4 * cin)
    got = torch.cat([quad[:-1], quad[1:]], dim=1) @ operand(st.prods["d1.ct.dg"]).t()
    assert torch.allclose(got.t(), want[0], atol=1e-4)


def test_resampling_kernels_match_the_oracle(static):
    P, _, _ = static
    for a, b in ((1, 2), (2, 1)):
        k, w = P.resample_kernels(a, b)
        ko, wo, _, _ = DM.resample_kernels(a, b)
        assert w == wo and np.allclose(k, ko.numpy(), atol=2e-7)


def test_unsupported_options_fail_loudly():
    from sehip import plan_demucs as P
    from sehip._lib import SehipError
    for bad in (dict(rewrite=False), dict(lstm_layers=2), dict(kernel_size=4), dict(context=3), dict(glu=False), dict(dconv_mode=3),
                dict(channels=20), dict(channels=32, dconv_lstm=0)):
        with pytest.raises(SehipError):
            P.DemucsConfig(**dict(SMALL, **bad))
