"""GPU parity of the Demucs HIP path (BASELINE config C3) against the CPU oracle (oracle/demucs_oracle.py,
pinned to vectors of the imported reference by tests/test_oracle_golden.py::test_demucs_oracle_matches_reference): the streaming
kernels, the LSTM layer and the LocalState attention op-locally through the C ABI, whole chains (every encoder / decoder output,
the separated sources, every parameter gradient) on 4-layer models with and without the resampler / stereo / several sources, the
full-width default network (channels 64, depth 6, 133.7 M parameters), and Solver steps at the C3 shape [16, 2, 96000]."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import demucs_oracle as DM
from oracle import dccrn_oracle as O
from util import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
SMALL = dict(sources=["a", "b"], audio_channels=2, channels=32, depth=4, norm_starts=2, dconv_lstm=2, dconv_attn=2)


def _lib():
    from sehip import _lib
    return _lib


def cl(x):
    """oracle [B, C, T] -> channels-last [B, T, C]"""
    return x.detach().transpose(1, 2).contiguous()


def randomise_small_terms(model, g):
    """non-trivial GroupNorm affine terms and layer scales (1e-4 at init would hide the DConv branches)"""
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith(".scale"):
                prm.copy_(0.3 + 0.1 * torch.randn(prm.shape, generator=g))
            elif prm.dim() == 1 and "lstm" not in name and name.endswith("weight"):
                prm.copy_(1 + 0.2 * torch.randn(prm.shape, generator=g))


def run_chain(kw, B, T, seed):
    from sehip.model import Demucs
    torch.manual_seed(seed)
    model = Demucs(**kw)
    g = torch.Generator().manual_seed(seed + 1)
    randomise_small_terms(model, g)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    mix = 0.3 * torch.randn(B, kw["audio_channels"], T, generator=g) + 0.05
    cfg = DM.DemucsConfig(**kw)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    taps = {}
    ref = DM.demucs_forward(leaves, mix, cfg, taps=taps)
    est = model(mix.cuda())
    ws = model.workspace(B, T)
    D = cfg.depth
    fwd = {}
    for i in range(D):
        fwd[f"enc{i}"] = rel_err(ws.bufs[f"e{i}.out"].t.float().cpu()[:, :, 0], cl(taps[f"enc{i}"]))
    for j in range(D - 1):
        i = D - 1 - j        # the oracle's dec{j} is decoder j's output; the HIP path stores it with the skip connection added
        fwd[f"dec{j}"] = rel_err(ws.bufs[f"d{i - 1}.in"].t.float().cpu()[:, :, 0], cl(taps[f"dec{j}"]) + cl(taps[f"enc{i - 1}"]))
    fwd["est"] = rel_err(est.detach().cpu(), ref.detach())
    ws.check_lstm_handoffs()
    G = torch.randn(ref.shape, generator=g) / ref.numel() ** 0.5
    names = sorted(leaves)
    grads = torch.autograd.grad((ref * G).sum(), [leaves[k] for k in names])
    est.backward(G.cuda())
    got = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
    num = den = 0.0
    rows = []
    for k, gr in zip(names, grads):
        e, n = float((got[k].double() - gr.double()).norm()), float(gr.double().norm())
        num += e * e; den += n * n
        rows.append((e, n, k))
    ws.check_lstm_handoffs()
    return dict(model=model, ws=ws, fwd=fwd, glob=(num / den) ** 0.5, gnorm=den ** 0.5, rows=rows, shape=tuple(est.shape), p=p, mix=mix, ref=ref.detach())


def check_chain(r, what, fwd_tol=2e-2, est_tol=1e-2, glob_tol=5e-2, big_tol=0.12):
    print(f"Demucs {what}: forward {', '.join(f'{k} {v:.2e}' for k, v in r['fwd'].items())}; global grad rel {r['glob']:.3e} (|g| = {r['gnorm']:.3f})")
    for k, v in r["fwd"].items():
        assert v < (est_tol if k == "est" else fwd_tol), (k, v)
    assert r["glob"] < glob_tol
    for e, n, k in r["rows"]:
        if n > 0.03 * r["gnorm"]:          # (the key bias of LocalState has an exactly-zero gradient: softmax over the keys)
            assert e < big_tol * n, (k, e, n)


def test_chain_stereo_two_sources_resampled():
    """4 layers (32..256 channels), GroupNorm(4) / BLSTM / LocalState from layer 2, x2 resampling, [2, 2, 6000]: T = 3028, 756, 188, 46."""
    r = run_chain(dict(SMALL), 2, 6000, 3)
    assert r["shape"] == (2, 2, 2, 6000)
    check_chain(r, "stereo, 2 sources, resampled")


def test_chain_mono_one_source_plain():
    """mono, one source (the padded input / output channels), no resampler, no normalisation, batch 3, an odd clip length."""
    kw = dict(SMALL, sources=["a"], audio_channels=1, resample=False, normalize=False)
    r = run_chain(kw, 3, 5001, 5)
    assert r["shape"] == (3, 1, 1, 5001)
    check_chain(r, "mono, 1 source, no resampler")


def test_chain_attention_without_lstm_and_deeper_dconv():
    """LocalState from layer 1 but BLSTM only from layer 3, three DConv branches per layer (dilations 1, 2, 4), GroupNorm(2)."""
    kw = dict(SMALL, channels=64, dconv_attn=1, dconv_lstm=3, dconv_depth=3, norm_groups=2, norm_starts=1, resample=False)
    r = run_chain(kw, 2, 4000, 7)
    check_chain(r, "attention from layer 1, LSTM from layer 3, DConv depth 3", est_tol=2e-2)    # (no normalisation: est is the raw last layer)


def test_chain_with_chunked_blstm():
    """[2, 2, 16000]: 500 frames at the first BLSTM layer (> max_steps = 200): the LSTM runs on 2 x 5 overlapping chunks of 200 frames
    (src/model/demucs.py:91-117), 124 frames (one chunk) at the second."""
    r = run_chain(dict(SMALL), 2, 16000, 13)
    assert r["ws"].chunks[2] == (5, 200, 100) and r["ws"].chunks[3][0] == 1 and r["ws"].bufs["e2.d0.pre0"].t.shape[0] == 10
    check_chain(r, "chunked BLSTM (500 frames)")


@pytest.mark.parametrize("T,C", [(201, 32), (500, 8), (437, 64), (150, 16)])
def test_chunk_gather_pick_and_adjoints(T, C):
    """sehip_dmx_frames against the oracle's unfold / middle-part concatenation (src/model/demucs.py:17-32, :104-117) and their adjoints."""
    import math
    L = _lib()
    g = torch.Generator().manual_seed(T)
    B = 2
    nf, W, S = (math.ceil(T / 100), 200, 100) if T > 200 else (1, T, T)
    x = torch.randn(B, T, C, generator=g).to(BF)
    skip = torch.randn(B, T, C, generator=g).to(BF)
    xl = x.float().transpose(1, 2).requires_grad_(True)                                      # [B, C, T]
    if nf > 1:
        fr = DM.unfold(xl, W, S)                                                               # [B, C, nf, W]
        fr_ref = fr.permute(0, 2, 3, 1).reshape(B * nf, W, C)
    else:
        fr_ref = xl.transpose(1, 2)
    dev = "cuda"
    xd, skd = x.to(dev), skip.to(dev)           # (named: a temporary's storage would be re-used by the next allocation)
    frames = torch.zeros(B * nf, W, C, dtype=BF, device=dev)
    L.call("sehip_dmx_frames", 0, xd.data_ptr(), None, B, T, C, nf, W, S, frames.data_ptr(), None)
    assert torch.equal(frames.float().cpu(), fr_ref.detach())
    y = torch.randn(B * nf, W, C, generator=g).to(BF)
    yl = y.float().requires_grad_(True)
    if nf > 1:
        f4 = yl.reshape(B, nf, W, C).permute(0, 1, 3, 2)                                       # [B, nf, C, W]
        lim = S // 2
        parts = [f4[:, k, :, :-lim] if k == 0 else f4[:, k, :, lim:] if k == nf - 1 else f4[:, k, :, lim:-lim] for k in range(nf)]
        picked = torch.cat(parts, -1)[..., :T].transpose(1, 2)
    else:
        picked = yl
    want = picked + skip.float()
    out = torch.zeros(B, T, C, dtype=BF, device=dev)
    yd = y.to(dev)
    L.call("sehip_dmx_frames", 1, yd.data_ptr(), skd.data_ptr(), B, T, C, nf, W, S, out.data_ptr(), None)
    assert rel_err(out.float().cpu(), want.detach()) < 4e-3
    d = torch.randn(B, T, C, generator=g).to(BF)
    dy_ref, = torch.autograd.grad((picked * d.float()).sum(), [yl])
    sel = torch.zeros(B * nf, W, C, dtype=BF, device=dev)
    dd = d.to(dev)
    L.call("sehip_dmx_frames", 2, dd.data_ptr(), None, B, T, C, nf, W, S, sel.data_ptr(), None)
    assert torch.equal(sel.float().cpu(), dy_ref)
    dfr = torch.randn(B * nf, W, C, generator=g).to(BF)
    dx_ref, = torch.autograd.grad((fr_ref * dfr.float()).sum(), [xl])
    got = torch.zeros(B, T, C, dtype=BF, device=dev)
    dfd = dfr.to(dev)
    L.call("sehip_dmx_frames", 3, dfd.data_ptr(), skd.data_ptr(), B, T, C, nf, W, S, got.data_ptr(), None)
    torch.cuda.synchronize()
    assert rel_err(got.float().cpu(), dx_ref.transpose(1, 2) + skip.float()) < 4e-3


def test_full_width_default_network():
    """The C3 network (channels 64, depth 6, every default; 133.7 M parameters) on [1, 2, 24000] against the oracle: forward taps,
    output, and every parameter gradient under a fixed upstream gradient."""
    r = run_chain(dict(sources=["clean"], audio_channels=2), 1, 24000, 11)
    assert sum(v.numel() for v in r["p"].values()) == 133749986
    check_chain(r, "full width (64 channels, depth 6)", fwd_tol=3e-2, est_tol=1.5e-2, glob_tol=6e-2, big_tol=0.15)


@pytest.mark.parametrize("kw,B,T", [(dict(sources=["clean"], audio_channels=2), 2, 24000),
                                     (dict(sources=["a", "b"], audio_channels=1, channels=32, depth=4), 3, 9000)])
def test_streaming_weight_gradients_match_the_generic_kernel(kw, B, T):
    """csrc/dtw.hip on Demucs' products (dense rows at frame offsets, quad views, chunked LSTM frames, padded K / N, short and long
    row spaces): every weight gradient the streaming kernel takes, against the table-gathered generic kernel on the same operands."""
    import ctypes as C
    from sehip.model import Demucs
    from sehip._lib import call, stream
    torch.manual_seed(3)
    model = Demucs(**kw).cuda().train()
    x = 0.3 * torch.randn(B, kw["audio_channels"], T, device="cuda")
    y = model(x)
    y.backward(torch.randn_like(y) / y.numel() ** 0.5)
    torch.cuda.synchronize()
    ws = model.workspace(B, T)
    assert len(ws._dtw) >= 0.4 * sum(1 for k in ws.desc if k.endswith(".wg")), (len(ws._dtw), "products on the streaming kernel")
    worst = 0.0
    for name in sorted(ws._dtw):
        d = ws.desc[name]
        p = ws.st.prods[name[:-3]]
        n = p.Npad * p.K
        lo = p.dw_off
        view = ws.gpack[lo:lo + n]
        bview = ws.gpack[p.db_off:p.db_off + p.Npad] if p.db_off is not None else None
        res = []
        for dense in (False, True):
            view.zero_()
            if bview is not None:
                bview.zero_()
            if dense:
                ws._launch_wgrad(name[:-3], stream())
            else:
                call("sehip_wgrad", C.byref(d), stream())
            torch.cuda.synchronize()
            res.append((view.view(p.Npad, p.K)[:p.N].clone(), bview[:p.N].clone() if bview is not None else None))
        (w0, b0), (w1, b1) = res
        e = rel_err(w1.cpu(), w0.cpu()) if float(w0.abs().max()) > 0 else float(w1.abs().max())
        worst = max(worst, e)
        assert e < 2e-3, (name, e)
        if b0 is not None and float(b0.abs().max()) > 0:
            assert rel_err(b1.cpu(), b0.cpu()) < 2e-3, name
    print(f"{len(ws._dtw)} streaming weight gradients, worst relative difference to the generic kernel {worst:.2e}")


# ---------------------------------------------------------------------------------------------------------------------------
# op-local
# ---------------------------------------------------------------------------------------------------------------------------
def _ptr(t):
    return None if t is None else t.data_ptr()


@pytest.mark.parametrize("mode,G,scaled,C", [(0, 0, False, 64), (0, 4, False, 64), (1, 0, False, 64), (1, 1, True, 48), (1, 4, False, 128), (0, 1, False, 8),
                                             (1, 1, True, 4096)])
def test_groupnorm_activation_family(mode, G, scaled, C):
    """GroupNorm(G) (or none) + GELU / GLU (+ LayerScale + residual, + addend), forward and backward, against torch on the same
    bf16-rounded input."""
    L = _lib()
    g = torch.Generator().manual_seed(C + 10 * mode + G)
    B, T = 3, 157 if C < 4096 else 23
    Co = C // 2 if mode else C
    y = (torch.randn(B, T, C, generator=g) * 1.5 + 0.3).to(BF)
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    scale = 0.3 + 0.2 * torch.randn(Co, generator=g) if scaled else None
    resid = torch.randn(B, T, Co, generator=g).to(BF) if scaled else None
    add = torch.randn(B, T, Co, generator=g).to(BF)
    dz = torch.randn(B, T, Co, generator=g).to(BF)
    yf = y.float().transpose(1, 2).requires_grad_(True)                       # [B, C, T]
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    sc = scale.clone().requires_grad_(True) if scaled else None
    n = F.group_norm(yf, G, gm, bt, eps=1e-5) if G else yf
    v = F.glu(n, dim=1) if mode else F.gelu(n)
    if scaled:
        v = resid.float().transpose(1, 2) + sc[:, None] * v
    v = v + add.float().transpose(1, 2)
    leaves = [yf] + ([gm, bt] if G else []) + ([sc] if scaled else [])
    refs = torch.autograd.grad((v * dz.float().transpose(1, 2)).sum(), leaves)
    dev = "cuda"
    yd, dzd = y.to(dev), dz.to(dev)
    stats = torch.zeros(B, 8, 2, dtype=torch.float64, device=dev)
    sums = torch.zeros(B, 8, 2, dtype=torch.float64, device=dev)
    out = torch.zeros(B, T, Co, dtype=BF, device=dev)
    dy = torch.zeros(B, T, C, dtype=BF, device=dev)
    gch = torch.zeros(2 * C + Co, device=dev)
    gmd, btd = gamma.to(dev), beta.to(dev)
    scd, rd, ad = (scale.to(dev) if scaled else None), (resid.to(dev) if scaled else None), add.to(dev)
    st = None
    if G:
        L.call("sehip_dmx_gn_stats", yd.data_ptr(), B, T, C, G, stats.data_ptr(), None)
    L.call("sehip_dmx_act_fwd", yd.data_ptr(), _ptr(stats) if G else None, _ptr(gmd) if G else None, _ptr(btd) if G else None, max(G, 1), 1e-5, mode,
           _ptr(scd), _ptr(rd), _ptr(ad), B, T, C, out.data_ptr(), None)
    L.call("sehip_dmx_act_bwd", dzd.data_ptr(), yd.data_ptr(), _ptr(stats) if G else None, _ptr(gmd) if G else None, _ptr(btd) if G else None, max(G, 1),
           1e-5, mode, _ptr(scd), B, T, C, _ptr(sums) if G else None, _ptr(gch) if G else None, dy.data_ptr(), None)
    torch.cuda.synchronize()
    assert rel_err(out.float().cpu(), v.detach().transpose(1, 2)) < 4e-3
    assert rel_err(dy.float().cpu(), refs[0].transpose(1, 2)) < 6e-3
    if G:
        assert rel_err(gch[:C].cpu(), refs[1]) < 5e-3 and rel_err(gch[C:2 * C].cpu(), refs[2]) < 5e-3
    if scaled:
        assert rel_err(gch[2 * C:].cpu(), refs[-1]) < 5e-3


@pytest.mark.parametrize("persistent", [True, False])
@pytest.mark.parametrize("Bn,T,H", [(2, 37, 32), (16, 50, 64), (19, 12, 256), (16, 187, 512), (40, 200, 128), (5, 9, 96)])
def test_bidirectional_lstm_layer(Bn, T, H, persistent):
    """One bidirectional layer, forward and backward, against the oracle's recurrence (given the same pre-gates and bf16 weights): h,
    and the pre-activation gate gradients of both directions -- as ONE persistent launch whose workgroups hand h(t) / the gate
    gradients over once per step (its time-out word must stay clear; H = 96 has no persistent instantiation and falls back), and as
    one launch per time step."""
    L = _lib()
    g = torch.Generator().manual_seed(H + T)
    whh = (torch.randn(2, 4 * H, H, generator=g) / H ** 0.5).to(BF)
    pre = torch.randn(Bn, T, 2, 4 * H, generator=g)
    dh = torch.randn(Bn, T, 2 * H, generator=g).to(BF)
    pre_l = pre.clone().requires_grad_(True)
    outs = []
    for d in range(2):
        h = torch.zeros(Bn, H); c = torch.zeros(Bn, H)
        seq = [None] * T
        for t in (range(T - 1, -1, -1) if d else range(T)):
            hb = h.to(BF).float()                                                   # the HIP path feeds h back in bf16
            i, f, gg, o = (pre_l[:, t, d] + hb @ whh[d].float().t()).chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            seq[t] = h
        outs.append(torch.stack(seq, 1))
    hs_ref = torch.cat(outs, dim=2)                                                # [Bn, T, 2H]
    dpre_ref, = torch.autograd.grad((hs_ref * dh.float()).sum(), [pre_l])
    dev = "cuda"
    pre_d = pre.to(dev).contiguous()
    hs = torch.zeros(Bn, T, 2 * H, dtype=BF, device=dev)
    cs = torch.zeros(Bn, T, 2 * H, device=dev)
    dG = torch.zeros(Bn, T, 2, 4 * H, dtype=BF, device=dev)
    dc = torch.zeros(2 * Bn * H, device=dev)
    whh_d, whhT_d = whh.to(dev).contiguous(), whh.transpose(1, 2).contiguous().to(dev)
    sync = torch.zeros(L.lib().sehip_dmx_lstm_sync_bytes() // 4, dtype=torch.int32, device=dev) if persistent else None
    L.call("sehip_dmx_lstm_fwd", pre_d.data_ptr(), whh_d.data_ptr(), Bn, T, H, hs.data_ptr(), cs.data_ptr(), _ptr(sync), None)
    torch.cuda.synchronize()
    assert sync is None or int(sync[60]) == 0, "a hand-off spin timed out"
    dh_d = dh.to(dev)
    L.call("sehip_dmx_lstm_bwd", pre_d.data_ptr(), whhT_d.data_ptr(), cs.data_ptr(), dh_d.data_ptr(), Bn, T, H, dG.data_ptr(), dc.data_ptr(), _ptr(sync), None)
    torch.cuda.synchronize()
    assert sync is None or int(sync[60]) == 0, "a hand-off spin timed out"
    assert rel_err(hs.float().cpu(), hs_ref.detach()) < 4e-3
    assert rel_err(dG.float().cpu(), dpre_ref) < 1.5e-2


@pytest.mark.parametrize("B,T,hid", [(2, 50, 32), (1, 187, 256), (3, 33, 64), (2, 47, 512), (1, 500, 64)])     # (the last: rows straight from memory, the score tiles fill the LDS)
def test_local_state_attention(B, T, hid):
    """The attention between LocalState's 1x1 convolutions against the oracle's local_state (identity convolutions feed it the same
    query / key / content / decay rows): output and the gradient of every row."""
    L = _lib()
    heads, nd = 4, 4
    g = torch.Generator().manual_seed(T + hid)
    nq = 3 * hid + heads * nd
    qkv = torch.randn(B, T, nq, generator=g).to(BF)
    dres = torch.randn(B, T, hid, generator=g).to(BF)
    x = qkv.float().requires_grad_(True)
    xc = x.transpose(1, 2)                                                         # [B, nq, T]
    q, k, content, raw = xc[:, :hid], xc[:, hid:2 * hid], xc[:, 2 * hid:3 * hid], xc[:, 3 * hid:]
    idx = torch.arange(T, dtype=torch.float32)
    delta = idx[:, None] - idx[None, :]
    qh, kh = q.reshape(B, heads, -1, T), k.reshape(B, heads, -1, T)
    dots = torch.einsum("bhct,bhcs->bhts", kh, qh) / kh.shape[2] ** 0.5
    dq = torch.sigmoid(raw.reshape(B, heads, -1, T)) / 2
    kern = -torch.arange(1, nd + 1, dtype=torch.float32).view(-1, 1, 1) * delta.abs() / nd ** 0.5
    dots = dots + torch.einsum("fts,bhfs->bhts", kern, dq)
    dots = dots.masked_fill(torch.eye(T, dtype=torch.bool), -100.0)
    w = torch.softmax(dots, dim=2)
    res = torch.einsum("bhts,bhct->bhcs", w, content.reshape(B, heads, -1, T)).reshape(B, -1, T)
    dref, = torch.autograd.grad((res * dres.float().transpose(1, 2)).sum(), [x])
    dev = "cuda"
    qd = qkv.to(dev)
    out = torch.zeros(B, T, hid, dtype=BF, device=dev)
    dqkv = torch.zeros(B, T, nq, dtype=BF, device=dev)
    slabs = torch.empty(L.lib().sehip_dmx_attn_bwd_scratch_floats(B, T, hid), device=dev)
    L.call("sehip_dmx_attn_fwd", qd.data_ptr(), B, T, hid, heads, nd, nq, out.data_ptr(), None)
    dres_d = dres.to(dev)
    L.call("sehip_dmx_attn_bwd", qd.data_ptr(), dres_d.data_ptr(), B, T, hid, heads, nd, nq, slabs.data_ptr(), dqkv.data_ptr(), None)
    torch.cuda.synchronize()
    assert rel_err(out.float().cpu(), res.detach().transpose(1, 2)) < 4e-3
    got = dqkv.float().cpu()
    for name, a, b in (("query", 0, hid), ("key", hid, 2 * hid), ("content", 2 * hid, 3 * hid), ("decay", 3 * hid, nq)):
        assert rel_err(got[..., a:b], dref[..., a:b]) < 6e-3, name      # (bf16 output)


@pytest.mark.parametrize("ac,S,resample,normalize,T", [(2, 2, True, True, 5000), (1, 1, True, False, 3001), (2, 1, False, True, 4000)])
def test_prep_and_post(ac, S, resample, normalize, T):
    """normalise + pad + x2 up-sampling, and /2 down-sampling + de-normalise + center_trim with its gradient, against the oracle's
    restated julius resampler."""
    from sehip import plan_demucs as P
    L = _lib()
    cfg = P.DemucsConfig(sources=["s"] * S, audio_channels=ac, channels=32, depth=4, resample=resample, normalize=normalize)
    g = torch.Generator().manual_seed(T)
    B = 2
    mix = 0.3 * torch.randn(B, ac, T, generator=g) + 0.1
    Tv = cfg.valid_length(T)
    padl = (Tv - T) // 2
    Tin = 2 * Tv if resample else Tv
    x = mix
    mean, std = 0.0, 1.0
    if normalize:
        mono = mix.mean(1, keepdim=True)
        mean, std = mono.mean(-1, keepdim=True), mono.std(-1, keepdim=True)
        x = (x - mean) / (1e-5 + std)
    x = F.pad(x, (padl, Tv - T - padl))
    if resample:
        x = DM.resample_frac(x, 1, 2)
    dev = "cuda"
    ms = torch.zeros(B, 2, device=dev)
    xb = torch.zeros(B, Tin, cfg.acp, dtype=BF, device=dev)
    kup = kdn = None
    wup = wdn = klu = kld = 0
    if resample:
        ku, wup = P.resample_kernels(1, 2)
        kd, wdn = P.resample_kernels(2, 1)
        kup, kdn = torch.from_numpy(ku.reshape(-1)).to(dev), torch.from_numpy(kd.reshape(-1)).to(dev)
        klu, kld = ku.shape[1], kd.shape[1]
    mix_d = mix.to(dev)
    acc = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    L.call("sehip_dmx_prep", mix_d.data_ptr(), B, ac, cfg.acp, T, padl, Tv, int(normalize), int(resample), _ptr(kup), wup, klu, acc.data_ptr(), ms.data_ptr(),
           xb.data_ptr(), None)
    torch.cuda.synchronize()
    assert rel_err(xb.float().cpu()[..., :ac], x.transpose(1, 2)) < 4e-3
    assert float(xb.float()[..., ac:].abs().max()) == 0 if cfg.acp > ac else True
    # post: a random network output y [B, Tin, cop] -> out [B, S*ac, T]
    co, cop = cfg.co, cfg.cop
    y = torch.randn(B, Tin, cop, generator=g)
    yl = y[..., :co].transpose(1, 2).clone().requires_grad_(True)              # [B, co, Tin]
    z = DM.resample_frac(yl, 2, 1) if resample else yl
    z = z * std + mean
    z = DM.center_trim(z, T)
    dout = torch.randn(B, co, T, generator=g)
    dy_ref, = torch.autograd.grad((z * dout).sum(), [yl])
    out = torch.zeros(B, co, T, device=dev)
    dy = torch.zeros(B, Tin, cop, dtype=BF, device=dev)
    y_d, dout_d = y.to(dev), dout.to(dev)
    L.call("sehip_dmx_post", y_d.data_ptr(), ms.data_ptr(), B, co, cop, Tin, padl, T, int(resample), _ptr(kdn), wdn, kld, out.data_ptr(), None)
    L.call("sehip_dmx_post_bwd", dout_d.data_ptr(), ms.data_ptr(), B, co, cop, Tin, padl, T, int(resample), _ptr(kdn), wdn, kld, dy.data_ptr(), None)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), z.detach()) < 1e-5
    assert rel_err(dy.float().cpu()[..., :co], dy_ref.transpose(1, 2)) < 4e-3


# ---------------------------------------------------------------------------------------------------------------------------
# API surface
# ---------------------------------------------------------------------------------------------------------------------------
def test_registry_state_dict_and_limits():
    from sehip import distrib
    from sehip._lib import SehipError
    from sehip.utils import dict2obj
    model = distrib.get_model(dict2obj(dict(SMALL, name="demucs")))
    cfg = DM.DemucsConfig(**SMALL)
    assert [k for k, _ in model.named_parameters()] == [n for n, _ in DM.param_shapes(cfg)]
    sd = {k: v + 1 for k, v in model.state_dict().items()}
    model.load_state_dict(sd)                                   # (the reference's key migration, src/model/demucs.py:492-501, finds nothing to move)
    assert all(torch.equal(v, sd[k]) for k, v in model.state_dict().items())
    assert model.valid_length(6000) == cfg.valid_length(6000)
    with pytest.raises(SehipError):
        model(torch.zeros(1, 2, 6000))                         # CPU tensor
    model = model.cuda()
    with pytest.raises(SehipError):
        model(torch.zeros(1, 2, 80000, device="cuda"))          # 2500 frames at the first LocalState: its score tile does not fit the LDS
    with pytest.raises(SehipError):
        distrib.get_model(dict2obj(dict(SMALL, name="demucs", dconv_mode=3)))


def c3_config(tmp):
    from sehip.utils import dict2obj
    return dict2obj({
        "seed": 10, "root": None, "ha": None,
        "model": {"name": "demucs", "audio_channels": 2, "num_spk": 1, "sources": ["clean"], "samplerate": 48000, "segment": 2},
        "optim": {"optim": "adam", "lr": 3e-4, "beta1": 0.9, "beta2": 0.999, "loss": "si-sdr", "clip_grad": 5, "pit": False, "load": False},
        "dset": {"name": "synthetic"},
        "solver": {"epochs": 1, "save_checkpoint_interval": 1000, "all_steps": True, "total_steps": 0, "patience": 0,
                   "root": str(tmp), "resume": None, "preloaded_model": None,
                   "validation": {"interval": 1000, "metric": "loss", "total_steps": 0}, "test": {"interval": 1000}},
    })


def test_c3_shape_solver_steps(tmp_path):
    """BASELINE config C3: 48 kHz stereo 2-s clips (96000 samples), batch 16, the default network: three Solver steps through the
    registry; the loss is finite and goes down, and the network output of the first 2 clips equals the oracle's."""
    from sehip.train import main
    from sehip.solver import ScalarLog
    g = torch.Generator().manual_seed(0)
    clean = 0.1 * torch.randn(16, 1, 2, 96000, generator=g)
    mix = clean[:, 0] + 0.05 * torch.randn(16, 2, 96000, generator=g)
    batches = [(mix, clean, [None], [None], ["x"], [0])] * 3
    log = ScalarLog()
    solver = main(c3_config(tmp_path), return_solver=True, device="gpu", train_dataloader=batches, validation_dataloader=[batches[0]], writer=log)
    p = {k: v.detach().cpu().clone() for k, v in solver.model.state_dict().items()}
    solver._run_one_epoch(0, 1, train=True)
    losses = [v for (t, v, _s) in log.scalars if t == "Train/Loss_step"]
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[2] < losses[0]
    solver.model.load_state_dict(p)
    with torch.no_grad():
        est = solver.model(mix[:2].cuda())
        ref = DM.demucs_forward(p, mix[:2], DM.DemucsConfig(sources=["clean"], audio_channels=2))
    assert tuple(est.shape) == (2, 1, 2, 96000) and rel_err(est.cpu(), ref) < 2e-2
    print("C3 losses", losses, "oracle loss on 2 clips", float(O.loss_sisdr(ref, clean[:2])))


def test_gradient_ranges_for_the_data_parallel_exchange():
    """The hook the Solver uses to start the all-reduce early (sehip/solver.py): the ranges tile the flat gradient buffer, arrive
    decoder first, and the gradients are the same as without the hook."""
    from sehip.model import Demucs
    torch.manual_seed(2)
    model = Demucs(**dict(SMALL, channels=64, depth=5, norm_starts=3, dconv_lstm=3, dconv_attn=3)).cuda().train()
    g = torch.Generator().manual_seed(3)
    mix = (0.3 * torch.randn(2, 2, 9000, generator=g)).cuda()
    G = torch.randn(2, 2, 2, 9000, generator=g).cuda() / 100
    def fresh():
        model.flat_grads.zero_()
        for _, prm in model._params:
            prm.grad = None
        model._grads_live = False

    model(mix).backward(G)
    torch.cuda.synchronize()
    want = model.flat_grads.clone()
    fresh()
    model(mix).backward(G)
    torch.cuda.synchronize()
    noise = rel_err(model.flat_grads.cpu(), want.cpu())     # run-to-run: the order of the fp32 / fp64 atomics flips bf16 roundings
    fresh()
    ranges = []

    def hook(lo, hi, st):
        st.synchronize()
        ranges.append((lo, hi, float(model.flat_grads[lo:hi].abs().sum())))
    model.grad_range_hook = hook
    model(mix).backward(G)
    torch.cuda.synchronize()
    model.grad_range_hook = None
    n = model.flat_grads.numel()
    assert len(ranges) >= 3 and ranges[0][1] == n and ranges[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(ranges, ranges[1:])) and all(r[2] > 0 for r in ranges)
    assert ranges[0][0] == model.static.layout.param_off["decoder.0.0.weight"][0]
    err = rel_err(model.flat_grads.cpu(), want.cpu())
    print(f"Demucs gradient ranges {[(lo, hi) for lo, hi, _ in ranges]}: difference to the run without hook {err:.2e}, run-to-run {noise:.2e}")
    # run-to-run differences of this model measure 2e-3 ... 9e-3 (eight repetitions), the difference with the hook 5e-3 ... 9e-3: one noise
    # sample is a poor yardstick on its own (3 x 2.1e-3 < 6.6e-3 failed once), so the floor is twice the largest noise seen.  A range
    # handed over before its last weight gradient has landed misses whole tensors: tens of percent
    assert err < max(3 * noise, 2e-2)


def test_bench_two_ranks_on_one_gpu_demucs():
    """The data-parallel path with the full C3 network (535 MB of gradients): two processes sharing cuda:0 over gloo (RCCL refuses
    two ranks on one device), gradient ranges handed to the all-reduce as the backward pass finishes them, 1/world folded into
    the optimizer launch."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SEHIP_DIST_BACKEND="gloo", SEHIP_LOCAL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--workload", "demucs", "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["value"] > 0
    assert out["final_loss"] == out["final_loss"]  # not NaN


def test_evaluate_with_the_hip_demucs():
    """The chunked inference path (src/evaluate.py:10-98 -> sehip.evaluate.evaluate) with the HIP Demucs, a multi-source WAV model:
    evaluate() == the model applied to each segment + the stitch rule; the eval-mode forward equals the oracle."""
    import types
    from sehip.evaluate import evaluate
    from sehip.model import Demucs
    torch.manual_seed(4)
    kw = dict(SMALL, sources=["a"])
    model = Demucs(**kw).cuda().eval()
    x = 0.1 * torch.randn(2, 2, 6000 + 2 * 512 + 9, generator=torch.Generator().manual_seed(1))
    m = types.SimpleNamespace(name="demucs", win_length=512, n_fft=512, hop_length=128, center=True, segment=0.375, sources=["a"])
    c = types.SimpleNamespace(model=m, dset=types.SimpleNamespace(norm="z-score", sample_rate=16000))
    y = evaluate(x, model, torch.device("cuda:0"), c)
    assert tuple(y.shape) == (2, 1, 2, x.shape[-1])
    mean, std = x.mean(-1, keepdim=True), x.std(-1, keepdim=True)
    xn = ((x - mean) / (std + 1e-9)).cuda()
    xp = torch.nn.functional.pad(xn, (0, 512 - 9))
    with torch.no_grad():
        segs = [model(xp[..., k * 512:k * 512 + 6000]) for k in range(4)]
        ref0 = DM.demucs_forward({k: v.detach().cpu() for k, v in model.state_dict().items()}, xp[..., :6000].cpu(), DM.DemucsConfig(**kw))
    assert rel_err(segs[0].cpu(), ref0) < 1e-2
    want = torch.cat([segs[0]] + [s[..., -512:] for s in segs[1:]], -1)[..., :x.shape[-1]]
    want = want * (std.unsqueeze(1).cuda() + 1e-9) + mean.unsqueeze(1).cuda()
    assert rel_err(y.cpu(), want.cpu()) < 2e-3     # (run-to-run: the GroupNorm sums are atomics, bf16 roundings flip)


def test_forced_handoff_timeout_skips_the_step_and_falls_back(tmp_path):
    """VERDICT r2 / ADVICE: a timed-out hand-off spin must never corrupt an optimizer step.  The spin limit is forced to 1 poll
    through the test word of the sync block (word 61), so the first persistent LSTM launch gives up and sets the sticky word:
    the optimizer step of that train step is a no-op ON THE DEVICE (parameters, Adam moments and the step counter unchanged),
    the Solver's next synchronisation point notices, the model falls back to one launch per time step, and the step after that
    updates the parameters exactly like a model that used the per-step launches from the start."""
    import copy
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    cfg = c3_config(tmp_path)
    for k, v in dict(SMALL, sources=["clean"]).items():
        setattr(cfg.model, k, v)
    g = torch.Generator().manual_seed(5)
    clean = 0.1 * torch.randn(2, 1, 2, 8000, generator=g)
    mix = clean[:, 0] + 0.05 * torch.randn(2, 2, 8000, generator=g)

    def make():
        torch.manual_seed(3)
        m = distrib.get_model(cfg.model)
        return Solver(copy.deepcopy(cfg), m, distrib.get_optimizer(cfg.optim, m), distrib.get_loss_function(cfg.optim), device="gpu",
                      writer=ScalarLog())
    s = make()
    assert s.model.static.lstms, "the test model must contain BLSTM layers"
    mx, sr = s._prepare_batch(mix, clean)
    p0 = s.model.flat_params.detach().clone()
    ws = s.model.workspace(2, 8000)
    ws.lstm_sync[61] = -1                                  # 0xffffffff: the first hand-off wait of every persistent launch times out
    s.train_step(mx, sr)
    torch.cuda.synchronize()
    assert int(ws.lstm_sync[60]) != 0, "the forced time-out did not fire"
    assert torch.equal(s.model.flat_params.detach(), p0), "an optimizer step was applied after a hand-off time-out"
    assert int(s.optimizer._step_dev.item()) == 0
    s._model_health()                                       # what the Solver does at its synchronisation points
    assert s.model.static.lstm_per_step and int(ws.lstm_sync[60]) == 0 and getattr(s, "lost_steps", 0) == 1
    loss, _ = s.train_step(mx, sr)                          # per-step launches now
    torch.cuda.synchronize()
    assert not torch.equal(s.model.flat_params.detach(), p0) and np.isfinite(float(loss))
    # reference: a fresh Solver on the per-step launches from the start
    r = make()
    r.model.static.lstm_per_step = True
    loss_r, _ = r.train_step(*r._prepare_batch(mix, clean))
    # (not bit-identical: the statistics / weight-gradient accumulators are fp32 / fp64 atomics whose order varies between launches)
    assert abs(float(loss) - float(loss_r)) < 1e-3 * max(1.0, abs(float(loss_r)))
    assert rel_err(s.model.flat_params.detach().cpu(), r.model.flat_params.detach().cpu()) < 1e-3      # (first Adam step = lr * sign(g): a few signs of ~0 gradients differ)


def test_handoff_timeout_on_one_rank_skips_the_step_on_every_rank(tmp_path):
    """the step guard is a per-rank device word, but the gradients are all-reduced -- a time-out on ONE rank must
    turn the optimizer step into a no-op on EVERY rank (Solver: MAX all-reduce of the word before the optimizer launch,
    distrib.allreduce_step_guard), every rank must take the per-step fall-back at the same health check, and the replicas must
    still hold identical parameters after the next step.  Two ranks share cuda:0 over gloo; rank 1 alone gets the forced time-out."""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SEHIP_DIST_BACKEND="gloo", SEHIP_LOCAL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dp_guard_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (torch.load(tmp_path / f"guard_r{k}.pt") for k in (0, 1))
    for o in (a, b):
        assert o["guard_after_step"] != 0, "the time-out word did not reach this rank"
        assert o["unchanged"] and o["step_dev"] == 0, "an optimizer step was applied from gradients that contain a timed-out rank's"
        assert o["per_step_after_health"] and o["lost_steps"] == 1
        assert o["moved"] and o["step_dev_2"] == 1 and np.isfinite(o["loss"])
    assert torch.equal(a["params"], b["params"]), "the replicas diverged"


def test_gradient_ranges_are_final_while_the_backward_pass_still_runs():
    """the early ranges of the data-parallel exchange are handed over BEFORE the backward pass ends -- on the device
    timeline, not in host order.  The full-width network (133.7 M parameters), one clip: every hand-over records an event on the
    stream it is final on (the third stream for the early ranges), an event closes the backward pass on the chain's stream.  Asserted:
    at least three ranges, the decoder's (47 % of the parameters) first; every early range's event precedes the end of the
    backward pass, the first by at least a third of the pass; the early ranges hold more than 70 % of the gradient bytes."""
    from sehip.model import Demucs
    torch.manual_seed(0)
    model = Demucs(sources=["clean"], audio_channels=2).cuda().train()
    g = torch.Generator().manual_seed(1)
    mix = (0.1 * torch.randn(1, 2, 24000, generator=g)).cuda()
    G = torch.randn(1, 1, 2, 24000, generator=g).cuda() / 100
    for _ in range(2):                                    # warm-up (allocations, first-use initialisation), then the measured pass
        ranges = []

        def hook(lo, hi, st):
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            ranges.append((lo, hi, e))
        est = model(mix)
        model.grad_range_hook = hook
        t0 = torch.cuda.Event(enable_timing=True); t0.record()
        est.backward(G)
        t1 = torch.cuda.Event(enable_timing=True); t1.record()
        model.grad_range_hook = None
        torch.cuda.synchronize()
        model.flat_grads.zero_()
        for _, prm in model._params:
            prm.grad = None
        model._grads_live = False
    n = model.flat_grads.numel()
    total = t0.elapsed_time(t1)
    early = ranges[:-1]
    lead = [e.elapsed_time(t1) for _, _, e in early]      # ms between a range being final and the end of the backward pass
    share = sum(hi - lo for lo, hi, _ in early) / n
    print(f"Demucs backward {total:.2f} ms; ranges {[(lo, hi) for lo, hi, _ in ranges]}; final {['%.2f' % v for v in lead]} ms before the end; "
          f"early share {share:.2f}")
    assert len(ranges) >= 3 and ranges[0][1] == n and ranges[0][0] == model.static.layout.param_off["decoder.0.0.weight"][0]
    assert all(v > 0 for v in lead) and lead[0] > total / 3 and share > 0.7
