"""CPU: pins oracle/dccrn_oracle.py against vectors produced by the imported reference
(oracle/gen_golden.py).  Tolerances are fp32 round-off of two different op orders."""
import math
import numpy as np
import pytest
import torch

from oracle import dccrn_oracle as O
from util import load_golden, sub, json_entry, rel_err, max_abs, golden_grads, make_batch

TINY = dict(rnn_units=16, kernel_num=[4, 4, 8, 8, 16, 16], length=4000)


@pytest.fixture(scope="module")
def fx():
    return load_golden("dccrn_tiny_fwd_bwd.npz")


def test_stft_bases_match_reference_rows():
    g = load_golden("stft_bases_rows.npz")
    a, s, w = O.stft_bases(400, 512)
    rows = g["rows"]
    assert max_abs(a[rows], g["stft"]) < 1e-6
    assert max_abs(s[rows], g["istft"]) < 1e-6
    assert max_abs(w, g["window"]) < 1e-7


def test_sisnr_cases():
    g = load_golden("sisnr_cases.npz")
    for k in ("a", "b", "zero_target", "equal"):
        v = O.si_snr(torch.from_numpy(g[k + "/est"]), torch.from_numpy(g[k + "/ref"]))
        assert abs(float(v) - float(g[k + "/si_snr"])) < 2e-4 * max(1.0, abs(float(g[k + "/si_snr"]))), k


def test_forward_activations(fx):
    cfg = O.DCCRNConfig(**TINY)
    p = sub(fx, "state")
    cap, stats = {}, {}
    est = O.dccrn_forward(p, torch.from_numpy(fx["noisy"]), cfg, training=True, capture=cap, stats_out=stats)
    acts = sub(fx, "act")
    assert rel_err(cap["stft"], acts["stft"]) < 1e-5
    for i in range(6):
        assert rel_err(cap[f"enc{i}.conv"], acts[f"enc{i}.conv"]) < 2e-5, i
        assert rel_err(cap[f"enc{i}"], acts[f"enc{i}"]) < 2e-5, i
        assert rel_err(cap[f"dec{i}"], acts[f"dec{i}.full"][..., 1:]) < 5e-5, i
    assert rel_err(cap["lstm0.r"], acts["lstm0.r"]) < 2e-5
    assert rel_err(cap["lstm1.i"], acts["lstm1.r.i"]) < 2e-5
    assert rel_err(cap["istft"], acts["istft"]) < 5e-5
    assert rel_err(est, fx["est"]) < 5e-5
    for k, v in sub(fx, "state_after").items():
        assert rel_err(stats[k].float(), v.float()) < 1e-5, k


def test_loss_and_gradients(fx):
    cfg = O.DCCRNConfig(**TINY)
    p = sub(fx, "state")
    names = [k for k in p if O.is_trainable(k)]
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    est = O.dccrn_forward(work, torch.from_numpy(fx["noisy"]), cfg, training=True)
    loss = O.loss_sisdr(est, torch.from_numpy(fx["clean"])[:, 0])
    assert abs(float(loss.detach()) - float(fx["loss"])) < 1e-4
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    ref = sub(fx, "grad")
    assert set(ref) == set(names)
    for k, g in zip(names, grads):
        # conv biases that feed a BatchNorm have an analytically zero gradient: both sides are
        # round-off noise there, hence the absolute floor.
        err = float((g - ref[k]).norm())
        assert err < 2e-3 * float(ref[k].norm()) + 2e-6 * g.numel() ** 0.5, (k, err, float(ref[k].norm()))


def test_eval_mode(fx):
    cfg = O.DCCRNConfig(**TINY)
    p = sub(fx, "state")
    p.update(sub(fx, "state_after"))
    est = O.dccrn_forward(p, torch.from_numpy(fx["noisy"]), cfg, training=False)
    assert rel_err(est, fx["est_eval"]) < 5e-5


def test_two_solver_steps():
    """Oracle train_step x2 vs the real Solver.train() (loss log, grad_norm metric, weights, Adam state)."""
    g = load_golden("dccrn_tiny_solver.npz")
    cfg = O.DCCRNConfig(**TINY)
    p = sub(g, "state0")
    adam = O.AdamState({k: v for k, v in p.items() if O.is_trainable(k)}, lr=3e-4, betas=(0.9, 0.999))
    log = json_entry(g, "log_json")
    losses = [v for (tag, v, _s) in log if tag == "Train/Loss_step"]
    gnorms = [v for (tag, v, _s) in log if tag == "Train/grad_norm_step"]
    for s in range(2):
        loss, metric, _ = O.train_step(p, torch.from_numpy(g[f"noisy{s}"]),
                                       torch.from_numpy(g[f"clean{s}"])[:, 0], cfg, adam, clip_grad=5)
        assert abs(loss - losses[s]) < 2e-3 * abs(losses[s]), (s, loss, losses[s])
        assert abs(metric - gnorms[s]) < 2e-2 * abs(gnorms[s]), (s, metric, gnorms[s])
    ref = sub(g, "state2")
    for k, v in ref.items():
        if v.dtype == torch.int64:
            assert int(p[k]) == int(v), k
        else:
            # a conv bias in front of a BatchNorm has zero true gradient; Adam turns the round-off
            # noise into +-lr steps of arbitrary sign, so those entries may differ by 2*steps*lr.
            noise_bias = k.endswith("conv.bias") and not k.startswith("decoder.5.")
            assert max_abs(p[k], v) < (1.3e-3 if noise_bias else 2e-4), (k, max_abs(p[k], v))
    for k, v in sub(g, "adam_m").items():
        if k.endswith("conv.bias") and not k.startswith("decoder.5."):
            continue
        assert rel_err(adam.m[k], v) < 5e-3, k
    keys = json_entry(g, "ckpt_keys_json")
    assert keys["top"] == ["best_score", "epoch", "model", "optimizer"]
    assert len(keys["model"]) == 204


LEGAL = dict(rnn_units=128, kernel_num=[16, 16, 32, 32, 64, 64], length=4000)
C1 = dict(rnn_units=128, kernel_num=[16, 32, 64, 128, 256, 256], length=32000)


def _oracle_fwd_bwd(kw, seed, pseed, bseed, b, n):
    cfg = O.DCCRNConfig(**kw)
    p = O.perturb_params(O.init_params(cfg, seed=seed), pseed)
    noisy, clean = make_batch(bseed, b, n)
    names = [k for k in p if O.is_trainable(k)]
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    stats = {}
    est = O.dccrn_forward(work, noisy, cfg, training=True, stats_out=stats)
    loss = O.loss_sisdr(est, clean[:, 0])
    grads = dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])))
    return cfg, p, noisy, est.detach(), float(loss.detach()), grads, stats


def test_hip_legal_config_against_reference():
    """A configuration the HIP path accepts (the tiny golden model is below its channel minimum): the oracle against the
    reference's outputs for weights rebuilt from a seed.  tests/test_gpu_c1_fullsize.py compares HIP with the same file."""
    g = load_golden("dccrn_legal_fwd_bwd.npz")
    cfg, p, noisy, est, loss, grads, stats = _oracle_fwd_bwd(LEGAL, 21, 22, 23, 2, 4000)
    assert rel_err(est, g["est"]) < 5e-5 and abs(loss - float(g["loss"])) < 2e-4
    full, norms = golden_grads(g)
    assert set(full) == set(grads)
    for k, gr in grads.items():
        # fp16 storage: 5e-4 of the tensor's largest element per entry; conv biases in front of a BatchNorm have an
        # analytically zero gradient (round-off noise on both sides): absolute floor
        tol = 5e-3 * norms[k] + 6e-4 * float(full[k].abs().max()) * gr.numel() ** 0.5 + 2e-6 * gr.numel() ** 0.5
        assert float((gr - full[k]).norm()) < tol, (k, float((gr - full[k]).norm()), norms[k])
    for k, v in sub(g, "state_after").items():
        assert rel_err(stats[k].float(), v.float()) < 1e-5, k
    q = dict(p); q.update({k: v for k, v in stats.items()})
    assert rel_err(O.dccrn_forward(q, noisy, cfg, training=False), g["est_eval"]) < 5e-5


def test_full_size_c1_checksum():
    """The headline model (kernel_num 16-32-64-128-256-256, 32000 samples) at B=2: oracle vs the reference's loss, waveform
    and gradient norms (one full-size C1 checksum)."""
    g = load_golden("dccrn_c1_checksum.npz")
    cfg, p, noisy, est, loss, grads, stats = _oracle_fwd_bwd(C1, 10, 11, 0, 2, 32000)
    assert abs(loss - float(g["loss"])) < 5e-4
    assert abs(float(est.double().norm()) - float(g["est_l2"])) < 1e-4 * float(g["est_l2"])
    assert rel_err(est, g["est16"].astype(np.float32)) < 1e-3          # float16 storage
    full, norms = golden_grads(g)
    for k, gr in grads.items():
        assert abs(float(gr.double().norm()) - norms[k]) < 3e-3 * norms[k] + 2e-5, (k, float(gr.norm()), norms[k])


def test_param_init_shapes_match_reference_schema():
    g = load_golden("dccrn_tiny_solver.npz")
    ref = sub(g, "state0")
    mine = O.init_params(O.DCCRNConfig(**TINY))
    assert set(mine) == set(ref)
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k


# ---- stft_custom / istft_custom: oracle/stft_oracle.py vs the reference's outputs
def test_stft_custom_oracle_matches_reference():
    from oracle import stft_oracle as S
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "stft_custom.npz"))
    names = sorted({k.split(".")[0] for k in g.files})
    assert len(names) == 9             # five with n_fft 512, four with other n_fft (256, 1024, 400, 255: round 6)
    for name in names:
        n_fft, hop, win, length = [int(v) for v in g[name + ".cfg"]]
        s = S.stft_custom(g[name + ".x"], n_fft, hop, win)
        assert s.shape == g[name + ".stft"].shape
        assert np.abs(s - g[name + ".stft"]).max() < 5e-8
        y = S.istft_custom(g[name + ".z"], length, n_fft, hop, win)
        ref = g[name + ".istft"]
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() < 1e-6 * np.abs(ref).max()
        rt = S.istft_custom(g[name + ".stft"], length, n_fft, hop, win)
        assert np.abs(rt - g[name + ".roundtrip"]).max() < 1e-6


# ---- complex DCUNet: oracle/dcunet_oracle.py vs the reference's activations / gradients
def test_dcunet_oracle_matches_reference():
    import os
    from oracle import dcunet_oracle as D
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dcunet_tiny.npz"))
    p = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    n_lin = sum(k.startswith("linear.") for k in p)
    assert int(g["n_state_dict_keys"]) == 2 * (len(p) - n_lin) + n_lin    # every block is registered twice in the reference
    x, tgt = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    # eval mode: running statistics
    out_eval = D.dcunet_forward(p, x, model_complexity=8, model_depth=10, training=False)
    assert rel_err(out_eval, torch.from_numpy(g["eval_out"])) < 2e-5
    # train mode: activations, output, loss, gradients, running-stat updates
    names = [k for k in p if D.is_trainable(k)]
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    taps, stats = {}, {}
    est = D.dcunet_forward(work, x, model_complexity=8, model_depth=10, training=True, stats_out=stats, taps=taps)
    for k in taps:
        assert rel_err(taps[k].detach(), torch.from_numpy(g["tap." + k])) < 2e-5, k
    assert rel_err(est.detach(), torch.from_numpy(g["train_out"])) < 2e-5
    loss = torch.nn.functional.mse_loss(est, tgt)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    assert len(names) == 84
    for k, gr in zip(names, grads):
        ref = torch.from_numpy(g["grad." + k])
        # (a conv bias in front of a BatchNorm has an analytically zero gradient: ~1e-9 of rounding noise on both sides)
        assert float((gr - ref).norm()) <= 2e-4 * float(ref.norm()) + 1e-7, k
    for k, v in stats.items():
        assert rel_err(v, torch.from_numpy(g["stat." + k])) < 1e-5, k
    assert len(stats) == 40


def test_dcunet20_oracle_matches_reference():
    """The depth-20 tables (src/model/dcunet.py:215-305) on [1, 1, 257, 257, 2], the one shape the reference accepts at that depth."""
    from oracle import dcunet_oracle as D
    g = load_golden("dcunet20_tiny.npz")
    p = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    x, tgt = torch.from_numpy(g["x"]).float(), torch.from_numpy(g["target"]).float()      # stored as fp16, exactly representable
    out_eval = D.dcunet_forward(p, x, model_complexity=8, model_depth=20, training=False)
    assert rel_err(out_eval, torch.from_numpy(g["eval_out"])) < 2e-5
    names = [k for k in p if D.is_trainable(k)]
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    stats = {}
    est = D.dcunet_forward(work, x, model_complexity=8, model_depth=20, training=True, stats_out=stats)
    assert rel_err(est.detach(), torch.from_numpy(g["train_out"])) < 2e-5
    loss = torch.nn.functional.mse_loss(est, tgt)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    assert len(names) == 164
    for k, gr in zip(names, grads):
        ref = torch.from_numpy(g["grad." + k])
        assert float((gr - ref).norm()) <= 5e-4 * float(ref.norm()) + 1e-7, k
    for k, v in stats.items():
        assert rel_err(v, torch.from_numpy(g["stat." + k])) < 1e-5, k
    assert len(stats) == 80


# ---- ConvTasNet: oracle/convtasnet_oracle.py vs the reference's activations / gradients
@pytest.mark.parametrize("fixture,extra", [("convtasnet_tiny.npz", {}), ("convtasnet_tiny_softmax.npz", dict(mask_nonlinear="softmax"))])
def test_convtasnet_oracle_matches_reference(fixture, extra):
    """(softmax: mask_nonlinear='softmax', src/model/conv_tasnet.py:298-299 -- F.softmax over the sources)"""
    from oracle import convtasnet_oracle as CT
    g = load_golden(fixture)
    kw = dict(C=2, N=16, L=8, B=16, H=32, P=3, X=3, R=2, audio_channels=1, **extra)
    p = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    taps = {}
    est = CT.convtasnet_forward(leaves, torch.from_numpy(g["mix"]), taps=taps, **kw)
    for k, v in taps.items():
        assert rel_err(v.detach(), g["tap." + k]) < 2e-5, k
    assert est.shape == (2, 2, 1, 404) and rel_err(est.detach(), g["est"]) < 2e-5
    loss = O.loss_sisdr(est, torch.from_numpy(g["target"]))
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    names = sorted(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    assert len(names) == len([k for k in g if k.startswith("grad.")])
    for k, gr in zip(names, grads):
        ref = torch.from_numpy(g["grad." + k])
        assert float((gr - ref).norm()) <= 5e-4 * float(ref.norm()) + 1e-6, k


DEMUCS_CASES = {
    "a": dict(sources=["s0", "s1"], audio_channels=2, channels=4, depth=5, norm_starts=3, dconv_lstm=3, dconv_attn=3, resample=False),
    "b": dict(sources=["s0"], audio_channels=1, channels=8, depth=3, dconv_lstm=1, dconv_attn=2, norm_starts=1, resample=False),
}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_demucs_oracle_matches_reference(tag):
    """Demucs with resample=False against vectors of the imported reference: every encoder / decoder output, the
    separated sources, the SI-SNR loss, every parameter gradient; case b runs the BLSTM on overlapping chunks (T > 200)."""
    from oracle import demucs_oracle as DM
    g = {k[2:]: v for k, v in load_golden("demucs_tiny.npz").items() if k.startswith(tag + ".")}
    cfg = DM.DemucsConfig(**DEMUCS_CASES[tag])
    p = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    shapes = DM.param_shapes(cfg)
    assert [n for n, _ in shapes] == list(g["names"]), "parameters() order"
    assert all(tuple(p[n].shape) == s for n, s in shapes)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    taps = {}
    mix = torch.from_numpy(g["mix"])
    est = DM.demucs_forward(leaves, mix, cfg, taps=taps)
    assert len(taps) == 2 * cfg.depth
    for k, v in taps.items():
        assert rel_err(v.detach(), g["tap." + k]) < 2e-5, k
    assert tuple(est.shape) == tuple(g["est"].shape) and rel_err(est.detach(), g["est"]) < 2e-5
    loss = O.loss_sisdr(est, torch.from_numpy(g["target"]))
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    names = sorted(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    assert len(names) == len([k for k in g if k.startswith("grad.")])
    for k, gr in zip(names, grads):
        ref = torch.from_numpy(g["grad." + k])
        assert float((gr - ref).norm()) <= 1e-3 * float(ref.norm()) + 1e-6, k


def test_demucs_resampler_properties():
    """julius.resample_frac restated (parity UNPINNED: the package is absent, oracle/demucs_oracle.py): lengths, unit DC gain, a
    band-limited sine survives x2 -> /2, and the x2 output interleaves the input samples with interpolated ones."""
    from oracle import demucs_oracle as DM
    x = torch.ones(1, 1, 300)
    up = DM.resample_frac(x, 1, 2)
    assert up.shape[-1] == 600 and float((up - 1).abs().max()) < 1e-5
    assert DM.resample_frac(torch.ones(1, 1, 601), 2, 1).shape[-1] == 300
    t = torch.arange(2000, dtype=torch.float32)
    s = torch.sin(2 * math.pi * 0.05 * t)[None, None]
    up = DM.resample_frac(s, 1, 2)
    ref = torch.sin(2 * math.pi * 0.025 * torch.arange(4000, dtype=torch.float32))
    assert float((up[0, 0, 100:-100] - ref[100:-100]).abs().max()) < 2e-3
    back = DM.resample_frac(up, 2, 1)
    assert back.shape == s.shape and float((back - s)[..., 100:-100].abs().max()) < 2e-3
    k, width, _, _ = DM.resample_kernels(1, 2)
    assert width == 26 and tuple(k.shape) == (2, 53) and float((k.sum(1) - 1).abs().max()) < 1e-6


# ---- constructor options beyond the shipped YAML (round 6): tests/golden/dccrn_variants.npz (oracle/gen_golden_dccrn_variants.py)
@pytest.mark.parametrize("case,extra", [("hamming", dict(win_type="hamming")), ("none", dict(win_type=None)),
                                        ("blackman", dict(win_type="blackman")), ("realbn", dict(use_cbn=False)),
                                        ("rnn1", dict(rnn_layers=1)), ("rnn3", dict(rnn_layers=3)), ("ru256", dict(rnn_units=256)),
                                        ("reallstm", dict(use_clstm=False))])
def test_window_types_against_reference(case, extra):
    """Constructor options of the reference beyond the shipped YAML (src/model/dccrn.py:12-27): win_type (init_kernels :650-653),
    use_cbn=False (nn.BatchNorm2d instead of ComplexBatchNorm, :110-113) and rnn_layers (:84-96): the oracle against the imported reference's waveform, loss,
    gradients, eval-mode waveform and running statistics."""
    g = {k[len(case) + 1:]: v for k, v in load_golden("dccrn_variants.npz").items() if k.startswith(case + "/")}
    kw = dict(LEGAL, **extra)
    assert np.allclose(O.window_of(extra.get("win_type", "hann"), 400), g["window"], atol=1e-7)
    cfg, p, noisy, est, loss, grads, stats = _oracle_fwd_bwd(kw, 31, 32, 33, 2, 4000)
    assert rel_err(est, g["est"]) < 5e-5 and abs(loss - float(g["loss"])) < 2e-4
    full, norms = golden_grads(g)
    assert set(full) == set(grads)
    for k, gr in grads.items():
        tol = 5e-3 * norms[k] + 6e-4 * float(full[k].abs().max()) * gr.numel() ** 0.5 + 2e-6 * gr.numel() ** 0.5
        if k.endswith("conv.bias") and not k.startswith("decoder.5."):
            # a convolution bias in front of a BatchNorm: the true gradient is zero, both sides hold fp32 round-off (|g| ~ 1e-6 per
            # entry, uncorrelated): bounded, not compared
            assert float(gr.norm()) < 1e-4 and norms[k] < 1e-4, (k, float(gr.norm()), norms[k])
            continue
        assert float((gr - full[k]).norm()) < tol, (k, float((gr - full[k]).norm()), norms[k])
    for k, v in sub(g, "state_after").items():           # running statistics after the step (running_var: unbiased for nn.BatchNorm2d)
        assert rel_err(stats[k].float(), v.float()) < 1e-5, k
    q = dict(p); q.update({k: v for k, v in stats.items()})
    assert rel_err(O.dccrn_forward(q, noisy, cfg, training=False), g["est_eval"]) < 5e-5
