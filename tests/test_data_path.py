"""The data path in front of the step: WavDataset's normalise + crop (src/dataset.py:145-170,
src/utils.py:63-87) and collate_fn_pad (src/distrib.py:38-98).
CPU: oracle/data_oracle.py and the host arithmetic of sehip/data.py against tests/golden/collate_cases.npz (the imported reference
functions, oracle/gen_golden_collate.py).  GPU: sehip.data.DeviceBatcher (csrc/data.hip) against the same vectors -- fp32 kernels,
2e-6 absolute on O(1) normalised samples -- plus the reference's RNG order of the crop offsets and the ragged / empty edge cases."""
import types

import numpy as np
import pytest
import torch

from util import load_golden

CASES = ("crop_zs", "crop_stereo_2spk", "crop_pad_last", "nocrop_ragged", "nocrop_pad")


def _case(g, key):
    C, S, sl, seg, drop_last, zs, n = [int(v) for v in g[key + ".cfg"]]
    items = [(torch.from_numpy(g[f"{key}.mix{i}"]), torch.from_numpy(g[f"{key}.src{i}"]), f"utt{i}") for i in range(n)]
    return C, S, sl, seg, bool(drop_last), bool(zs), items


@pytest.mark.parametrize("key", CASES)
def test_oracle_and_host_plan_match_reference_vectors(key):
    from oracle import data_oracle as DO
    from sehip.data import plan_batch
    g = load_golden("collate_cases.npz")
    C, S, sl, seg, drop_last, zs, items = _case(g, key)
    starts = [int(v) for v in g[key + ".starts"]]
    proc = []
    for (m, s_, _), st in zip(items, starts):
        m, s_ = DO.normalise(m, s_, "z-score" if zs else "")
        if sl:
            m, s_ = DO.crop([m, s_], sl, st)
        proc.append((m, s_))
    bm, bs, idx = DO.collate(proc, seg, drop_last)
    assert idx == g[key + ".index_batch"].tolist()
    assert float((bm - torch.from_numpy(g[key + ".mixture"])).abs().max()) < 1e-6
    assert float((bs - torch.from_numpy(g[key + ".sources"])).abs().max()) < 1e-6
    plan = plan_batch([int(m.shape[-1]) for m, _, _ in items], seg, sl, drop_last, starts)
    assert [p[2] for p in plan] == idx


def test_crop_offsets_follow_the_reference_rng_order():
    from sehip.data import DeviceBatcher
    g = load_golden("collate_cases.npz")
    for key in ("crop_zs", "crop_stereo_2spk", "crop_pad_last"):
        C, S, sl, seg, drop_last, zs, items = _case(g, key)
        np.random.seed(7)                                   # the seed of the generator script
        b = DeviceBatcher(types.SimpleNamespace(segment=seg / 16000, sample_rate=16000), sample_length=sl, device="cpu")
        assert b.draw_starts([int(m.shape[-1]) for m, _, _ in items]) == g[key + ".starts"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("key", CASES)
def test_device_batcher_matches_reference_vectors(key):
    from sehip.data import DeviceBatcher
    g = load_golden("collate_cases.npz")
    C, S, sl, seg, drop_last, zs, items = _case(g, key)
    b = DeviceBatcher(types.SimpleNamespace(segment=seg / 16000, sample_rate=16000), normalize="z-score" if zs else "", sample_length=sl,
                      drop_last=drop_last)
    mixture, sources, mm, sm, names, idx = b(items, starts=[int(v) for v in g[key + ".starts"]])
    assert idx == g[key + ".index_batch"].tolist() and names == [f"utt{i}" for i in range(len(items))]
    assert tuple(mixture.shape) == g[key + ".mixture"].shape and tuple(sources.shape) == g[key + ".sources"].shape
    assert float((mixture.cpu() - torch.from_numpy(g[key + ".mixture"])).abs().max()) < 2e-6 * max(1.0, float(np.abs(g[key + ".mixture"]).max()))
    assert float((sources.cpu() - torch.from_numpy(g[key + ".sources"])).abs().max()) < 2e-6 * max(1.0, float(np.abs(g[key + ".sources"]).max()))
    if zs:      # the per-utterance dictionaries of src/dataset.py:131-152
        m0 = items[0][0]
        assert float((mm[0]["mean"].cpu() - m0.mean(-1, keepdim=True)).abs().max()) < 1e-6
        assert float((mm[0]["std"].cpu() - m0.std(-1, keepdim=True)).abs().max()) < 1e-6
        assert tuple(sm[0]["mean"].shape) == (S, C, 1)


@pytest.mark.gpu
def test_device_batcher_feeds_the_solver_and_edge_cases():
    """The tuple goes through Solver._prepare_batch + train_step unchanged (DCCRN, 2000-sample segments); linear-scale mode against the
    oracle; an empty batch, an unknown normalisation and mismatched shapes raise."""
    from oracle import data_oracle as DO
    from sehip import distrib
    from sehip._lib import SehipError
    from sehip.data import DeviceBatcher
    from sehip.solver import Solver, ScalarLog
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    g = torch.Generator().manual_seed(3)
    items = [(0.1 * torch.randn(1, n, generator=g), 0.1 * torch.randn(1, 1, n, generator=g), f"u{i}") for i, n in enumerate((9000, 5000, 4100))]
    cfg_d = types.SimpleNamespace(segment=0.25, sample_rate=16000)
    b = DeviceBatcher(cfg_d, normalize="linear-scale", sample_length=0, drop_last=True)
    mixture, sources, _, _, _, idx = b(items)
    proc = [DO.normalise(m, s_, "linear-scale") for m, s_, _ in items]
    bm, bs, idx_ref = DO.collate(proc, 4000, True)
    assert idx == idx_ref == [2, 1, 1]
    assert float((mixture.cpu() - bm).abs().max()) < 2e-6 and float((sources.cpu() - bs).abs().max()) < 2e-6
    cfg = bench.bench_config(4000)
    cfg.model.kernel_num = [16, 16, 32, 32, 64, 64]
    torch.manual_seed(0)
    model = distrib.get_model(cfg.model)
    solver = Solver(cfg, model, distrib.get_optimizer(cfg.optim, model), distrib.get_loss_function(cfg.optim), device="gpu", writer=ScalarLog())
    batch = DeviceBatcher(cfg_d, normalize="z-score")(items)
    mix, src = solver._prepare_batch(batch[0], batch[1])
    loss, _ = solver.train_step(mix, src)
    assert np.isfinite(float(loss)) and tuple(mix.shape) == (4, 1, 4000)
    with pytest.raises(SehipError):
        b([])
    with pytest.raises(SehipError):
        DeviceBatcher(cfg_d, normalize="bogus")
    with pytest.raises(SehipError):
        b([(torch.zeros(2, 100), torch.zeros(1, 1, 100), "shape mismatch")])
