"""Permutation-invariant training loss (src/loss.py:58-100; SURVEY section 8 row f2, the opt-in config.optim.pit_apply).
CPU: oracle/pit_oracle.py against tests/golden/pit_cases.npz (the imported reference function, oracle/gen_golden_pit.py).
GPU: sehip.loss.pit_loss (fused device kernels for si-sdr, S*S launches for l1 / mse) against the same vectors:
loss within 2e-4 relative (fp32 reduction order), the chosen pairs identical, d loss / d enhance within 2e-4 relative."""
import numpy as np
import pytest
import torch

from util import load_golden, rel_err

CASES = ("s2_swap", "s2_id", "s3_rot", "s2_c2", "s2_l1", "s3_mse")


def _oracle_loss(name):
    from oracle import dccrn_oracle as O
    return {"sisdr": O.loss_sisdr, "l1": torch.nn.functional.l1_loss, "mse": torch.nn.functional.mse_loss}[name]


@pytest.mark.parametrize("case", CASES)
def test_oracle_pit_matches_reference_vectors(case):
    from oracle import pit_oracle
    g = load_golden("pit_cases.npz")
    est = torch.from_numpy(g[case + ".est"]).requires_grad_(True)
    tgt = torch.from_numpy(g[case + ".tgt"])
    loss, comb, _ = pit_oracle.pit(est, tgt, _oracle_loss(str(g[case + ".lname"])))
    loss.backward()
    want = float(g[case + ".loss"][0])
    assert abs(float(loss) - want) < 2e-5 * max(1.0, abs(want))
    assert [list(c) for c in comb] == g[case + ".comb"].tolist()
    assert rel_err(est.grad, torch.from_numpy(g[case + ".grad"])) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_pit_matches_reference_vectors(case):
    from sehip import loss as L
    dev = torch.device("cuda:0")
    g = load_golden("pit_cases.npz")
    fn = {"sisdr": L.loss_sisdr, "l1": L.l1_loss, "mse": L.mse_loss}[str(g[case + ".lname"])]
    est = torch.from_numpy(g[case + ".est"]).to(dev).requires_grad_(True)
    tgt = torch.from_numpy(g[case + ".tgt"]).to(dev)
    loss, comb = L.pit_loss(est, tgt, fn, return_comb=True)
    loss.backward()
    want = float(g[case + ".loss"][0])
    assert abs(float(loss) - want) < 2e-4 * max(1.0, abs(want))
    pairs = g[case + ".comb"].tolist()
    if torch.is_tensor(comb):      # fused path: perm[j] = estimated speaker matched with target j
        assert comb.cpu().tolist() == [i for i, _ in sorted(pairs, key=lambda p: p[1])]
    else:
        assert [list(c) for c in comb] == pairs
    assert rel_err(est.grad.cpu(), torch.from_numpy(g[case + ".grad"])) < 2e-4


@pytest.mark.gpu
def test_solver_pit_apply_opt_in():
    """ConvTasNet Solver step with optim.pit_apply: targets in swapped speaker order give (to rounding) the loss of the
    un-swapped step, while the default (the reference's behaviour: PIT computed and discarded) does not."""
    import copy
    from sehip import distrib
    from sehip.solver import Solver
    import tempfile
    from test_gpu_convtasnet import c4_config     # the C4 Solver configuration of that file
    cfg = c4_config(tempfile.mkdtemp(prefix="sehip_pit_"))
    torch.manual_seed(0)
    model = distrib.get_model(cfg.model)
    state = copy.deepcopy(model.state_dict())
    g = torch.Generator().manual_seed(1)
    B, N = 2, 8000
    src = 0.1 * torch.randn(B, 2, 1, N, generator=g)
    src[:, 1] *= 0.3                      # speakers of different level: the two orders give different plain losses
    mix = src.sum(1)

    def step(pit, swap):
        c = copy.deepcopy(cfg)
        c.optim.pit_apply = pit
        m = distrib.get_model(c.model)
        m.load_state_dict(state)
        s = Solver(c, m, distrib.get_optimizer(c.optim, m), distrib.get_loss_function(c.optim), device="gpu")
        mx, sr = s._prepare_batch(mix, src.flip(1) if swap else src)
        loss, _ = s.train_step(mx, sr)
        return float(loss)
    plain, plain_sw = step(False, False), step(False, True)
    pit, pit_sw = step(True, False), step(True, True)
    assert abs(pit - pit_sw) < 1e-4 * max(1.0, abs(pit))
    assert pit <= min(plain, plain_sw) + 1e-4
    assert abs(plain - plain_sw) > 1e-3 or abs(plain - pit) < 1e-4
