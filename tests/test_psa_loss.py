"""The phase-sensitive spectral approximation loss (`optim.loss: psa`; src/loss.py:32-56, src/distrib.py:270-271, called with the mixture
as third argument by src/solver.py:480).  CPU: oracle/loss_oracle.py against vectors of the imported reference
(tests/golden/psa_loss.npz, oracle/gen_golden_psa.py).  GPU: sehip_psa_loss_fwd / _bwd against the same vectors and, at the C2 size,
against the oracle; a Solver step of DCUnet under it."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle as LO
from util import load_golden, rel_err


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_against_reference_vectors(case):
    g = {k[2:]: torch.from_numpy(np.asarray(v)) for k, v in load_golden("psa_loss.npz").items() if k.startswith(case + "/")}
    enh = g["enh"].clone().requires_grad_(True)
    loss = LO.psa_loss(enh, g["tgt"], g["mix"])
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    loss.backward()
    assert rel_err(enh.grad, g["denh"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["a", "b"])
def test_hip_against_reference_vectors(case):
    from sehip.loss import loss_phase_sensitive_spectral_approximation as psa
    g = {k[2:]: torch.from_numpy(np.asarray(v)) for k, v in load_golden("psa_loss.npz").items() if k.startswith(case + "/")}
    enh = g["enh"].cuda().requires_grad_(True)
    loss = psa(enh, g["tgt"].cuda(), g["mix"].cuda())
    assert abs(float(loss) - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
    loss.backward()
    assert rel_err(enh.grad.cpu(), g["denh"]) < 2e-6           # (fp32 on both sides; tanh / cos / sqrt of the two libraries)


@pytest.mark.gpu
def test_hip_against_oracle_at_the_c2_size():
    """[8, 1, 257, 257, 2] (an eighth of the C2 batch), a scaled upstream gradient, an all-zero enhanced bin (gradient 0 here, NaN in
    torch: documented in include/sehip.h), the factory and the shape check."""
    from sehip import distrib, SehipError
    from sehip.utils import dict2obj
    psa = distrib.get_loss_function(dict2obj({"loss": "psa"}))
    gen = torch.Generator().manual_seed(9)
    enh, tgt, mix = (torch.randn(8, 1, 257, 257, 2, generator=gen) for _ in range(3))
    enh[0, 0, 0, 0] = 0.0
    e = enh.cuda().requires_grad_(True)
    loss = psa(e, tgt.cuda(), mix.cuda())
    (3.0 * loss).backward()
    eo = enh.double().clone().requires_grad_(True)
    lo = LO.psa_loss(eo, tgt.double(), mix.double())
    (3.0 * lo).backward()
    assert abs(float(loss) - float(lo)) < 1e-5 * float(lo)
    want = eo.grad.clone()
    assert torch.isnan(want[0, 0, 0, 0]).all()
    want[0, 0, 0, 0] = 0.0
    assert torch.isfinite(e.grad).all() and rel_err(e.grad.cpu().double(), want) < 1e-5
    with pytest.raises(SehipError):
        psa(e, tgt.cuda()[:4], mix.cuda())
    with pytest.raises(SehipError):
        psa(enh, tgt, mix)                       # CPU tensors: there is no CPU path


@pytest.mark.gpu
def test_solver_step_of_dcunet_under_psa(tmp_path):
    """Solver.train() with `optim.loss: psa` on the STFT branch (src/solver.py:454-458, :480): the first step's loss against the oracle
    (DCUnet forward + psa on the spectra of noisy / clean), then a second step that must lower it."""
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    from oracle import stft_oracle as S, dcunet_oracle as D
    from test_gpu_dcunet import c2_config
    cfg = c2_config(tmp_path, complexity=8)
    cfg.optim.loss = "psa"
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("encoders.", "decoders."))}
    opt = distrib.get_optimizer(cfg.optim, model)
    g = torch.Generator().manual_seed(5)
    clean = 0.1 * torch.randn(2, 1, 1, 8192, generator=g)
    noisy = clean[:, 0] + 0.05 * torch.randn(2, 1, 8192, generator=g)
    batch = (noisy, clean, [None], [None], ["x"], [0])
    log = ScalarLog()
    solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), train_dataloader=[batch, batch],
                    validation_dataloader=[batch], device="gpu", writer=log)
    solver._run_one_epoch(0, 1, train=True)
    losses = [v for (t, v, _s) in log.scalars if t == "Train/Loss_step"]
    spec = torch.from_numpy(S.stft_custom(noisy.numpy(), 512, 128, 512))
    tgt = torch.from_numpy(S.stft_custom(clean[:, 0].numpy(), 512, 128, 512))
    est = D.dcunet_forward(p, spec, model_complexity=8, model_depth=10, training=True)
    ref = float(LO.psa_loss(est, tgt, spec))
    print("psa Solver steps:", losses, "oracle first step", ref)
    assert abs(losses[0] - ref) < 2e-2 * abs(ref) and losses[1] < losses[0]
