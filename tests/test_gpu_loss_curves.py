""""SI-SNR parity" of a TRAINING path needs a loss curve, not 2-4 steps.  60 optimizer steps of the HIP
path (Solver.train_step: forward, loss, backward, clip 5, Adam 3e-4) against 60 steps of the fp32 CPU oracle from the same initial
weights on the same cycle of batches, for a reduced DCCRN (SI-SNR) and a reduced DCUnet (mse in the STFT domain) -- sizes the oracle
finishes in seconds.  Asserted: the curves stay together (every step, loose bound: two optimisers fed gradients that differ by bf16
rounding drift apart slowly), the mean of the last ten losses agrees to 0.2 dB / 2 %, and the loss went DOWN on both sides.
Reference step: src/solver.py:454-498; networks src/model/dccrn.py:145-229, src/model/dcunet.py:102-162."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dccrn_oracle as O
from oracle import dcunet_oracle as D

pytestmark = pytest.mark.gpu
STEPS = 60


def test_dccrn_sixty_steps_hip_vs_oracle(tmp_path):
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    from test_gpu_solver import solver_config, make_batch, KW
    cfg = solver_config(tmp_path)
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("stft.", "istft."))}
    solver = Solver(cfg, model, distrib.get_optimizer(cfg.optim, model), distrib.get_loss_function(cfg.optim), device="gpu",
                    writer=ScalarLog())
    batches = [make_batch(700 + s, 2, 4000) for s in range(6)]
    dev = [solver._prepare_batch(n, c) for n, c in batches]
    hip = []
    for s in range(STEPS):
        loss, _ = solver.train_step(*dev[s % 6])
        hip.append(float(loss))
    cfg_o = O.DCCRNConfig(**KW)
    adam = O.AdamState({k: v for k, v in p.items() if O.is_trainable(k)}, lr=3e-4)
    bases = O.stft_bases(cfg_o.win_len, cfg_o.fft_len)
    ref = []
    for s in range(STEPS):
        n, c = batches[s % 6]
        loss, _, _ = O.train_step(p, n, c[:, 0], cfg_o, adam, clip_grad=5, bases=bases)
        ref.append(loss)
    hip, ref = np.asarray(hip), np.asarray(ref)
    gap = np.abs(hip - ref)
    print(f"DCCRN {STEPS} steps: loss hip {hip[0]:.3f} -> {hip[-10:].mean():.3f}, oracle {ref[0]:.3f} -> {ref[-10:].mean():.3f}; "
          f"max |gap| {gap.max():.3f} dB at step {int(gap.argmax())}, last-10 mean gap {abs(hip[-10:].mean() - ref[-10:].mean()):.3f} dB")
    assert ref[-10:].mean() < ref[:6].mean() - 1.0 and hip[-10:].mean() < hip[:6].mean() - 1.0      # both train (loss = -SI-SNR, dB)
    assert gap.max() < 0.5
    assert abs(hip[-10:].mean() - ref[-10:].mean()) < 0.2


def test_dcunet_sixty_steps_hip_vs_oracle(tmp_path):
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    from oracle import stft_oracle as S
    from test_gpu_dcunet import c2_config
    cfg = c2_config(tmp_path, complexity=23)             # 16 / 32 complex channels
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("encoders.", "decoders."))}
    solver = Solver(cfg, model, distrib.get_optimizer(cfg.optim, model), distrib.get_loss_function(cfg.optim), device="gpu",
                    writer=ScalarLog())
    g = torch.Generator().manual_seed(8)
    batches = []
    for s in range(6):
        clean = torch.randn(2, 1, 1, 4096, generator=g)
        noisy = clean[:, 0] + 0.5 * torch.randn(2, 1, 4096, generator=g)
        batches.append((noisy, clean))
    hip = []
    for s in range(STEPS):
        n, c = batches[s % 6]
        loss, _ = solver.train_step(*solver._prepare_batch(n, c))     # stft_custom of mixture and sources inside the step
        hip.append(float(loss))
    specs = [torch.from_numpy(S.stft_custom(n.numpy(), 512, 128, 512)) for n, _ in batches]
    tgts = [torch.from_numpy(S.stft_custom(c[:, 0].numpy(), 512, 128, 512)) for _, c in batches]
    assert specs[0].shape == (2, 1, 257, 33, 2)
    names = [k for k in p if D.is_trainable(k)]
    leaves = {k: torch.nn.Parameter(p[k].clone()) for k in names}
    opt = torch.optim.Adam([leaves[k] for k in names], lr=3e-4, betas=(0.9, 0.999))
    work = dict(p); work.update(leaves)
    ref = []
    for s in range(STEPS):
        stats = {}
        est = D.dcunet_forward(work, specs[s % 6], model_complexity=23, model_depth=10, training=True, stats_out=stats)
        loss = F.mse_loss(est, tgts[s % 6])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([leaves[k] for k in names], 5.0)
        opt.step()
        work.update(stats)
        ref.append(float(loss.detach()))
    hip, ref = np.asarray(hip), np.asarray(ref)
    gap = np.abs(hip - ref) / ref
    print(f"DCUnet {STEPS} steps: loss hip {hip[0]:.4e} -> {hip[-10:].mean():.4e}, oracle {ref[0]:.4e} -> {ref[-10:].mean():.4e}; "
          f"max rel gap {gap.max():.3e} at step {int(gap.argmax())}, last-10 mean rel gap "
          f"{abs(hip[-10:].mean() - ref[-10:].mean()) / ref[-10:].mean():.3e}")
    assert ref[-10:].mean() < 0.97 * ref[:6].mean() and hip[-10:].mean() < 0.97 * hip[:6].mean()
    assert gap.max() < 0.05
    assert abs(hip[-10:].mean() - ref[-10:].mean()) < 0.02 * ref[-10:].mean()
