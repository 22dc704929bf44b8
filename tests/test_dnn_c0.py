"""BASELINE config C0 ("DNN magnitude-mask model on STFT, batch 2, PyTorch CPU, plumbing"): the stock-PyTorch DNN module
against vectors from the imported reference (oracle/gen_golden_dnn.py), and the whole config through main() / Solver on
CPU: registry, STFT branch of the train step (src/solver.py:454-458), mse in the STFT domain, Adam, checkpoints."""
import numpy as np
import pytest
import torch

from util import load_golden, rel_err

KW = dict(n_fft=64, nfft=64, hidden_layer=48, bias=True, activation="leaky-relu", drop_out=0.0, dnn_method="mask", dnn_ema=True)


def test_dnn_matches_reference_vectors():
    from sehip.model import DeepNeuralNetwork
    g = load_golden("dnn_c0.npz")
    model = DeepNeuralNetwork(**KW)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    assert set(model.state_dict()) == set(sd)                    # the reference's checkpoint keys, one for one
    model.load_state_dict(sd)
    x, tgt = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    model.train()
    y = model(x)
    assert rel_err(y.detach(), g["train_out"]) < 2e-5            # the EMA here is the closed form of the reference's time loop
    loss = torch.nn.functional.mse_loss(y, tgt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, p in model.named_parameters():
        if "grad." + k in g:
            ref = torch.from_numpy(g["grad." + k])
            # (a Linear bias in front of a BatchNorm has an analytically zero gradient: rounding noise on both sides)
            assert float((p.grad - ref).norm()) < 1e-4 * float(ref.norm()) + 2e-6, k
    for k, v in model.state_dict().items():
        if "stat." + k in g:
            assert rel_err(v.float(), torch.from_numpy(g["stat." + k]).float()) < 1e-5, k
    model.eval()
    with torch.no_grad():
        assert rel_err(model(x), g["eval_out"]) < 2e-5


def test_ema_closed_form_equals_time_loop():
    from sehip.model.dnn import ExponentialMovingAverage
    x = torch.randn(2, 40, 5, generator=torch.Generator().manual_seed(0))
    for alpha in (0.1, 0.85):
        out = ExponentialMovingAverage(alpha)(x)
        y, prev = [], None
        for t in range(x.shape[1]):                              # src/model/ema.py:27-37 restated
            prev = alpha * x[:, t] if t == 0 else (1 - alpha) * prev + alpha * x[:, t]
            y.append(prev)
        assert torch.allclose(out, torch.stack(y, 1), atol=1e-6)


def c0_config(tmp):
    from sehip.utils import dict2obj
    return dict2obj({
        "seed": 10, "root": None, "ha": None,
        "model": {"name": "dnn", "audio_channels": 1, "num_spk": 1, "sample_rate": 16000, "segment": 4,
                  "n_fft": 512, "hop_length": 128, "win_length": 512, "center": True,
                  "n_layers": 4, "hidden_layer": 64, "bias": True, "activation": "leaky-relu", "drop_out": 0.5,
                  "dnn_method": "mask", "dnn_ema": True},
        "optim": {"optim": "adam", "lr": 3e-4, "beta1": 0.9, "beta2": 0.999, "momentum": 0.9, "loss": "mse", "clip_grad": 5,
                  "pit": False, "load": False},
        "dset": {"name": "synthetic"},
        "solver": {"epochs": 1, "save_checkpoint_interval": 1, "all_steps": True, "total_steps": 0, "patience": 0,
                   "root": str(tmp), "resume": None, "preloaded_model": None,
                   "validation": {"interval": 1, "metric": "loss", "total_steps": 0}, "test": {"interval": 1}},
    })


def test_c0_runs_through_main_on_cpu(tmp_path):
    """configs[0] end to end: get_model('dnn') -> Solver(device='cpu') -> 2 train steps + validation + checkpoints."""
    from sehip.train import main
    from sehip.solver import ScalarLog
    g = torch.Generator().manual_seed(1)
    batches = []
    for s in range(2):                                          # B=2, 4 s @ 16 kHz like config C0
        clean = 0.1 * torch.randn(2, 1, 1, 64000, generator=g)
        noisy = clean[:, 0] + 0.05 * torch.randn(2, 1, 64000, generator=g)
        batches.append((noisy, clean, [None], [None], ["x"], [s]))
    log = ScalarLog()
    solver = main(c0_config(tmp_path), mode="train", device="cpu", train_dataloader=batches, validation_dataloader=[batches[0]],
                  writer=log)
    losses = [v for (t, v, _s) in log.scalars if t == "Train/Loss_step"]
    gnorm = [v for (t, v, _s) in log.scalars if t == "Train/grad_norm_step"]
    assert len(losses) == 2 and all(np.isfinite(losses)) and all(v > 0 for v in gnorm)
    names = sorted(f.name for f in solver.checkpoints_dir.glob("*"))
    assert "latest_model.tar" in names and "state.json" in names
    tar = torch.load(solver.checkpoints_dir / "latest_model.tar", weights_only=False)
    assert any(k.startswith("model.0.model.0.") for k in tar["model"]) and "ema_in.ema0" in tar["model"]


def test_main_has_the_reference_positional_signature(tmp_path):
    """src/train.py:18-23: main(obj_config, return_solver, mode, save, dev, device); :87-90: mode "validation" runs
    Solver._run_one_epoch(1, 1, train=False).  A reference-style positional call must bind `device` in the sixth position."""
    import inspect
    from sehip import SehipError
    from sehip.train import main
    from sehip.solver import ScalarLog
    names = list(inspect.signature(main).parameters)
    assert names[:6] == ["obj_config", "return_solver", "mode", "save", "dev", "device"]
    g = torch.Generator().manual_seed(2)
    clean = 0.1 * torch.randn(2, 1, 1, 16000, generator=g)
    noisy = clean[:, 0] + 0.05 * torch.randn(2, 1, 16000, generator=g)
    batches = [(noisy, clean, [None], [None], ["x"], [0])]
    log = ScalarLog()
    solver = main(c0_config(tmp_path), False, "validation", False, False, "cpu", train_dataloader=batches,
                  validation_dataloader=batches, writer=log)
    tags = [t for (t, _v, _s) in log.scalars]
    assert "Validation/Loss_step" in tags and "Validation/Loss" in tags and "Train/Loss_step" not in tags
    assert np.isfinite(solver.score["loss_valid"]) and solver.score["loss_valid"] > 0
    with pytest.raises(SehipError):
        main(c0_config(tmp_path), False, "test", False, False, "cpu", train_dataloader=batches, validation_dataloader=batches,
             writer=ScalarLog())
    with pytest.raises(TypeError):
        main(c0_config(tmp_path), False, "train", False, False, False)      # a non-device in the device position is refused by name


def test_hip_models_refuse_the_cpu_solver(tmp_path):
    from sehip import distrib, SehipError
    from sehip.solver import Solver, ScalarLog
    cfg = c0_config(tmp_path)
    cfg.model.name = "dccrn"
    with pytest.raises(SehipError):
        Solver(cfg, distrib.get_model(cfg.model), None, None, device="cpu", writer=ScalarLog())
