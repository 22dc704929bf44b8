#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the real reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
Needs /root/reference (read-only).  Nothing from the reference is copied: the fixtures hold
inputs and the reference's numeric outputs only.  The reference never travels to the GPU box;
these vectors pin oracle/dccrn_oracle.py, which does.

Fixtures
  dccrn_tiny_fwd_bwd.npz : tiny DCCRN (kernel_num [4,4,8,8,16,16], rnn_units 16, N=4000, B=2):
                           state_dict, noisy/clean, per-layer activations (forward hooks),
                           SI-SNR loss, every parameter gradient.
  dccrn_tiny_solver.npz  : the real ``Solver.train()`` (src/solver.py:355-532) for 1 epoch x 2
                           steps on synthetic batches with non-numeric deps stubbed: logged
                           Train/Loss_step, Train/grad_norm_step, final state_dict, Adam state.
  dccrn_legal_fwd_bwd.npz: a configuration the HIP path accepts (kernel_num [16,16,32,32,64,64], rnn_units 128,
                           N=4000, B=2) so that one GPU test compares HIP with reference vectors DIRECTLY.  Weights are
                           rebuilt from a seed (oracle.init_params + perturb_params) and loaded into the reference
                           model, so only outputs are stored: waveform, loss, eval-mode waveform, running statistics,
                           every parameter gradient (float16 x per-tensor scale).
  dccrn_c1_checksum.npz  : the FULL-SIZE C1 model (kernel_num [16,32,64,128,256,256], N=32000, B=2), same recipe:
                           loss, waveform (float16) + its exact L2 norm, per-tensor gradient norms, small gradients.
  stft_bases_rows.npz    : sampled rows of stft.weight / istft.weight (pins init_kernels).
  sisnr_cases.npz        : si_snr on a few shapes incl. zero target.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

TINY = dict(rnn_units=16, kernel_num=[4, 4, 8, 8, 16, 16], length=4000)


def to_np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def make_batch(seed, b, n):
    g = torch.Generator().manual_seed(seed)
    clean = 0.1 * torch.randn(b, 1, 1, n, generator=g)
    noisy = clean[:, 0] + 0.05 * torch.randn(b, 1, n, generator=g)
    return noisy, clean


def fwd_bwd_fixture():
    from src.model.dccrn import DCCRN
    from src.loss import loss_sisdr
    torch.manual_seed(10)
    model = DCCRN(**TINY)
    # make BN affine / PReLU / biases non-trivial so the fixture exercises them
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith(".bias") or name.endswith((".Br", ".Bi")):
                prm.copy_(0.1 * torch.randn(prm.shape, generator=g))
            if name.endswith((".Wrr", ".Wii")):
                prm.copy_(1.0 + 0.2 * torch.randn(prm.shape, generator=g))
            if name.endswith("2.weight"):
                prm.copy_(0.25 + 0.1 * torch.randn(prm.shape, generator=g))
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    noisy, clean = make_batch(0, 2, 4000)
    acts = {}

    def hook(name):
        def f(_m, _i, o):
            acts[name] = (o[0] if isinstance(o, (list, tuple)) else o).detach().clone()
            if isinstance(o, (list, tuple)):
                acts[name + ".i"] = o[1].detach().clone()
        return f

    model.stft.register_forward_hook(hook("stft"))
    model.istft.register_forward_hook(hook("istft"))
    for i, layer in enumerate(model.encoder):
        layer[0].register_forward_hook(hook(f"enc{i}.conv"))
        layer.register_forward_hook(hook(f"enc{i}"))
    for i, layer in enumerate(model.decoder):
        layer[0].register_forward_hook(hook(f"dec{i}.conv_full"))
        layer.register_forward_hook(hook(f"dec{i}.full"))
    for i, layer in enumerate(model.enhance):
        layer.register_forward_hook(hook(f"lstm{i}.r"))
    model.train()
    est = model(noisy)
    loss = loss_sisdr(est, clean[:, 0])
    loss.backward()
    out = {}
    for k, v in state0.items():
        if k in ("stft.weight", "istft.weight", "istft.enframe", "istft.window"):
            continue
        out["state/" + k] = v.numpy()
    for k, v in model.state_dict().items():
        if k.endswith(("RMr", "RMi", "RVrr", "RVri", "RVii", "num_batches_tracked")):
            out["state_after/" + k] = v.numpy()
    out["noisy"] = noisy.numpy()
    out["clean"] = clean.numpy()
    out["est"] = est.detach().numpy()
    out["loss"] = np.float32(loss.item())
    for k, v in acts.items():
        out["act/" + k] = v.numpy()
    for k, prm in model.named_parameters():
        out["grad/" + k] = prm.grad.numpy()
    # eval-mode forward with the (now updated) running statistics
    model.eval()
    with torch.no_grad():
        out["est_eval"] = model(noisy).numpy()
    np.savez_compressed(os.path.join(OUT, "dccrn_tiny_fwd_bwd.npz"), **out)
    print("fwd_bwd: loss", loss.item(), "keys", len(out))

    w = model.state_dict()
    rows = np.arange(0, 514, 37)
    np.savez_compressed(os.path.join(OUT, "stft_bases_rows.npz"),
                        rows=rows,
                        stft=w["stft.weight"][rows, 0].numpy(),
                        istft=w["istft.weight"][rows, 0].numpy(),
                        window=w["istft.window"][0, :, 0].numpy())


def _reference_from_seed(kw, seed, perturb_seed):
    """Reference DCCRN carrying the weights oracle.init_params(seed) + perturb_params(perturb_seed) produce."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import dccrn_oracle as O
    from src.model.dccrn import DCCRN
    p = O.perturb_params(O.init_params(O.DCCRNConfig(**kw), seed=seed), perturb_seed)
    model = DCCRN(**kw)
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected and all(k.startswith(("stft.", "istft.")) for k in missing), (missing, unexpected)
    return model


def _pack_grads(model, out, full_below):
    """float16 x per-tensor scale for every gradient with fewer than `full_below` elements; norms for all."""
    names, norms = [], []
    for k, prm in model.named_parameters():
        g = prm.grad.detach()
        names.append(k); norms.append(float(g.double().norm()))
        if g.numel() < full_below:
            scale = float(g.abs().max()) or 1.0
            out["grad16/" + k] = (g / scale).numpy().astype(np.float16)
            out["gscale/" + k] = np.float32(scale)
    out["grad_norms"] = np.asarray(norms, dtype=np.float64)
    out["grad_names_json"] = np.frombuffer(json.dumps(names).encode(), dtype=np.uint8)


LEGAL = dict(rnn_units=128, kernel_num=[16, 16, 32, 32, 64, 64], length=4000)
C1 = dict(rnn_units=128, kernel_num=[16, 32, 64, 128, 256, 256], length=32000)


def legal_fixture():
    from src.loss import loss_sisdr
    model = _reference_from_seed(LEGAL, seed=21, perturb_seed=22)
    noisy, clean = make_batch(23, 2, 4000)
    model.train()
    est = model(noisy)
    loss = loss_sisdr(est, clean[:, 0])
    loss.backward()
    out = {"est": est.detach().numpy(), "loss": np.float32(loss.item())}
    for k, v in model.state_dict().items():
        if k.endswith(("RMr", "RMi", "RVrr", "RVri", "RVii", "num_batches_tracked")):
            out["state_after/" + k] = v.numpy()
    _pack_grads(model, out, full_below=1 << 30)
    model.eval()
    with torch.no_grad():
        out["est_eval"] = model(noisy).numpy()
    np.savez_compressed(os.path.join(OUT, "dccrn_legal_fwd_bwd.npz"), **out)
    print("legal: loss", loss.item())


def c1_fixture():
    from src.loss import loss_sisdr
    model = _reference_from_seed(C1, seed=10, perturb_seed=11)
    noisy, clean = make_batch(0, 2, 32000)
    model.train()
    est = model(noisy)
    loss = loss_sisdr(est, clean[:, 0])
    loss.backward()
    e = est.detach()
    out = {"est16": e.numpy().astype(np.float16), "est_l2": np.float64(e.double().norm()), "loss": np.float32(loss.item())}
    _pack_grads(model, out, full_below=4096)
    np.savez_compressed(os.path.join(OUT, "dccrn_c1_checksum.npz"), **out)
    print("c1: loss", loss.item(), "est L2", float(e.double().norm()))


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class SummaryWriter:
        log = []

        def __init__(self, *a, **k):
            pass

        def add_text(self, *a, **k):
            pass

        def add_scalar(self, tag, value, step=None, *a, **k):
            SummaryWriter.log.append((tag, float(value), int(step) if step is not None else -1))

        def add_scalars(self, *a, **k):
            pass

        def add_figure(self, *a, **k):
            pass

    mod("tensorboard")
    tb = mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    import torch.utils
    torch.utils.tensorboard = tb
    mod("librosa", display=mod("librosa.display"))
    mod("omegaconf", OmegaConf=object)
    mod("pesq", pesq=None, cypesq=None)
    mod("pypesq", pesq=None)
    mod("pystoi", stoi=None)
    mod("museval", metrics=mod("museval.metrics", bss_eval=None))
    mod("julius", resample_frac=None)
    mod("soundfile")
    mod("torchaudio", transforms=mod("torchaudio.transforms"))
    return SummaryWriter


def solver_fixture():
    import yaml
    writer_cls = install_stubs()
    from src.utils import dict2obj
    from src.distrib import get_model, get_optimizer, get_loss_function
    from src.solver import Solver

    with open(os.path.join(REF, "test", "conf", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    tmp = tempfile.mkdtemp(prefix="sehip_golden_")
    root_file = os.path.join(tmp, "cfg.yaml")
    with open(root_file, "w") as f:
        f.write("dummy: 1\n")
    cfg["root"] = root_file
    cfg["ha"] = None
    cfg["model"].update(name="dccrn", audio_channels=1, num_spk=1, **TINY)
    cfg["optim"].update(loss="si-sdr", clip_grad=5, pit=False)
    cfg["solver"].update(root=tmp, all_steps=True, epochs=1, resume=None, preloaded_model=None)
    config = dict2obj(cfg)

    torch.manual_seed(cfg["seed"])
    model = get_model(config.model)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    optimizer = get_optimizer(config.optim, model)
    loss_fn = get_loss_function(config.optim)
    batches = []
    for s in range(2):
        noisy, clean = make_batch(100 + s, 2, 4000)
        batches.append((noisy, clean, [None], [None], ["x"], [s]))
    val = [batches[0]]
    solver = Solver(config=config, model=model, optimizer=optimizer, loss_function=loss_fn,
                    train_dataloader=batches, validation_dataloader=val, test_dataloader=None,
                    device="cpu")
    solver.train()

    out = {}
    for k, v in state0.items():
        if k.startswith(("stft.", "istft.")):
            continue
        out["state0/" + k] = v.numpy()
    for k, v in model.state_dict().items():
        if k.startswith(("stft.", "istft.")):
            continue
        out["state2/" + k] = v.cpu().numpy()
    names = [n for n, _ in model.named_parameters()]
    opt_state = optimizer.state_dict()["state"]
    for idx, n in enumerate(names):
        out["adam_m/" + n] = opt_state[idx]["exp_avg"].cpu().numpy()
        out["adam_v/" + n] = opt_state[idx]["exp_avg_sq"].cpu().numpy()
    for s, (noisy, clean, *_rest) in enumerate(batches):
        out[f"noisy{s}"] = noisy.numpy()
        out[f"clean{s}"] = clean.numpy()
    log = writer_cls.log
    out["log_json"] = np.frombuffer(json.dumps(log).encode(), dtype=np.uint8)
    ckpt = torch.load(os.path.join(str(solver.checkpoints_dir), "latest_model.tar"),
                      map_location="cpu", weights_only=False)
    out["ckpt_keys_json"] = np.frombuffer(json.dumps(
        {"top": sorted(ckpt.keys()), "model": list(ckpt["model"].keys())}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "dccrn_tiny_solver.npz"), **out)
    print("solver log:", log)


def sisnr_fixture():
    from src.loss import si_snr
    g = torch.Generator().manual_seed(3)
    out = {}
    cases = {
        "a": (torch.randn(3, 1, 257, generator=g), torch.randn(3, 1, 257, generator=g)),
        "b": (0.1 * torch.randn(2, 2, 1, 1000, generator=g), 0.1 * torch.randn(2, 2, 1, 1000, generator=g)),
        "zero_target": (torch.randn(2, 1, 64, generator=g), torch.zeros(2, 1, 64)),
        "equal": (None, torch.randn(2, 1, 64, generator=g)),
    }
    for k, (a, b) in cases.items():
        if a is None:
            a = b.clone()
        out[k + "/est"] = a.numpy()
        out[k + "/ref"] = b.numpy()
        out[k + "/si_snr"] = np.float32(si_snr(a, b).item())
    np.savez_compressed(os.path.join(OUT, "sisnr_cases.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    if "--new" not in sys.argv:   # the round-1 fixtures (unchanged files: regenerate only on purpose)
        fwd_bwd_fixture()
        sisnr_fixture()
    legal_fixture()
    c1_fixture()
    if "--new" not in sys.argv:
        solver_fixture()
