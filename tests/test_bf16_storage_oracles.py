"""VERDICT r3 item 4a / ADVICE: the DCUnet and ConvTasNet gradient tests compare the bf16 HIP path with the fp32 oracle under a random
fixed upstream gradient and see 13 % / 28 % (DCUnet-10 / -20) and 5 % (ConvTasNet) -- explained in DESIGN section 2 as sign flips of
near-zero pre-activations of the non-smooth layers (LeakyReLU(0.01), PReLU, ReLU) under a 1 % bf16 forward error, but until round 4
only ARGUED.  This file MEASURES it on the CPU with no HIP code involved: the oracle evaluated with bf16 round-trips at the storage
points of the HIP path (oracle Bf16Sim: activations, activation gradients, MFMA weight operands) deviates from the plain fp32 oracle
by the same 13 % / 28 % / 5 %.  So those figures are a property of bf16 storage in these networks (reference math:
src/model/dcunet.py:8-50, src/model/conv_tasnet.py:307-349), and the GPU tests can gate the HIP path against the bf16-storage oracle
instead (tests/test_gpu_dcunet.py, tests/test_gpu_convtasnet.py)."""
import pytest
import torch

from oracle import convtasnet_oracle as CT
from oracle import dcunet_oracle as D
from oracle.dccrn_oracle import Bf16Sim, NoSim


def _global_rel(a, b):
    num = sum(float(((x.double() - y.double()) ** 2).sum()) for x, y in zip(a, b))
    den = sum(float((y.double() ** 2).sum()) for y in b)
    return (num / den) ** 0.5


def dcunet_problem(depth, frames, batch):
    """Exactly the problem of tests/test_gpu_dcunet.py::test_full_width_gradients_vs_oracle (same seeds, same model)."""
    from sehip.model import DCUnet
    torch.manual_seed(11)
    model = DCUnet(data_type=True, model_complexity=45, model_depth=depth)
    p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("encoders.", "decoders."))}
    g = torch.Generator().manual_seed(12)
    x = 0.5 * torch.randn(batch, 1, 257, frames, 2, generator=g)
    names = sorted(k for k in p if D.is_trainable(k))
    return model, p, x, names, g


def dcunet_oracle_run(p, x, names, depth, sim, G=None, gen=None):
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    ref = D.dcunet_forward(work, x, model_complexity=45, model_depth=depth, training=True, sim=sim)
    if G is None:
        G = torch.randn(ref.shape, generator=gen) / ref.numel() ** 0.5
    return ref.detach(), torch.autograd.grad((ref * G).sum(), [leaves[k] for k in names]), G


@pytest.mark.parametrize("depth,frames,batch,lo,hi", [(10, 65, 2, 0.09, 0.18), (20, 257, 1, 0.20, 0.36)])
def test_dcunet_bf16_storage_alone_moves_the_gradients_by_what_the_hip_path_shows(depth, frames, batch, lo, hi):
    _, p, x, names, g = dcunet_problem(depth, frames, batch)
    ref, g0, G = dcunet_oracle_run(p, x, names, depth, NoSim, gen=g)
    ref_s, g1, _ = dcunet_oracle_run(p, x, names, depth, Bf16Sim, G=G)
    out = float((ref_s - ref).norm() / ref.norm())
    dev = _global_rel(g1, g0)
    print(f"DCUnet-{depth}: bf16-storage oracle vs fp32 oracle: output {out:.3e}, global gradient {dev:.3e}")
    # measured 1.33e-1 (HIP path vs fp32 oracle: 1.33e-1) / 2.80e-1 (HIP: 2.77e-1); outputs 9.9e-3 / 1.75e-2 (HIP: 9.9e-3 / 1.75e-2)
    assert lo < dev < hi
    assert out < (1.5e-2 if depth == 10 else 2.5e-2)


def convtasnet_problem():
    """Exactly the problem of tests/test_gpu_convtasnet.py::test_full_width_model_vs_oracle."""
    from sehip.model import ConvTasNet
    torch.manual_seed(5)
    model = ConvTasNet(sources=["None", "None"], audio_channels=1)
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    mix = 0.3 * torch.randn(2, 1, 8000, generator=g)
    return model, p, mix, sorted(p), g


def test_convtasnet_bf16_storage_alone_moves_the_gradients_by_what_the_hip_path_shows():
    _, p, mix, names, g = convtasnet_problem()
    lv = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = CT.convtasnet_forward(lv, mix, audio_channels=1)
    _tgt_noise = torch.randn(ref.shape, generator=g)      # (the GPU test draws its target noise here: keep the generator in step)
    G = torch.randn(ref.shape, generator=g) / ref.numel() ** 0.5
    g0 = torch.autograd.grad((ref * G).sum(), [lv[k] for k in names])
    lv2 = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref_s = CT.convtasnet_forward(lv2, mix, audio_channels=1, sim=Bf16Sim)
    g1 = torch.autograd.grad((ref_s * G).sum(), [lv2[k] for k in names])
    out = float((ref_s.detach() - ref.detach()).norm() / ref.detach().norm())
    dev = _global_rel(g1, g0)
    print(f"ConvTasNet: bf16-storage oracle vs fp32 oracle: output {out:.3e}, global gradient {dev:.3e}")
    assert 0.03 < dev < 0.08 and out < 1.5e-2        # measured 5.1e-2 (HIP path vs fp32 oracle: 5.1e-2), output 7.2e-3
