"""GPU parity: STFT/iSTFT FFT kernels, SI-SNR loss and the fused optimizer vs the CPU oracle.
Tolerances are fp32 round-off (these kernels compute in fp32)."""
import math

import numpy as np
import pytest
import torch

from oracle import dccrn_oracle as O
from util import load_golden, rel_err, max_abs

pytestmark = pytest.mark.gpu

WIN, HOP, FFT = 400, 100, 512


@pytest.fixture(scope="module")
def dev():
    from sehip import _lib
    assert torch.cuda.is_available()
    _lib.call("sehip_check_device", 0)
    return torch.device("cuda:0")


# (win_len, win_inc) of the DCCRN constructor (src/model/dccrn.py:14-15): the defaults and three other frame geometries -- half-overlap,
# a window as long as the transform, a short window
GEOMETRIES = [(400, 100), (320, 160), (512, 128), (256, 64)]


@pytest.mark.parametrize("WIN,HOP", GEOMETRIES)
@pytest.mark.parametrize("b,n", [(2, 4000), (3, 32000), (1, 700)])
def test_stft_fwd(dev, b, n, WIN, HOP):
    from sehip import ops
    g = torch.Generator().manual_seed(1)
    wav = torch.randn(b, n, generator=g) * 0.3
    analysis, _, window = O.stft_bases(WIN, FFT)
    ref = O.conv_stft(wav[:, None], analysis, WIN, HOP)  # [B, 514, T]
    spec, enc = ops.stft_fwd(wav.to(dev), window.to(dev), WIN, HOP)
    t = ref.shape[-1]
    assert spec.shape == (b, t, 257, 2)
    ref_c = torch.stack([ref[:, :257], ref[:, 257:]], -1).permute(0, 2, 1, 3)  # [B,T,257,2]
    assert rel_err(spec.cpu(), ref_c) < 2e-6
    assert rel_err(enc.float().cpu(), ref_c[:, :, 1:]) < 4e-3  # bf16 storage


@pytest.mark.parametrize("WIN,HOP", GEOMETRIES)
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("b,n", [(2, 4000), (2, 32000)])
def test_istft_fwd_bwd(dev, mode, b, n, WIN, HOP):
    from sehip import ops
    g = torch.Generator().manual_seed(2)
    wav = torch.randn(b, n, generator=g) * 0.3
    analysis, synthesis, window = O.stft_bases(WIN, FFT)
    spec_ref = O.conv_stft(wav[:, None], analysis, WIN, HOP)
    t = spec_ref.shape[-1]
    mask = (torch.randn(b, 2, 256, t, generator=g) * 0.7).requires_grad_(True)
    # oracle: same formulas as dccrn_forward's tail
    real, imag = spec_ref[:, :257], spec_ref[:, 257:]
    m_r = torch.nn.functional.pad(mask[:, 0], [0, 0, 1, 0])
    m_i = torch.nn.functional.pad(mask[:, 1], [0, 0, 1, 0])
    if mode == 0:
        mags = torch.sqrt(real ** 2 + imag ** 2 + 1e-8)
        phase = torch.atan2(imag, real)
        m_mag = (m_r ** 2 + m_i ** 2) ** 0.5
        m_phase = torch.atan2(m_i / (m_mag + 1e-8), m_r / (m_mag + 1e-8))
        est_mag = torch.tanh(m_mag) * mags
        er, ei = est_mag * torch.cos(phase + m_phase), est_mag * torch.sin(phase + m_phase)
    elif mode == 1:
        er, ei = real * m_r - imag * m_i, real * m_i + imag * m_r
    else:
        er, ei = real * m_r, imag * m_i
    out_ref = torch.clamp(O.conv_istft(torch.cat([er, ei], 1), synthesis, window, WIN, HOP, n), -1, 1)
    gout = torch.randn(out_ref.shape, generator=g)
    (gmask_ref,) = torch.autograd.grad((out_ref * gout).sum(), mask)

    spec, _ = ops.stft_fwd(wav.to(dev), window.to(dev), WIN, HOP)
    inv_coff = torch.from_numpy(ops.inv_window_energy(WIN, HOP, t, n)).to(dev)
    mask_cl = mask.detach().permute(0, 3, 2, 1).contiguous().to(dev)  # [B,T,256,2]
    out = ops.istft_fwd(spec, mask_cl, window.to(dev), inv_coff, WIN, HOP, n, mode)
    assert rel_err(out.cpu(), out_ref[:, 0]) < 1e-5
    dmask = ops.istft_bwd(gout[:, 0].contiguous().to(dev), out, spec, mask_cl, window.to(dev), inv_coff, WIN, HOP, n, mode)
    gm = dmask.float().cpu().permute(0, 3, 2, 1)
    assert rel_err(gm, gmask_ref) < 6e-3  # bf16 output


def test_sisnr(dev):
    from sehip import ops
    g = load_golden("sisnr_cases.npz")
    for k in ("a", "b", "zero_target", "equal"):
        est = torch.from_numpy(g[k + "/est"]).float()
        ref = torch.from_numpy(g[k + "/ref"]).float()
        est2 = est.reshape(-1, est.shape[-1]).contiguous()
        ref2 = ref.reshape(-1, ref.shape[-1]).contiguous()
        loss, rowstat = ops.sisnr_fwd(est2.to(dev), ref2.to(dev))
        want = -float(g[k + "/si_snr"])
        assert abs(float(loss) - want) < 2e-4 * max(1.0, abs(want)), k
        if k in ("a", "b"):
            e = est2.clone().requires_grad_(True)
            O.loss_sisdr(e, ref2).backward()
            d = ops.sisnr_bwd(est2.to(dev), ref2.to(dev), rowstat)
            assert rel_err(d.cpu(), e.grad) < 2e-4, k


def test_clip_adam_matches_oracle(dev):
    from sehip import _lib
    from sehip._lib import call, ptr, stream
    g = torch.Generator().manual_seed(3)
    sizes = [7, 1024, 3, 50000, 1]
    n = sum(sizes)
    p0 = torch.randn(n, generator=g)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    params = {str(i): p0[offs[i]:offs[i + 1]].clone() for i in range(len(sizes))}
    adam = O.AdamState(params, lr=3e-4)
    p = p0.clone().to(dev)
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
    tsum = torch.zeros(len(sizes), device=dev); metric = torch.zeros(2, device=dev)
    offs_d = torch.from_numpy(offs).to(dev)
    for step in range(1, 4):
        grads = torch.randn(n, generator=g) * (10.0 if step == 2 else 0.01)
        gd = {str(i): grads[offs[i]:offs[i + 1]].clone() for i in range(len(sizes))}
        total = O.clip_grad_norm(gd, 5.0)
        O.adam_update(params, gd, adam)
        want_metric = math.sqrt(sum(float(x.sum()) ** 2 for x in gd.values()))
        gdev = grads.to(dev)
        call("sehip_grad_sumsq", ptr(gdev), n, ptr(sumsq), stream())
        call("sehip_opt_step", ptr(p), ptr(gdev), ptr(m), ptr(v), n, ptr(sumsq), 5.0, 3e-4, 0.9, 0.999, 1e-8, step, None,
             0.0, 0, 1.0, stream())
        call("sehip_grad_metric", ptr(gdev), ptr(offs_d), len(sizes), max(sizes), ptr(sumsq), ptr(tsum), ptr(metric), stream())
        assert abs(float(metric[1]) - float(total)) < 1e-4 * float(total)
        assert abs(float(metric[0]) - want_metric) < 1e-3 * max(1.0, want_metric)
        ref_p = torch.cat([params[str(i)] for i in range(len(sizes))])
        assert max_abs(p.cpu(), ref_p) < 2e-6


def test_opt_step_grad_scale_is_the_data_parallel_mean(dev):
    """grad_scale = 1/world on the SUMMED gradients == the step on the mean gradient (clipping included)."""
    from sehip._lib import call, ptr, stream
    g = torch.Generator().manual_seed(17)
    n, world = 30011, 4
    p0 = torch.randn(n, generator=g)
    gsum = torch.randn(n, generator=g) * 0.3           # norm of the mean ~ 13: clipping at 5 is active
    res = []
    for grads, scale in ((gsum.clone(), 1.0 / world), (gsum / world, 1.0)):
        p = p0.clone().to(dev); gd = grads.to(dev)
        m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
        sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        call("sehip_grad_sumsq", ptr(gd), n, ptr(sumsq), stream())
        call("sehip_opt_step", ptr(p), ptr(gd), ptr(m), ptr(v), n, ptr(sumsq), 5.0, 3e-4, 0.9, 0.999, 1e-8, 1, None, 0.0, 0, scale,
             stream())
        res.append((p.cpu(), gd.cpu()))
    assert max_abs(res[0][0], res[1][0]) < 1e-6 and max_abs(res[0][1], res[1][1]) < 1e-6
    assert abs(float(res[0][1].norm()) - 5.0) < 1e-3  # the written-back gradient is the clipped mean


@pytest.mark.parametrize("momentum", [0.0, 0.9])
def test_clip_sgd_matches_torch(dev, momentum):
    """optim.hip mode 1 against torch.optim.SGD (what src/distrib.py:246-250 builds), with and without momentum, with the
    reference's clip-then-step order (src/solver.py:487-492)."""
    from sehip._lib import call, ptr, stream
    g = torch.Generator().manual_seed(13)
    n = 40007
    p_ref = torch.nn.Parameter(torch.randn(n, generator=g))
    opt = torch.optim.SGD([p_ref], lr=0.05, momentum=momentum)
    p = p_ref.detach().clone().to(dev)
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
    for step in range(1, 5):
        grads = torch.randn(n, generator=g) * (1.0 if step == 2 else 0.01)
        p_ref.grad = grads.clone()
        torch.nn.utils.clip_grad_norm_([p_ref], 5.0)
        opt.step()
        gdev = grads.to(dev)
        call("sehip_grad_sumsq", ptr(gdev), n, ptr(sumsq), stream())
        call("sehip_opt_step", ptr(p), ptr(gdev), ptr(m), ptr(v), n, ptr(sumsq), 5.0, 0.05, momentum, 0.0, 1e-8, step, None,
             0.0, 1, 1.0, stream())
        assert max_abs(gdev.cpu(), p_ref.grad) < 1e-6          # the clipped gradient is written back like p.grad
        assert max_abs(p.cpu(), p_ref.detach()) < 2e-6, step
    if momentum:
        assert max_abs(m.cpu(), opt.state[p_ref]["momentum_buffer"]) < 2e-6


@pytest.mark.parametrize("name", ["l1", "mse"])
def test_l1_mse_losses(dev, name):
    from sehip import distrib, utils
    fn = distrib.get_loss_function(utils.dict2obj({"loss": name}))
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 1, 5000, generator=g)
    y = torch.randn(3, 1, 5000, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = (torch.nn.functional.l1_loss if name == "l1" else torch.nn.functional.mse_loss)(xr, y)
    ref.backward()
    xd = x.to(dev).requires_grad_(True)
    out = fn(xd, y.to(dev))
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-6
