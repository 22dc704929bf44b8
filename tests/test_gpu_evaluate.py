"""GPU: on-device evaluate() against vectors of the imported reference's evaluate()
(tests/golden/evaluate_cases.npz, oracle/gen_golden_evaluate.py): normalisation, segmentation, STFT round trip through the
HIP stft_custom / istft_custom kernels, stitching, de-normalisation, the two-speaker layout."""
import types

import pytest
import torch

from util import load_golden, rel_err, max_abs

pytestmark = pytest.mark.gpu


class Toy(torch.nn.Module):
    def __init__(self, nspk=0):
        super().__init__()
        self.nspk = nspk

    def forward(self, x):
        y = 0.5 * x + 0.1 * x * x
        return torch.stack([y, -y], dim=1) if self.nspk == 2 else y


def cfg(name, norm, segment, sources=None):
    m = types.SimpleNamespace(name=name, win_length=512, n_fft=512, hop_length=128, center=True, segment=segment, sources=sources)
    return types.SimpleNamespace(model=m, dset=types.SimpleNamespace(norm=norm, sample_rate=16000))


CASES = {"dccrn_z": ("dccrn", "z-score", 0.25, 0), "dccrn_none_exact": ("dccrn", None, 0.25, 0),
         "dcunet_z": ("dcunet", "z-score", 0.256, 0), "convtasnet_2spk": ("conv-tasnet", "z-score", 0.25, 2)}


@pytest.mark.parametrize("key", sorted(CASES))
def test_evaluate_matches_reference(key):
    from sehip.evaluate import evaluate
    g = load_golden("evaluate_cases.npz")
    name, norm, seg, spk = CASES[key]
    x = torch.from_numpy(g[key + ".x"])
    y = evaluate(x, Toy(spk).cuda(), torch.device("cuda:0"), cfg(name, norm, seg, ["None"] * spk if spk else None))
    ref = torch.from_numpy(g[key + ".y"])
    assert y.is_cuda and tuple(y.shape) == tuple(ref.shape)
    assert max_abs(y.cpu(), ref) < 2e-5 * float(ref.abs().max()) + 1e-6, key
    # chunking the segments differently does not change the result
    y2 = evaluate(x, Toy(spk).cuda(), torch.device("cuda:0"), cfg(name, norm, seg, ["None"] * spk if spk else None), max_segments_per_call=3)
    assert max_abs(y2, y) < 1e-6


def test_evaluate_with_the_hip_dccrn():
    """The real model on the chunked path: evaluate() == the model applied to each segment + the stitch rule."""
    from sehip.evaluate import evaluate
    from sehip.model import DCCRN
    torch.manual_seed(3)
    model = DCCRN(kernel_num=[16, 16, 32, 32, 64, 64], length=4000).cuda().eval()
    x = 0.1 * torch.randn(2, 1, 4000 + 2 * 512 + 9, generator=torch.Generator().manual_seed(1))
    c = cfg("dccrn", "z-score", 0.25)
    y = evaluate(x, model, torch.device("cuda:0"), c)
    xn = ((x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + 1e-9)).cuda()
    xp = torch.nn.functional.pad(xn, (0, 512 - 9))
    with torch.no_grad():
        segs = [model(xp[..., k * 512:k * 512 + 4000]) for k in range(4)]
    want = torch.cat([segs[0]] + [s[..., -512:] for s in segs[1:]], -1)[..., :x.shape[-1]]
    want = want * (x.std(-1, keepdim=True).cuda() + 1e-9) + x.mean(-1, keepdim=True).cuda()
    assert rel_err(y.cpu(), want.cpu()) < 1e-5


def test_si_sdr_metric_on_device():
    """sehip.metric.SI_SDR against the formula of src/metric.py:92-123 restated in numpy float32 (the reference module itself
    cannot be imported here: it needs pesq / pystoi / museval at import time, so this metric's parity is restatement-only)."""
    import numpy as np
    from sehip.metric import SI_SDR

    def ref_si_sdr(reference, estimation):
        eps = np.finfo(np.float32).eps
        energy = np.sum(reference ** 2, axis=-1, keepdims=True)
        scale = np.sum(estimation * reference, axis=-1, keepdims=True) / (energy + eps)
        proj = scale * reference
        noise = estimation - proj
        ratio = np.mean(np.sum(proj ** 2, axis=-1) / (np.sum(noise ** 2, axis=-1) + eps))
        return 10 * np.log10(ratio + eps)

    g = torch.Generator().manual_seed(5)
    for shape in ((3, 1, 16000), (2, 2, 1, 4001), (1, 8)):
        clean = torch.randn(shape, generator=g)
        est = clean + 0.3 * torch.randn(shape, generator=g)
        got = float(SI_SDR(clean.cuda(), est.cuda()))
        want = float(ref_si_sdr(clean.numpy(), est.numpy()))
        assert abs(got - want) < 2e-3, (shape, got, want)
    z = torch.zeros(2, 100)
    assert abs(float(SI_SDR(z.cuda(), z.cuda())) - float(ref_si_sdr(z.numpy(), z.numpy()))) < 1e-3   # silent rows: log10(eps)
