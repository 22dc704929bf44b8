"""CPU oracle for stft_custom / istft_custom -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's STFT front-end for the STFT-domain models
(reference: src/evaluate.py:101-128 stft_custom, :130-162 istft_custom, called 2x per train step
from src/solver.py:457-458).  The reference calls torch.stft / torch.istft; this file restates what
those calls compute for the arguments the reference passes -- periodic hann window of win_length
zero-padded (centred) to n_fft, center=True -> reflect padding by n_fft//2, one-sided spectrum,
normalized=False, then the reference's own division / multiplication by win_length -- with the FFT
itself taken from numpy.  Only tests/ (and bench-style timing scripts under tools/) may import it.

Parity pinning: tests/test_oracle_golden.py::test_stft_custom_* check both functions against
tests/golden/stft_custom.npz, produced by oracle/gen_golden_stft.py, which imports the real
reference in the build container.
"""
from __future__ import annotations

import numpy as np


def hann_periodic(win_length: int) -> np.ndarray:
    """torch.hann_window(win_length) (periodic=True): 0.5 - 0.5 cos(2 pi n / win_length)."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def padded_window(n_fft: int, win_length: int) -> np.ndarray:
    """torch.stft pads a shorter window with zeros on both sides to n_fft (left = (n_fft - win_length) // 2)."""
    w = np.zeros(n_fft, dtype=np.float64)
    left = (n_fft - win_length) // 2
    w[left:left + win_length] = hann_periodic(win_length)
    return w


def n_frames(n_samples: int, n_fft: int, hop: int, center: bool) -> int:
    return 1 + (n_samples // hop if center else (n_samples - n_fft) // hop)


def stft_custom(x: np.ndarray, n_fft: int, hop_length: int, win_length: int, center: bool = True) -> np.ndarray:
    """x [..., N] float -> [..., n_fft//2+1, T, 2] (src/evaluate.py:101-128): leading dims are kept, the last one is
    framed; the spectrum is divided by win_length (:118)."""
    x = np.asarray(x)
    lead, n = x.shape[:-1], x.shape[-1]
    rows = x.reshape(-1, n).astype(np.float64)
    if center:
        p = n_fft // 2
        rows = np.pad(rows, ((0, 0), (p, p)), mode="reflect")
    t = 1 + (rows.shape[1] - n_fft) // hop_length
    w = padded_window(n_fft, win_length)
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(t)[:, None]          # [T, n_fft]
    frames = rows[:, idx] * w                                                      # [R, T, n_fft]
    spec = np.fft.rfft(frames, axis=-1) / win_length                               # [R, T, F]
    out = np.stack([spec.real, spec.imag], axis=-1).transpose(0, 2, 1, 3)          # [R, F, T, 2]
    return out.reshape(*lead, n_fft // 2 + 1, t, 2).astype(np.float32)


def istft_custom(spec: np.ndarray, length: int, n_fft: int, hop_length: int, win_length: int, center: bool = True) -> np.ndarray:
    """spec [..., F, T, 2] -> [..., length] (src/evaluate.py:130-162): multiply by win_length (:131), inverse one-sided
    FFT of every frame (the imaginary parts of DC and Nyquist are ignored, as by any c2r transform), window, overlap-add,
    divide by the overlap-added squared window, drop the leading n_fft//2 centre padding, keep `length` samples (zero-filled
    beyond the overlap-added signal)."""
    spec = np.asarray(spec)
    lead = spec.shape[:-3]
    f, t = spec.shape[-3], spec.shape[-2]
    assert f == n_fft // 2 + 1
    z = (spec[..., 0].astype(np.float64) + 1j * spec[..., 1].astype(np.float64)) * win_length
    z = z.reshape(-1, f, t).transpose(0, 2, 1)                                     # [R, T, F]
    w = padded_window(n_fft, win_length)
    frames = np.fft.irfft(z, n=n_fft, axis=-1) * w                                 # [R, T, n_fft]
    total = n_fft + hop_length * (t - 1)
    y = np.zeros((z.shape[0], total))
    env = np.zeros(total)
    for i in range(t):
        y[:, i * hop_length:i * hop_length + n_fft] += frames[:, i]
        env[i * hop_length:i * hop_length + n_fft] += w * w
    start = n_fft // 2 if center else 0
    # with an explicit length torch.istft keeps [start, start + length) -- it does NOT drop the trailing centre padding then
    # (only the part beyond the overlap-added signal is zero-filled); without one it drops n_fft//2 at both ends
    end = min(total, start + length) if length is not None else total - start
    y, env = y[:, start:end], env[start:end]
    assert env.min() > 1e-11, "window overlap-add is zero somewhere (torch.istft raises here as well)"
    y = y / env
    if length is not None and y.shape[1] < length:
        y = np.pad(y, ((0, 0), (0, length - y.shape[1])))
    return y.reshape(*lead, y.shape[1]).astype(np.float32)
