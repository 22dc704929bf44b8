"""Thin tensor-level wrappers over the C ABI (one function per exported op).

These do shape bookkeeping and output allocation only; all arithmetic happens in libsehip.
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream, require_gpu

BF16 = torch.bfloat16


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) (src/model/dccrn.py:653)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)


def window_of(win_type, n):
    """The window of init_kernels (src/model/dccrn.py:650-653): ones for None / 'None', else
    scipy.signal.get_window(win_type, n, fftbins=True).  'hann' / 'hamming' in closed form, anything else through scipy."""
    if win_type is None or win_type == "None":
        return np.ones(n, dtype=np.float32)
    if win_type == "hann":
        return hann_periodic(n)
    k = np.arange(n, dtype=np.float64)
    if win_type == "hamming":
        return (0.54 - 0.46 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)
    try:
        from scipy.signal import get_window
        return np.asarray(get_window(win_type, n, fftbins=True), dtype=np.float64).astype(np.float32)
    except Exception as e:      # an unknown name, or scipy absent
        raise _lib.SehipError(f"sehip DCCRN: win_type={win_type!r}: {e}")


def inv_window_energy(win, hop, frames, length, win_type="hann"):
    """1 / (overlap-added window^2 + 1e-8), trimmed like src/model/dccrn.py:733-745."""
    w2 = window_of(win_type, win).astype(np.float64) ** 2
    total = (frames - 1) * hop + win
    e = np.zeros(total, dtype=np.float64)
    for t in range(frames):
        e[t * hop:t * hop + win] += w2
    e = e.astype(np.float32) + np.float32(1e-8)
    pad = win - hop
    return (np.float32(1.0) / e[pad:pad + length]).astype(np.float32)


def stft_frames(n, win, hop):
    return (n + 2 * (win - hop) - win) // hop + 1


def stft_fwd(wav, window, win, hop, fft=512):
    """wav [B,N] fp32 -> (spec [B,T,257,2] fp32, enc_in [B,T,256,2] bf16)."""
    require_gpu(wav, "stft_fwd")
    b, n = wav.shape
    t = stft_frames(n, win, hop)
    spec = torch.empty(b, t, 257, 2, device=wav.device, dtype=torch.float32)
    enc = torch.empty(b, t, 256, 2, device=wav.device, dtype=BF16)
    call("sehip_stft_fwd", ptr(wav), ptr(window), b, n, win, hop, fft, ptr(spec), ptr(enc), stream())
    return spec, enc


def istft_fwd(spec, mask, window, inv_coff, win, hop, length, mode=0, fft=512):
    """spec [B,T,257,2], mask [B,T,256,2] fp32 -> wav [B,length] (clamped to [-1,1])."""
    require_gpu(spec, "istft_fwd")
    b, t = spec.shape[:2]
    frames = torch.empty(b, t, win, device=spec.device, dtype=torch.float32)
    wav = torch.empty(b, length, device=spec.device, dtype=torch.float32)
    call("sehip_istft_fwd", ptr(spec), ptr(mask), ptr(window), ptr(inv_coff), b, t, win, hop, fft, length, mode,
         ptr(frames), ptr(wav), stream())
    return wav


def istft_bwd(dwav, wav, spec, mask, window, inv_coff, win, hop, length, mode=0, fft=512):
    """-> d mask [B,T,256,2] bf16."""
    b, t = spec.shape[:2]
    dmask = torch.empty(b, t, 256, 2, device=spec.device, dtype=BF16)
    call("sehip_istft_bwd", ptr(dwav), ptr(wav), ptr(spec), ptr(mask), ptr(window), ptr(inv_coff), b, t, win, hop, fft,
         length, mode, ptr(dmask), stream())
    return dmask


def sisnr_fwd(est, ref):
    """est/ref [R,N] fp32 -> (loss scalar tensor = -mean si_snr, rowstat [R,4])."""
    require_gpu(est, "sisnr_fwd")
    r, n = est.shape
    rowstat = torch.empty(r, 4, device=est.device, dtype=torch.float32)
    loss = torch.empty(1, device=est.device, dtype=torch.float32)
    call("sehip_sisnr_fwd", ptr(est), ptr(ref), r, n, ptr(rowstat), ptr(loss), stream())
    return loss, rowstat


def sisnr_bwd(est, ref, rowstat, upstream=None):
    r, n = est.shape
    dest = torch.empty_like(est)
    call("sehip_sisnr_bwd", ptr(est), ptr(ref), ptr(rowstat), ptr(upstream), r, n, ptr(dest), stream())
    return dest


def sisnr_pit_fwd(est, ref):
    """est/ref [B,S,C,N] fp32 contiguous -> (loss [1], rowstat [S*S,B*C,4], pairloss [S*S], perm [S] int32): the
    permutation-invariant SI-SNR of src/loss.py:58-100 (batch-level permutation, first minimum in itertools order)."""
    require_gpu(est, "sisnr_pit_fwd")
    b, s, c, n = est.shape
    rowstat = torch.empty(s * s, b * c, 4, device=est.device, dtype=torch.float32)
    pairloss = torch.empty(s * s, device=est.device, dtype=torch.float32)
    perm = torch.empty(s, device=est.device, dtype=torch.int32)
    loss = torch.empty(1, device=est.device, dtype=torch.float32)
    call("sehip_sisnr_pit_fwd", ptr(est), ptr(ref), b, s, c, n, ptr(rowstat), ptr(pairloss), ptr(perm), ptr(loss), stream())
    return loss, rowstat, pairloss, perm


def sisnr_pit_bwd(est, ref, rowstat, perm, upstream=None):
    b, s, c, n = est.shape
    dest = torch.empty_like(est)
    call("sehip_sisnr_pit_bwd", ptr(est), ptr(ref), ptr(rowstat), ptr(perm), ptr(upstream), b, s, c, n, ptr(dest), stream())
    return dest
