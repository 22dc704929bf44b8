"""stft_custom / istft_custom behind the reference's interface (reference: src/evaluate.py:101-162).

Same names, arguments and shapes as the reference functions the Solver calls twice per step for the STFT-domain models
(src/solver.py:457-458): ``config`` is any object with ``n_fft, hop_length, win_length, center``; 3-D ``[B, C, N]`` and
4-D ``[B, S, C, N]`` inputs give ``[B, C, F, T, 2]`` / ``[B, S, C, F, T, 2]``.  The arithmetic is the HIP kernels of
csrc/stft_custom.hip through the C ABI; there is no CPU or torch.stft fallback (a CPU tensor raises SehipError).
``evaluate()`` itself (chunked inference) is out of scope.
"""
import torch

from ._lib import SehipError, call, lib, ptr, stream


def _cfg(config):
    return int(config.n_fft), int(config.hop_length), int(config.win_length), 1 if bool(config.center) else 0


def _need_gpu(t, who):
    if not t.is_cuda:
        raise SehipError(f"{who}: the tensor is on {t.device}; sehip has no CPU path")
    if t.dtype != torch.float32:
        raise SehipError(f"{who}: float32 expected, got {t.dtype}")


def stft_custom(tensor: torch.Tensor, config):
    _need_gpu(tensor, "stft_custom")
    if tensor.dim() not in (3, 4):
        raise SehipError(f"stft_custom: [B, C, N] or [B, S, C, N] expected, got {tuple(tensor.shape)}")
    n_fft, hop, win, center = _cfg(config)
    lead, n = tuple(tensor.shape[:-1]), tensor.shape[-1]
    x = tensor.contiguous().view(-1, n)
    t = lib().sehip_stft_custom_frames(n, n_fft, hop, center)
    out = torch.empty(x.shape[0], n_fft // 2 + 1, max(t, 0), 2, dtype=torch.float32, device=tensor.device)
    call("sehip_stft_custom_fwd", ptr(x), x.shape[0], n, n_fft, hop, win, center, ptr(out), stream())
    return out.view(*lead, n_fft // 2 + 1, t, 2)


def istft_custom(tensor: torch.Tensor, length, config):
    _need_gpu(tensor, "istft_custom")
    if tensor.dim() not in (5, 6) or tensor.shape[-1] != 2:
        raise SehipError(f"istft_custom: [B, C, F, T, 2] or [B, S, C, F, T, 2] expected, got {tuple(tensor.shape)}")
    n_fft, hop, win, center = _cfg(config)
    lead = tuple(tensor.shape[:-3])
    f, t = tensor.shape[-3], tensor.shape[-2]
    if f != n_fft // 2 + 1:
        raise SehipError(f"istft_custom: {f} frequency bins do not match n_fft {n_fft}")
    z = tensor.contiguous().view(-1, f, t, 2)
    if length is None:  # torch.istft without a length: the centre padding is dropped at both ends
        length = n_fft + hop * (t - 1) - (n_fft if center else 0)
    frames = torch.empty(z.shape[0], t, n_fft, dtype=torch.float32, device=tensor.device)
    wav = torch.empty(z.shape[0], int(length), dtype=torch.float32, device=tensor.device)
    call("sehip_istft_custom_fwd", ptr(z), z.shape[0], t, n_fft, hop, win, center, int(length), ptr(frames), ptr(wav), stream())
    return wav.view(*lead, int(length))
