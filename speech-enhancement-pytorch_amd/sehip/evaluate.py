"""stft_custom / istft_custom behind the reference's interface (reference: src/evaluate.py:101-162).

Same names, arguments and shapes as the reference functions the Solver calls twice per step for the STFT-domain models
(src/solver.py:457-458): ``config`` is any object with ``n_fft, hop_length, win_length, center``; 3-D ``[B, C, N]`` and
4-D ``[B, S, C, N]`` inputs give ``[B, C, F, T, 2]`` / ``[B, S, C, F, T, 2]``.  The arithmetic is the HIP kernels of
csrc/stft_custom.hip through the C ABI; there is no CPU or torch.stft fallback (a CPU tensor raises SehipError).
``evaluate()`` (the chunked inference of src/evaluate.py:10-98) runs ON THE DEVICE here: segmentation is a strided view,
the stitch one gather, the STFT-domain branch the HIP kernels above -- the reference builds segments and output with Python
loops on the CPU and moves each half batch to the GPU.
"""
import torch

from ._lib import SehipError, call, lib, ptr, stream
from .model.types import MONARCH_SPEECH_SEPARTAION_MODELS, MULTI_SPEECH_SEPERATION_MODELS, STFT_MODELS


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


def _prepare_input_wav_zero_filled(wav, num_feature, stride):
    """[..., N] -> [num_segment, ..., num_feature]: windows of num_feature samples every `stride`, the tail zero-padded to a whole
    number of strides (src/evaluate.py:164-183) -- a strided view instead of a copy loop."""
    assert wav.shape[-1] >= num_feature, "the length of data is too short comparing the number of features..."
    extra = (wav.shape[-1] - num_feature) % stride
    if extra:
        wav = torch.nn.functional.pad(wav, (0, stride - extra))
    seg = wav.unfold(-1, num_feature, stride)                    # [..., num_segment, num_feature]
    return seg.movedim(-2, 0)


def evaluate(mixture, model, device, config, max_segments_per_call=None):
    """Same contract as the reference's evaluate() (src/evaluate.py:10-98): mixture [B, C, N] -> enhanced [B, C, N] (or
    [B, S, C, N] for the multi-speaker models), with config.dset.norm in {"z-score", "linear-scale", None},
    config.model.{name, segment, win_length, ...} and config.dset.sample_rate.  Everything stays on `device`; the segments go
    through the model in chunks of `max_segments_per_call` (default: two halves, like the reference)."""
    with torch.no_grad():
        x = mixture.to(device)
        norm = getattr(config.dset, "norm", None)
        if norm == "z-score":
            mean, std = torch.mean(x, dim=-1, keepdim=True), torch.std(x, dim=-1, keepdim=True)
            x = (x - mean) / (std + 1e-9)
        elif norm == "linear-scale":
            # (the reference subtracts torch.max's (values, indices) tuple here and raises; the evident intent is implemented)
            hi, lo = torch.amax(x, dim=-1, keepdim=True), torch.amin(x, dim=-1, keepdim=True)
            x = (x - lo) / (hi - lo + 1e-9)
        stride = int(config.model.win_length)
        num_feature = int(config.dset.sample_rate * config.model.segment)
        seg = _prepare_input_wav_zero_filled(x, num_feature, stride)          # [S, B, C, F] (view)
        num_segment, nbatch, nchannel, nsample = seg.shape
        batch = seg.reshape(num_segment * nbatch, nchannel, nsample).contiguous()
        name = config.model.name
        if name in STFT_MODELS:
            batch = stft_custom(batch, config.model)
        if model is not None:
            model.eval()
            step = max_segments_per_call or max(1, batch.shape[0] // 2)
            first = batch.shape[0] // 2 if not max_segments_per_call else step
            cuts = [0, first] + list(range(first + step, batch.shape[0], step)) + [batch.shape[0]] if first else [0, batch.shape[0]]
            output = torch.cat([model(batch[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a], dim=0)
        else:
            output = batch
        if name in MONARCH_SPEECH_SEPARTAION_MODELS:
            output = torch.unsqueeze(output, dim=1)
        if name in STFT_MODELS:
            output = istft_custom(output.contiguous(), nsample, config.model)
        if model is not None and name in MULTI_SPEECH_SEPERATION_MODELS:
            nsrc = len(config.model.sources)
            output = output.reshape(num_segment, nbatch, nsrc, nchannel, nsample)
        else:
            output = output.reshape(num_segment, nbatch, nchannel, nsample)
        # stitch: the first segment whole, then the last `stride` samples of every further segment
        tail = output[1:, ..., nsample - stride:]                             # [S-1, ..., stride]
        tail = tail.movedim(0, -2).reshape(*output.shape[1:-1], (num_segment - 1) * stride)
        enhanced = torch.cat([output[0], tail], dim=-1)[..., :mixture.shape[-1]]
        if model is not None and name in MULTI_SPEECH_SEPERATION_MODELS:
            mean_, std_ = (mean.unsqueeze(1), std.unsqueeze(1)) if norm == "z-score" else (None, None)
        else:
            mean_, std_ = (mean, std) if norm == "z-score" else (None, None)
        if norm == "z-score":
            enhanced = enhanced * (std_ + 1e-9) + mean_
        elif norm == "linear-scale":
            enhanced = enhanced * (hi - lo + 1e-9) + lo
    health = getattr(model, "check_health", None)
    if health is not None and health():
        # a hand-off time-out of Demucs' persistent LSTM kernels invalidated this output: the model has switched to the per-step
        # launches, run the utterance again
        return evaluate(mixture, model, device, config, max_segments_per_call)
    return enhanced
