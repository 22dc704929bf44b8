"""The explicit CPU plumbing configuration (BASELINE config C0: "DNN magnitude-mask model on STFT, batch 2, PyTorch CPU
(plumbing, no GPU)") -- NOT a fallback of the HIP path.

It is reached only through ``Solver(device="cpu")`` with a model of ``TORCH_MODELS`` (stock-PyTorch modules by design,
SURVEY section 2 row 11).  Nothing here is ever selected because a GPU, the HIP library or a kernel is missing: the HIP
models (dccrn, dcunet, ...) raise SehipError on a CPU tensor, and ``sehip.evaluate.stft_custom`` / the HIP losses do so too.
"""
import torch
import torch.nn.functional as F

TORCH_MODELS = ("dnn",)


def stft_custom(tensor, config):
    """src/evaluate.py:101-128 as the reference runs it on CPU: torch.stft(hann, reflect-centre, one-sided) / win_length."""
    lead, n = tuple(tensor.shape[:-1]), tensor.shape[-1]
    spec = torch.stft(tensor.contiguous().view(-1, n), n_fft=config.n_fft, hop_length=config.hop_length,
                      win_length=config.win_length, window=torch.hann_window(config.win_length, dtype=tensor.dtype),
                      center=config.center, pad_mode="reflect", normalized=False, onesided=None, return_complex=True)
    spec = torch.view_as_real(spec) / config.win_length
    return spec.reshape(*lead, *spec.shape[1:])


LOSSES = {"l1": F.l1_loss, "mse": F.mse_loss}
