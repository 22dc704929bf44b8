"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's phase-sensitive spectral approximation loss,
src/loss.py:32-56 (`optim.loss: psa`; src/distrib.py:270-271; called as loss(enhanced, sources, mixture) by src/solver.py:480).
Pinned by tests/golden/psa_loss.npz (oracle/gen_golden_psa.py imports the reference)."""
import torch


def psa_loss(enhance, target, mixture, eps=1e-9):
    """enhance / target / mixture [..., 2] (real, imaginary) spectra of one shape -> scalar.
    src/loss.py:48-55: the 'angles' are tanh of the imaginary / real ratio (sic), the amplitudes sqrt(re^2 + im^2)."""
    angle_mixture = torch.tanh(mixture[..., 1] / (mixture[..., 0] + eps))          # :48
    angle_target = torch.tanh(target[..., 1] / (target[..., 0] + eps))             # :49
    amplitude_enhance = torch.sqrt(enhance[..., 1] ** 2 + enhance[..., 0] ** 2)    # :51
    amplitude_target = torch.sqrt(target[..., 1] ** 2 + target[..., 0] ** 2)       # :52
    d = amplitude_enhance - amplitude_target * torch.cos(angle_target - angle_mixture)   # :54
    return torch.mean(d ** 2)                                                      # :55
