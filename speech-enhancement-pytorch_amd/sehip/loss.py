"""Loss functions of the train step (reference: src/loss.py:14-29 si_snr / loss_sisdr; l1 / mse are
torch.nn.functional in the reference, src/distrib.py:263-275)."""
import torch

from . import ops
from ._lib import SehipError


class _SiSdrLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, ref):
        n = est.shape[-1]
        e2 = est.reshape(-1, n).contiguous().float()
        r2 = ref.reshape(-1, n).contiguous().float()
        loss, rowstat = ops.sisnr_fwd(e2, r2)
        ctx.save_for_backward(e2, r2, rowstat)
        ctx.shape = est.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        e2, r2, rowstat = ctx.saved_tensors
        d = ops.sisnr_bwd(e2, r2, rowstat, g.reshape(1).contiguous().float())
        return d.view(ctx.shape), None


def loss_sisdr(inputs, targets):
    """-mean(si_snr(inputs, targets)) over all leading dims (src/loss.py:25-29)."""
    if inputs.shape != targets.shape:
        raise SehipError(f"loss_sisdr: shape mismatch {tuple(inputs.shape)} vs {tuple(targets.shape)}")
    return _SiSdrLoss.apply(inputs, targets)


def si_snr(s1, s2):
    return -loss_sisdr(s1, s2)


class _PointwiseLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mode):
        from ._lib import call, ptr, stream, require_gpu
        require_gpu(x, "l1/mse loss")
        xf, yf = x.contiguous().float(), y.contiguous().float()
        acc = torch.zeros(1, dtype=torch.float64, device=x.device)
        loss = torch.empty(1, device=x.device)
        call("sehip_pointwise_loss_fwd", ptr(xf), ptr(yf), xf.numel(), mode, ptr(acc), ptr(loss), stream())
        ctx.save_for_backward(xf, yf)
        ctx.mode, ctx.shape = mode, x.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        from ._lib import call, ptr, stream
        xf, yf = ctx.saved_tensors
        dx = torch.empty_like(xf)
        call("sehip_pointwise_loss_bwd", ptr(xf), ptr(yf), xf.numel(), ctx.mode, ptr(g.reshape(1).contiguous().float()), ptr(dx),
             stream())
        return dx.view(ctx.shape), None, None


def l1_loss(inputs, targets):
    """torch.nn.functional.l1_loss(reduction='mean') as used by src/distrib.py:264-265."""
    if inputs.shape != targets.shape:
        raise SehipError(f"l1_loss: shape mismatch {tuple(inputs.shape)} vs {tuple(targets.shape)}")
    return _PointwiseLoss.apply(inputs, targets, 0)


def mse_loss(inputs, targets):
    """torch.nn.functional.mse_loss(reduction='mean') as used by src/distrib.py:266-267."""
    if inputs.shape != targets.shape:
        raise SehipError(f"mse_loss: shape mismatch {tuple(inputs.shape)} vs {tuple(targets.shape)}")
    return _PointwiseLoss.apply(inputs, targets, 1)
