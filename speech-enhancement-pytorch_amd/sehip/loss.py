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


class _PitSiSdrLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, ref):
        b, s = est.shape[0], est.shape[1]
        n = est.shape[-1]
        e4 = est.reshape(b, s, -1, n).contiguous().float()
        r4 = ref.reshape(b, s, -1, n).contiguous().float()
        loss, rowstat, pairloss, perm = ops.sisnr_pit_fwd(e4, r4)
        ctx.save_for_backward(e4, r4, rowstat, perm)
        ctx.shape = est.shape
        ctx.mark_non_differentiable(perm, pairloss)
        return loss.reshape(()), perm, pairloss

    @staticmethod
    def backward(ctx, g, _gp, _gl):
        e4, r4, rowstat, perm = ctx.saved_tensors
        d = ops.sisnr_pit_bwd(e4, r4, rowstat, perm, g.reshape(1).contiguous().float())
        return d.view(ctx.shape), None


def pit_loss_sisdr(enhance, target, return_comb=False):
    """UtterenceBaasedPermutationInvariantTraining(enhance, target, loss_function=loss_sisdr) of src/loss.py:58-100 on the
    device: enhance / target [B, S, ...], speakers on axis 1.  As in the reference the permutation is chosen once per BATCH
    (on the batch-mean pair losses, no gradient through the choice) and the result is the mean of the matched pairs' losses.
    With return_comb the device tensor perm [S] (perm[j] = estimated speaker matched with target j) is returned as well --
    the reference's list of (ienhance, itarget) pairs without the host round trip."""
    if enhance.shape != target.shape:
        raise SehipError(f"enhance and target shape did not match...{tuple(enhance.shape)}, {tuple(target.shape)}")
    if enhance.dim() < 3:
        raise SehipError("pit_loss_sisdr: expected [batch, speakers, ..., samples]")
    loss, perm, _ = _PitSiSdrLoss.apply(enhance, target)
    return (loss, perm) if return_comb else loss


def pit_loss(enhance, target, loss_function, return_comb=False):
    """The same for any loss function of this module.  si-sdr runs fused on the device; the other losses evaluate the S x S
    pair matrix with S*S small launches and read it back once to pick the permutation (src/loss.py:67-86)."""
    if loss_function is loss_sisdr:
        return pit_loss_sisdr(enhance, target, return_comb)
    if enhance.shape != target.shape:
        raise SehipError(f"enhance and target shape did not match...{tuple(enhance.shape)}, {tuple(target.shape)}")
    from itertools import permutations
    s = enhance.shape[1]
    with torch.no_grad():
        m = torch.stack([torch.stack([loss_function(enhance[:, i].contiguous(), target[:, j].contiguous()) for j in range(s)])
                         for i in range(s)]).cpu()
    best, lmin = None, 1e9
    for pe in permutations(range(s)):
        l = sum(float(m[pe[j], j]) for j in range(s))
        if lmin > l:
            best, lmin = [(pe[j], j) for j in range(s)], l
    loss = sum(loss_function(enhance[:, i].contiguous(), target[:, j].contiguous()) for i, j in best) / s
    return (loss, best) if return_comb else loss


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


class _PsaLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enh, tgt, mix):
        from ._lib import call, ptr, stream, require_gpu
        require_gpu(enh, "psa loss")
        e, t, m = enh.contiguous().float(), tgt.contiguous().float(), mix.contiguous().float()
        acc = torch.zeros(1, dtype=torch.float64, device=enh.device)
        loss = torch.empty(1, device=enh.device)
        call("sehip_psa_loss_fwd", ptr(e), ptr(t), ptr(m), e.numel() // 2, ptr(acc), ptr(loss), stream())
        ctx.save_for_backward(e, t, m)
        ctx.shape = enh.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        from ._lib import call, ptr, stream
        e, t, m = ctx.saved_tensors
        de = torch.empty_like(e)
        call("sehip_psa_loss_bwd", ptr(e), ptr(t), ptr(m), e.numel() // 2, ptr(g.reshape(1).contiguous().float()), ptr(de), stream())
        return de.view(ctx.shape), None, None


def loss_phase_sensitive_spectral_approximation(enhance, target, mixture):
    """src/loss.py:32-56 (`optim.loss: psa`): mean (|E| - |T| cos(tanh(Ti / (Tr + eps)) - tanh(Mi / (Mr + eps))))^2 over [..., 2] spectra;
    the gradient flows into `enhance` only (target and mixture are data in the Solver, src/solver.py:480)."""
    if not (enhance.shape == target.shape == mixture.shape) or enhance.shape[-1] != 2:
        raise SehipError(f"psa loss: three [..., 2] tensors of one shape expected, got {tuple(enhance.shape)}, {tuple(target.shape)}, "
                         f"{tuple(mixture.shape)}")
    return _PsaLoss.apply(enhance, target, mixture)
