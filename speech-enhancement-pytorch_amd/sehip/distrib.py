"""Factories with the reference's names and argument meaning (src/distrib.py:226-275):
get_model / get_optimizer / get_loss_function, plus the data-parallel helpers the reference lacks
(it wraps the model in single-process nn.DataParallel, src/solver.py:144-145): one process per GPU,
gradient all-reduce of the flat buffer over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" on CPU tests).
"""
import os

import torch
import torch.distributed as dist

from ._lib import SehipError
from .loss import loss_sisdr, l1_loss, mse_loss, loss_phase_sensitive_spectral_approximation
from .model.dccrn import DCCRN
from .model.conv_tasnet import ConvTasNet
from .model.dcunet import DCUnet
from .model.dnn import DeepNeuralNetwork
from .model.demucs import Demucs
from .optim import FlatOptimizer
from .utils import obj2dict

MODEL_REGISTRY = {"dccrn": DCCRN, "dcunet": DCUnet, "dnn": DeepNeuralNetwork, "conv-tasnet": ConvTasNet, "demucs": Demucs}
_REFERENCE_NAMES = ("dnn", "mel-rnn", "unet", "dccrn", "dcunet", "demucs", "wav-unet", "conv-tasnet", "crn", "rnn-stft-mask")


def get_model(config):
    if config.name not in MODEL_REGISTRY:
        if config.name in _REFERENCE_NAMES:
            raise SehipError(f"model '{config.name}' is in the reference registry but has no HIP path yet (built: "
                             f"{sorted(MODEL_REGISTRY)})")
        raise KeyError(config.name)
    return MODEL_REGISTRY[config.name](**obj2dict(config))


def get_optimizer(config, model):
    if not hasattr(model, "flat_params"):   # stock-PyTorch model (C0 plumbing config): what src/distrib.py:244-261 builds
        if config.optim == "sgd":
            return torch.optim.SGD(params=model.parameters(), lr=config.lr, momentum=config.momentum)
        if config.optim == "adam":
            return torch.optim.Adam(params=model.parameters(), lr=config.lr, betas=(config.beta1, config.beta2))
        raise ValueError(f"Optimizer {config.optim} cannot use...")
    if config.optim == "sgd":
        return FlatOptimizer(model, lr=config.lr, kind="sgd", momentum=config.momentum)
    if config.optim == "adam":
        return FlatOptimizer(model, lr=config.lr, kind="adam", betas=(config.beta1, config.beta2))
    raise ValueError(f"Optimizer {config.optim} cannot use...")


def get_loss_function(config, device="gpu"):
    if str(device) == "cpu":   # the explicit CPU plumbing configuration (sehip/plumbing.py), never a fallback
        from . import plumbing
        if config.loss in plumbing.LOSSES:
            return plumbing.LOSSES[config.loss]
        raise SehipError(f"loss '{config.loss}' is not part of the CPU plumbing configuration")
    if config.loss == "l1":  # mae
        return l1_loss
    if config.loss == "mse":
        return mse_loss
    if config.loss == "si-sdr":
        return loss_sisdr
    if config.loss == "psa":      # src/distrib.py:270-271; the Solver passes the mixture as the third argument (Solver._loss)
        return loss_phase_sensitive_spectral_approximation
    raise ValueError(f"Loss function {config.loss} cannot use...")


# ---- data parallel over utterances -----------------------------------------------------------------
def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: several ranks on ONE GPU (RCCL refuses duplicate devices, gloo moves CUDA tensors through the host)
    backend = backend or os.environ.get("SEHIP_DIST_BACKEND")
    if "SEHIP_LOCAL_DEVICE" in os.environ:
        local = int(os.environ["SEHIP_LOCAL_DEVICE"])
    direct = backend == "sehip-rccl"
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend="gloo" if direct else backend, rank=rank, world_size=world)
    if direct:
        global _direct
        if _direct is None:
            _direct = DirectComm(rank, world, torch.device("cuda", local))
    return rank, world, local


class DirectComm:
    """The gradient exchange through libsehip's own RCCL entry points (sehip_comm_init / sehip_allreduce_f32, include/sehip.h)
    instead of torch.distributed's collectives: selected with SEHIP_DIST_BACKEND=sehip-rccl.  torch.distributed (gloo) then only
    carries the control plane: the 128-byte RCCL id from rank 0, barriers, the epoch score."""

    def __init__(self, rank, world, device):
        import ctypes as C
        from ._lib import call
        self.rank, self.world, self.device = rank, world, device
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            call("sehip_comm_unique_id", ident.data_ptr())
        if world > 1:
            dist.broadcast(ident, 0)
        h = C.c_void_p()
        with torch.cuda.device(device):
            call("sehip_comm_init", ident.data_ptr(), world, rank, C.byref(h))
        self.handle = h

    def all_reduce_(self, flat, lo=0, hi=None, stream=None):
        """in-place SUM of flat[lo:hi] (fp32, contiguous) on `stream` (torch.cuda.Stream; None = current); returns an object whose
        wait() makes the CURRENT stream wait for the exchange (like a torch.distributed work handle)."""
        from ._lib import call
        hi = flat.numel() if hi is None else hi
        st = stream if stream is not None else torch.cuda.current_stream()
        call("sehip_allreduce_f32", self.handle, flat.data_ptr() + 4 * lo, hi - lo, st.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(st)

        class _Work:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(ev)
        return _Work()

    def all_reduce_max_i32_(self, words, stream=None):
        """in-place MAX of an int32 device tensor over the ranks on `stream` (None = current): the global step guard."""
        from ._lib import call
        st = stream if stream is not None else torch.cuda.current_stream()
        call("sehip_allreduce_i32_max", self.handle, words.data_ptr(), words.numel(), st.cuda_stream)

    def info(self):
        """(nranks, rank, device) as the live RCCL communicator reports them (sehip_comm_info)."""
        import ctypes as C
        from ._lib import call
        n, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        call("sehip_comm_info", self.handle, C.byref(n), C.byref(r), C.byref(d))
        return n.value, r.value, d.value

    def close(self):
        from ._lib import call
        if self.handle:
            call("sehip_comm_destroy", self.handle)
            self.handle = None


_direct = None


def direct_comm():
    """The DirectComm of this process (SEHIP_DIST_BACKEND=sehip-rccl), or None."""
    return _direct


def broadcast_parameters(flat_params, flat_buffers=None, src=0):
    """Every replica starts from rank 0's weights (DataParallel replicates module 0 each step)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src)
        if flat_buffers is not None:
            dist.broadcast(flat_buffers, src)


def allreduce_module_gradients(model):
    """Models without a flat gradient buffer (stock-PyTorch plumbing models): one coalesced all-reduce of their .grad tensors."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        flat = torch._utils._flatten_dense_tensors(grads)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / dist.get_world_size())
        for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(f)


def allreduce_gradients(flat_grads, scale=True):
    """ONE all-reduce(sum) of the flat fp32 gradient buffer [then x 1/world]: the mean over the global batch
    (equal shards), i.e. what DataParallel's gather + loss mean + backward reduce produces on GPU 0.
    Clipping happens after this (src/solver.py:487-490 clips the reduced gradients).  scale=False leaves the SUM: the fused
    optimizer applies 1/world itself (FlatOptimizer.grad_scale, no extra pass over the buffer)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if _direct is not None and flat_grads.is_cuda:
            _direct.all_reduce_(flat_grads).wait()
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
        if scale:
            flat_grads.mul_(1.0 / dist.get_world_size())
    return flat_grads


def allreduce_step_guard(guard):
    """Makes a model's device-side step guard (model.step_guard(): int32 word(s), non-zero = this step's gradients are invalid, e.g.
    Demucs' hand-off time-out word) GLOBAL before the optimizer launch: MAX over the ranks, in place, on the current stream.  Without
    it a time-out on one rank sends that rank's garbage into the SUM all-reduce and only that rank skips its update -- the replicas
    diverge in parameters, Adam moments and step count.  With it every rank's sehip_opt_*_g sees the same word, every
    rank skips the same step and model.check_health() takes the same fall-back on all of them."""
    if guard is None or not (dist.is_initialized() and dist.get_world_size() > 1):
        return guard
    if _direct is not None and guard.is_cuda:
        _direct.all_reduce_max_i32_(guard)
    else:
        dist.all_reduce(guard, op=dist.ReduceOp.MAX)
    return guard


def global_flag(flag, device=None):
    """OR of a host-side boolean over the ranks (MAX all-reduce of one int; a no-op without a process group).  For decisions every
    rank must take together although the evidence is rank-local: model.check_health()'s fall-back after a hand-off time-out seen in
    an eval forward on one rank only -- a rank that alone changed its launch path would pair a different collective
    sequence with the others'.  Collective: every rank must call it at the same point."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return bool(flag)
    on_gpu = device is not None and torch.device(device).type == "cuda" and dist.get_backend() != "gloo"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t[0]))


def allreduce_step_guard_async(guard, stream=None):
    """The same MAX all-reduce enqueued behind `stream` (a torch.cuda.Stream; None = current), returning a work handle whose .wait()
    makes the CURRENT stream wait for it: for models whose guard word is final before the backward pass ends (DCCRN: once the fused
    LSTM backward launch is enqueued) the exchange then hides under the rest of the pass instead of sitting between the last
    gradient and the optimizer."""
    class _Done:
        def wait(self_inner):
            pass
    if guard is None or not (dist.is_initialized() and dist.get_world_size() > 1):
        return _Done()
    if _direct is not None and guard.is_cuda:
        st = stream if stream is not None else torch.cuda.current_stream()
        _direct.all_reduce_max_i32_(guard, st)
        ev = torch.cuda.Event()
        ev.record(st)

        class _Work:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(ev)
        return _Work()
    if stream is not None and guard.is_cuda:
        with torch.cuda.stream(stream):
            return dist.all_reduce(guard, op=dist.ReduceOp.MAX, async_op=True)
    return dist.all_reduce(guard, op=dist.ReduceOp.MAX, async_op=True)


def allreduce_range_async(flat_grads, lo, hi, stream=None):
    """all-reduce(sum) of flat_grads[lo:hi], enqueued behind `stream` (a torch.cuda.Stream, or None for the current one);
    returns the work handle (.wait() makes the CURRENT stream wait for it).  Used by the Solver to start the exchange of
    the decoder / LSTM gradients while the encoder's backward pass still runs (the reference has no overlap: DataParallel
    reduces during its single backward, src/solver.py:144-145, 485)."""
    if _direct is not None and flat_grads.is_cuda:
        return _direct.all_reduce_(flat_grads, lo, hi, stream)
    view = flat_grads[lo:hi]
    if stream is not None and flat_grads.is_cuda:
        with torch.cuda.stream(stream):
            return dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)
    return dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_mean_scalar(value):
    """Mean of a host scalar over the ranks (the epoch score that drives _is_best / early stopping in Solver.train)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()
