"""Solver: the train loop of the reference (src/solver.py:109-532) on the HIP path.

Kept from the reference: constructor signature, train() / _run_one_epoch() control flow (step caps, asserts on
channels and speakers, MONARCH reshape, model.train()/eval(), the overwritten PIT loss, zero_grad / backward /
clip / step, the sum-based grad_norm metric, tag names of the logged scalars, score dict incl. the
'validation metric loss returns the TRAIN loss' quirk), checkpoint files and their keys (latest_model.tar,
model_<epoch>_<metric>_<best>.pth, best_model.tar, state.json), resume / preload.
Changed by design: one process per GPU with an RCCL all-reduce of the flat gradient buffer instead of
nn.DataParallel (src/solver.py:144-145); clip + optimizer are one fused launch; per-step `.item()` syncs happen
every `config.solver.log_interval` steps (default 1 = the reference's cadence); no CUDA_LAUNCH_BLOCKING.
Out of scope: inference / metrics / plots (src/solver.py:534-746).
"""
import datetime
import json
import os
import time
from pathlib import Path
from shutil import copyfile

import numpy as np
import torch
import torch.distributed as dist

from . import distrib, plumbing
from ._lib import SehipError
from .model.types import MONARCH_SPEECH_SEPARTAION_MODELS, MULTI_SPEECH_SEPERATION_MODELS, STFT_MODELS
from .evaluate import stft_custom
from .optim import FlatOptimizer
from .utils import obj2dict

try:  # tensorboard is optional in this image
    from torch.utils.tensorboard import SummaryWriter as _TBWriter
except Exception:  # pragma: no cover
    _TBWriter = None


class ScalarLog:
    """SummaryWriter stand-in that keeps the scalars in memory (used when tensorboard is absent)."""

    def __init__(self, *a, **k):
        self.scalars = []

    def add_text(self, *a, **k):
        pass

    def add_scalar(self, tag, value, step=None, *a, **k):
        self.scalars.append((tag, float(value), -1 if step is None else int(step)))

    def add_scalars(self, *a, **k):
        pass

    def add_figure(self, *a, **k):
        pass


def _cfg(obj, name, default):
    return getattr(obj, name, default)


class DevicePrefetcher:
    """Iterates a dataloader ONE batch ahead: the host -> HBM copies of batch k + 1 are enqueued on a stream of their own while step k
    computes, and the consumer's stream waits for them when it takes the batch (with a pinned-memory DataLoader the copies overlap
    the step; 2 x 4.1 MB per DCCRN step cost 0.37 ms on the step's own stream, `bench.py --h2d`).  Yields the batch tuple with its
    tensors on the device; everything else (lengths, names) passes through.  The reference moves the batch with `.to(device)` at the
    top of the step (src/solver.py:448-451): same tensors, earlier."""

    def __init__(self, loader, device, limit=None):
        self.it = iter(loader)
        self.device = device
        self.limit = limit
        self.taken = 0
        self.stream = torch.cuda.Stream(device=device)
        self.ready = None
        self._preload()

    def _preload(self):
        self.ready = None
        if self.limit is not None and self.taken >= self.limit:
            return                                  # (do not pull batches the epoch will not use)
        try:
            batch = next(self.it)
        except StopIteration:
            return
        self.taken += 1
        with torch.cuda.stream(self.stream):
            self.ready = tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)

    def __iter__(self):
        return self

    def __next__(self):
        if self.ready is None:
            raise StopIteration
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        batch = self.ready
        for t in batch:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)                # the caching allocator must not hand the memory back to the copy stream early
        self._preload()
        return batch


class Solver(object):
    def __init__(self, config, model, optimizer=None, loss_function=None, train_dataloader=None,
                 validation_dataloader=None, test_dataloader=None, device="gpu", writer=None):
        self.train_dataloader = train_dataloader
        self.validation_dataloader = validation_dataloader
        self.test_dataloader = test_dataloader
        self.audiogram_dataloader = None  # hearing-aid post-processing is outside the train step

        self.rank, self.world_size, self.local_rank = distrib.init_distributed() if device == "gpu" else (0, 1, 0)
        self.n_gpu = torch.cuda.device_count()
        if device == "gpu":
            if self.n_gpu == 0:
                raise SehipError("Solver(device='gpu'): no GPU visible and the HIP path has no CPU fallback")
            self.device = torch.device(f"cuda:{self.local_rank}")
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device(device) if not isinstance(device, torch.device) else device
            if self.device.type == "cpu" and config.model.name not in plumbing.TORCH_MODELS:
                raise SehipError(f"Solver(device='cpu'): model '{config.model.name}' runs on the HIP path only; the CPU "
                                 f"plumbing configuration exists for {plumbing.TORCH_MODELS} (BASELINE config C0)")
        self.is_main = self.rank == 0
        # src/solver.py:133 hands config.solver.cudnn_deterministic to prepare_device (src/utils.py:108-111; the shipped YAML sets it
        # True).  Here it selects the fixed-order reductions of libsehip and of the model's plan: two runs of the same steps are
        # bit-identical.  Built for every HIP model (DCCRN, DCUnet, ConvTasNet, Demucs); a model without set_deterministic() keeps
        # its own atomics and says so once.
        self.deterministic = bool(_cfg(config.solver, "cudnn_deterministic", False)) and device == "gpu"
        if self.deterministic and _cfg(config.solver, "use_graph", False):
            # (the deterministic schedule sizes its per-split partial arrays on demand, which cannot happen inside a stream capture: say so
            #  before a capture is left half-open -- ADVICE r4)
            raise SehipError("solver.cudnn_deterministic and solver.use_graph cannot be combined: the deterministic schedule allocates its "
                             "partial-sum arrays on demand (not capturable); switch one of them off")
        if device == "gpu":
            # process-wide library switch: set BOTH ways, so that a Solver without the flag does not inherit the fixed-order (~1.6 x slower)
            # reductions of an earlier Solver of the same process
            from .utils import set_deterministic
            set_deterministic(self.deterministic)
            if hasattr(model, "set_deterministic"):
                model.set_deterministic(self.deterministic)
        if self.deterministic and not hasattr(model, "set_deterministic"):
            import warnings
            warnings.warn(f"solver.cudnn_deterministic: model '{config.model.name}' has no deterministic plan yet (its normalisation "
                          f"sums use fp32 atomics); the library-level reductions are deterministic, run-to-run bit equality is not guaranteed")

        self.optimizer = optimizer
        self.loss_function = loss_function
        self.model = model.to(self.device)
        self.flat_model = hasattr(self.model, "flat_params")
        if self.world_size > 1:
            if self.flat_model:
                distrib.broadcast_parameters(self.model.flat_params, self.model._bflat)
            else:
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    dist.broadcast(t.data, 0)

        self.epochs = config.solver.epochs
        self.save_checkpoint_interval = config.solver.save_checkpoint_interval
        self.validation_interval = config.solver.validation.interval
        self.test_interval = config.solver.test.interval
        self.log_interval = int(_cfg(config.solver, "log_interval", 1))

        self.find_max = config.solver.validation.metric in ("stoi", "pesq", "sisdr", "haspi", "hasqi")
        self.score = {"best_score": -np.inf if self.find_max else np.inf, "loss": 0, "loss_valid": 0, "grad_norm": 0,
                      "stoi": [], "pesq": [], "sisdr": [], "haspi": [], "hasqi": []}

        self.root_dir = Path(config.solver.root) / "result" / config.model.name / datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
        self.checkpoints_dir = self.root_dir / "checkpoints"
        self.logs_dir = self.root_dir / "logs"
        if self.is_main:
            for d in (self.checkpoints_dir, self.logs_dir):
                d.mkdir(parents=True, exist_ok=True)
        if writer is not None:
            self.writer = writer
        elif _TBWriter is not None and self.is_main:
            self.writer = _TBWriter(log_dir=self.logs_dir.as_posix(), max_queue=5, flush_secs=30)
        else:
            self.writer = ScalarLog()
        self.writer.add_text(tag="Configuration",
                             text_string=f"<pre>  \n{json.dumps(obj2dict(config), indent=4, sort_keys=False)}  \n</pre>",
                             global_step=1)
        self.config = config
        if config.solver.preloaded_model:
            self._preload_model()
        elif config.solver.resume:
            self._resume_checkpoint()
        if self.is_main and getattr(config, "root", None) and os.path.exists(config.root):
            copyfile(config.root, (self.root_dir / "config.yaml").as_posix())
        self._print_networks([self.model])

    # ---- checkpoints (src/solver.py:233-341) ---------------------------------------------------------
    def _resume_checkpoint(self):
        latest = Path(self.config.solver.resume) / "checkpoints/latest_model.tar"
        assert latest.exists(), f"{latest} does not exist, can not load latest checkpoint."
        checkpoint = torch.load(latest.as_posix(), map_location=self.device, weights_only=False)
        self.best_score = checkpoint["best_score"]
        if self.config.optim.load:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        self.model.load_state_dict(checkpoint["model"])
        print(f"\tModel checkpoint loaded, {latest.as_posix()}")

    def _preload_model(self):
        path = Path(self.config.solver.preloaded_model)
        assert path.exists(), f"Preloaded *.pth file is not exist. Please check the file path: {path.as_posix()}"
        self.model.load_state_dict(torch.load(path, map_location=self.device, weights_only=False), strict=False)
        print(f"\tModel preloaded successfully from {path.as_posix()}.")

    @staticmethod
    def _print_networks(nets):
        total = 0
        for i, net in enumerate(nets, start=1):
            n = sum(p.numel() for p in net.parameters())
            print(f"\tNetwork {i}: {n / 1e6} million.")
            total += n
        print(f"The amount of parameters in the project is {total / 1e6} million.")

    def _model_health(self):
        """Models with device-side failure words (Demucs: the hand-off time-out of its persistent LSTM kernels) are asked at every
        point where the Solver waits for the device anyway; the affected optimizer steps were already no-ops on the device."""
        fn = getattr(self.model, "check_health", None)
        if fn is not None and fn():
            self.lost_steps = getattr(self, "lost_steps", 0) + 1
            self.writer.add_scalar("Train/lost_steps", self.lost_steps, -1)

    def _save_checkpoint(self, epoch, is_best=False):
        self._model_health()
        if not self.is_main:
            return
        state_dict = {"epoch": epoch, "best_score": self.score["best_score"], "optimizer": self.optimizer.state_dict()}
        state_dict["model"] = {k: v.cpu() for k, v in self.model.state_dict().items()}
        torch.save(state_dict, (self.checkpoints_dir / "latest_model.tar").as_posix())
        torch.save(state_dict["model"], (self.checkpoints_dir /
                   f"model_{str(epoch).zfill(4)}_{self.config.solver.validation.metric}_{self.score['best_score']:2.8f}.pth").as_posix())
        with open(self.checkpoints_dir / "state.json", "w") as tmp:
            json.dump(self.score, tmp, indent=4)
        if is_best:
            torch.save(state_dict, (self.checkpoints_dir / "best_model.tar").as_posix())

    def _is_best(self, score, find_max=True):
        """Same rule as src/solver.py:343-353.  With several ranks `score` is the mean over ranks (train() reduces it
        first), so every rank takes the same branch, keeps the same best_score / early-stopping counter and leaves the
        epoch loop together -- a rank-local decision would strand the others in the next gradient all-reduce."""
        if find_max and score >= self.score["best_score"]:
            self.score["best_score"] = score
            return True
        if not find_max and score <= self.score["best_score"]:
            self.score["best_score"] = score
            return True
        return False

    # ---- train loop (src/solver.py:355-532) ------------------------------------------------------------
    def train(self):
        patience = self.config.solver.patience
        early_stopping = 0
        for epoch in range(self.epochs):
            start_time = time.time()
            score = self._run_one_epoch(epoch, self.epochs, train=True)
            if epoch % self.save_checkpoint_interval == 0:
                self._save_checkpoint(epoch)
            if epoch % self.validation_interval == 0:
                score = self._run_one_epoch(epoch, self.epochs, train=False)
                score = distrib.allreduce_mean_scalar(score)   # one decision for all ranks (see _is_best)
                if self._is_best(score, find_max=self.find_max):
                    self._save_checkpoint(epoch, is_best=True)
                    early_stopping = 0
                else:
                    early_stopping += 1
            if early_stopping > patience:
                break
            print(f"[{int(time.time() - start_time)} seconds] End this epoch.")

    def _prepare_batch(self, mixture, sources):
        cfg = self.config
        mixture = mixture.to(self.device, non_blocking=True)
        sources = sources.to(self.device, non_blocking=True)
        batch, nchannel, nsample = mixture.shape
        num_spk = sources.shape[1]
        assert cfg.model.audio_channels == nchannel, f"Channel between {cfg.dset.name} and {cfg.model.name} did not match..."
        assert cfg.model.num_spk == num_spk, f"number of speakers between {cfg.dset.name} and {cfg.model.name} did not match..."
        if cfg.model.name in MULTI_SPEECH_SEPERATION_MODELS:
            assert num_spk == len(cfg.model.sources)
        if cfg.model.name in MONARCH_SPEECH_SEPARTAION_MODELS:
            sources = torch.squeeze(sources, dim=1)
            mixture = torch.reshape(mixture, shape=(batch * nchannel, 1, nsample))
            sources = torch.reshape(sources, shape=(batch * num_spk * nchannel, 1, nsample))
        if cfg.model.name in STFT_MODELS:
            # src/solver.py:454-458: mixture and sources go to the STFT domain [B, C, F, T, 2] and the loss is taken there.
            # HIP kernels on the GPU; torch.stft only in the explicit CPU plumbing configuration (sehip/plumbing.py)
            stft = plumbing.stft_custom if mixture.device.type == "cpu" else stft_custom
            mixture = stft(tensor=mixture, config=cfg.model)
            sources = stft(tensor=sources, config=cfg.model)
        return mixture, sources

    def train_step(self, mixture, sources):
        """One optimisation step on device tensors; returns (loss, grad_metric[2]) as DEVICE tensors (no sync)."""
        if mixture.is_cuda:
            from ._lib import stream_scope
            with stream_scope():
                return self._train_step(mixture, sources)
        return self._train_step(mixture, sources)

    def _unit_grad(self, loss):
        """d loss / d loss = 1 as a CACHED device tensor: `loss.backward()` alone makes autograd fill a fresh ones_like(loss) every step --
        a 7-us kernel plus a boundary on the step's dependent chain for one float."""
        if os.environ.get("SEHIP_NO_UNIT_GRAD"):
            return None
        g = getattr(self, "_unit", None)
        if g is None or g.device != loss.device or g.dtype != loss.dtype or g.shape != loss.shape:
            g = self._unit = torch.ones_like(loss)
        return g

    def _loss(self, enhanced, sources, mixture):
        """src/solver.py:476-480: `optim.loss: psa` takes the mixture's spectrum as a third argument (broadcast over the speakers when the
        model separates); every other loss is loss(enhanced, sources)."""
        if _cfg(self.config.optim, "loss", None) != "psa":
            return self.loss_function(enhanced, sources)
        if sources.dim() == mixture.dim() + 1:
            mixture = mixture.unsqueeze(1).expand_as(sources)
        return self.loss_function(enhanced, sources, mixture)

    def _train_step(self, mixture, sources):
        if not self.model.training:
            self.model.train()
        enhanced = self.model(mixture)
        # The reference computes a PIT loss here and then overwrites it with the plain loss (src/solver.py:469-480; its shipped
        # config has optim.pit: True), so optim.pit alone changes nothing here either.  The opt-in optim.pit_apply keeps the
        # permutation-invariant value (src/loss.py:58-100) for models that return [batch, speakers, ...].
        if _cfg(self.config.optim, "pit_apply", False) and sources.dim() >= 3 and sources.shape[1] >= 2 and \
                self.config.model.name in MULTI_SPEECH_SEPERATION_MODELS:
            from .loss import pit_loss
            loss = pit_loss(enhanced, sources, self.loss_function)
        else:
            loss = self._loss(enhanced, sources, mixture)
        self.optimizer.zero_grad()
        fused = isinstance(self.optimizer, FlatOptimizer)
        works, early_guard = [], False
        if self.world_size > 1 and self.flat_model and fused:
            # one SUM all-reduce of the flat buffer, in two ranges where the model hands them over early (the decoder / LSTM
            # range starts its exchange under the encoder's backward pass); 1/world is folded into the optimizer launch
            self.optimizer.grad_scale = 1.0 / self.world_size
            if hasattr(self.model, "_tail_sink"):
                self.model._tail_sink = None      # the optimizer's sums must be taken AFTER the all-reduce: no fused single-replica tail
            works = []
            early_guard = bool(getattr(self.model, "step_guard_early", False)) and hasattr(self.model, "step_guard")
            if hasattr(self.model, "grad_range_hook"):
                def hook(lo, hi, st):
                    # (a model whose guard word is final when it hands over its first gradient range -- DCCRN: after the fused LSTM
                    #  backward launch -- gets the guard's MAX all-reduce started there, on the same communication stream)
                    if early_guard and not works:
                        works.append(distrib.allreduce_step_guard_async(self.model.step_guard(), st))
                    works.append(distrib.allreduce_range_async(self.model.flat_grads, lo, hi, st))
                self.model.grad_range_hook = hook
            loss.backward(self._unit_grad(loss))
            if hasattr(self.model, "grad_range_hook"):
                self.model.grad_range_hook = None
            if works:
                for w in works:
                    w.wait()
            else:
                distrib.allreduce_gradients(self.model.flat_grads, scale=False)
        else:
            loss.backward(self._unit_grad(loss))
            if self.world_size > 1:
                if self.flat_model:
                    distrib.allreduce_gradients(self.model.flat_grads)
                else:
                    distrib.allreduce_module_gradients(self.model)
        if self.world_size > 1 and hasattr(self.model, "step_guard") and not (fused and self.flat_model and works and early_guard):
            # a device-side failure word (Demucs: hand-off time-out) must stop the update on EVERY rank: the all-reduced gradients
            # already contain the failing rank's contribution
            distrib.allreduce_step_guard(self.model.step_guard())
        if self.config.optim.clip_grad:
            if fused:
                self.optimizer.clip_grad_norm_(self.config.optim.clip_grad)
            else:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.config.optim.clip_grad)
        self.optimizer.step()
        if fused:
            metric = self.optimizer.grad_metric()
        else:   # the reference's sum-based metric (src/solver.py:494-498), kept on the device
            sums = torch.stack([p.grad.sum() for p in self.model.parameters() if p.grad is not None])
            metric = torch.stack([sums.square().sum().sqrt(), sums.new_zeros(())])
        return loss.detach(), metric

    # ---- hipGraph replay of the step (config.solver.use_graph) ------------------------------------------
    def train_step_graphed(self, mixture, sources):
        """Same step as train_step(), captured once per batch shape into two hipGraphs (forward+loss+backward, and
        clip+optimizer+metric; the RCCL all-reduce runs between them) and replayed: ~330 launches become 2.
        Inputs are copied into static buffers; returns the static (loss, metric) device tensors."""
        key = (tuple(mixture.shape), tuple(sources.shape))
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        g = self._graphs.get(key)
        if g is not None and g["epoch"] != self.model.storage_epoch:
            g = None   # the model's flat buffers were re-created (.to() / _apply): the captured pointers are stale
        if g is None:
            g = self._graphs[key] = self._capture_step(mixture, sources)
        g["mix"].copy_(mixture)
        g["src"].copy_(sources)
        g["fb"].replay()
        if self.world_size > 1:
            distrib.allreduce_gradients(self.model.flat_grads, scale=False)   # 1/world is baked into the captured optimizer launch
            if hasattr(self.model, "step_guard"):
                distrib.allreduce_step_guard(self.model.step_guard())
        g["upd"].replay()
        return g["loss"], g["metric"]

    def _capture_step(self, mixture, sources):
        from ._lib import call
        if not isinstance(self.optimizer, FlatOptimizer):
            raise SehipError("train_step_graphed needs the fused FlatOptimizer")
        call("sehip_init")
        self.model.train()
        self.optimizer.grad_scale = 1.0 / self.world_size
        self.model.workspace(mixture.shape[0], mixture.shape[-1]).pinned = True  # all buffers exist before capture and stay
        self.optimizer._ensure_state()
        self.model.flat_grads
        if self.model._anchor is None or self.model._anchor.device != mixture.device:
            self.model._anchor = torch.zeros(1, device=mixture.device, requires_grad=True)
        mix, src = mixture.clone(), sources.clone()
        torch.cuda.synchronize()
        fb, upd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(fb):
            enhanced = self.model(mix)
            loss = self._loss(enhanced, src, mix)
            self.optimizer.zero_grad()
            loss.backward()
            loss_out = loss.detach().clone()
        with torch.cuda.graph(upd, pool=fb.pool()):
            if self.config.optim.clip_grad:
                self.optimizer.clip_grad_norm_(self.config.optim.clip_grad)
            self.optimizer.step()
            metric_out = self.optimizer.grad_metric().clone()
        # capture executed nothing: undo the host-side step increment made while recording
        self.optimizer._step -= 1
        return dict(fb=fb, upd=upd, mix=mix, src=src, loss=loss_out, metric=metric_out, epoch=self.model.storage_epoch)

    def _run_one_epoch(self, epoch, total_epoch, train=False):
        cfg = self.config
        loss_total = 0.0
        grad_norm_total = 0.0
        dataloader = self.train_dataloader if train else self.validation_dataloader
        total_step = len(dataloader)
        if not cfg.solver.all_steps:
            if train and total_step > cfg.solver.total_steps:
                total_step = cfg.solver.total_steps
            elif not train and total_step > cfg.solver.validation.total_steps:
                total_step = cfg.solver.validation.total_steps
        pending = []  # device scalars waiting for the next logging point

        def flush():
            nonlocal loss_total, grad_norm_total
            for step_, loss_t, metric_t in pending:
                loss = float(loss_t)
                loss_total += loss
                if train:
                    grad_norm = float(metric_t[0]) if metric_t is not None else 0.0
                    grad_norm_total += grad_norm
                    self.writer.add_scalar("Train/Loss_step", loss, epoch * total_step + step_)
                    self.writer.add_scalar("Train/grad_norm_step", grad_norm, epoch * total_step + step_)
                else:
                    self.writer.add_scalar("Validation/Loss_step", loss, epoch * total_step + step_)
            pending.clear()
            self._model_health()

        batches = dataloader
        if self.device.type == "cuda" and _cfg(cfg.solver, "prefetch", True) and not os.environ.get("SEHIP_NO_PREFETCH"):
            batches = DevicePrefetcher(dataloader, self.device, limit=total_step)
        for step, batch in enumerate(batches):
            if step >= total_step:
                break
            mixture, sources = batch[0], batch[1]
            mixture, sources = self._prepare_batch(mixture, sources)
            if train:
                step_fn = self.train_step_graphed if _cfg(cfg.solver, "use_graph", False) else self.train_step
                loss_t, metric_t = step_fn(mixture, sources)
                # the graphed step returns its STATIC output tensors (overwritten by the next replay) and the fused
                # optimizer's metric scratch is reused too: deferred read-backs need their own copies
                pending.append((step, loss_t.clone(), metric_t.clone() if metric_t is not None else None))
            else:
                self.model.eval()
                with torch.no_grad():
                    enhanced = self.model(mixture)
                    loss_t = self._loss(enhanced, sources, mixture)
                pending.append((step, loss_t.detach().clone(), None))
            if (step + 1) % self.log_interval == 0:
                flush()
        flush()
        if train:
            self.score["loss"] = loss_total / total_step
            self.score["grad_norm"] = grad_norm_total / total_step
            self.writer.add_scalar("Train/Loss", self.score["loss"], epoch)
            self.writer.add_scalar("Train/Grad_norm", self.score["grad_norm"], epoch)
        else:
            self.score["loss_valid"] = loss_total / total_step
            self.writer.add_scalar("Validation/Loss", self.score["loss_valid"], epoch)
        # NB: with validation metric 'loss' this returns the TRAIN loss, exactly like src/solver.py:532
        return self.score["loss"] if train else self.score[cfg.solver.validation.metric]
