"""Fused optimizer on the model's flat parameter / gradient buffers
(reference: src/distrib.py:244-261 builds torch.optim.Adam / SGD; src/solver.py:487-498 clips, steps and
computes the sum-based grad_norm metric with one Python loop over 134 tensors).

`FlatOptimizer` keeps torch.optim.Optimizer's interface and state_dict format (per-parameter `step`, `exp_avg`,
`exp_avg_sq` / `momentum_buffer`, so checkpoints interchange with the reference), but the state tensors are views
into two flat fp32 buffers and `step()` is ONE HIP launch (plus one reduction when clipping).
"""
import torch

from ._lib import call, ptr, stream, SehipError


class FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, model, lr, kind="adam", betas=(0.9, 0.999), eps=1e-8, momentum=0.0, weight_decay=0.0):
        if not hasattr(model, "flat_params"):
            raise SehipError("FlatOptimizer needs a sehip model (flat_params / flat_grads)")
        self.model = model
        self.kind = kind
        defaults = dict(lr=lr, betas=betas, eps=eps, momentum=momentum, weight_decay=weight_decay)
        super().__init__(list(model.parameters()), defaults)
        self._m = self._v = None
        self._step = 0
        self._step_dev = None
        self._scratch = None
        self.max_norm = 0.0  # set by clip_grad_norm_() for the next step only
        self.grad_scale = 1.0  # 1/world when the flat gradients hold the all-reduced SUM of the replicas (Solver sets it)

    # ---- flat state -------------------------------------------------------------------------------
    def _ensure_state(self):
        flat = self.model.flat_params
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            old_m, old_v = self._m, self._v
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            if old_m is not None and old_m.numel() == flat.numel():
                self._m.copy_(old_m); self._v.copy_(old_v)
            self._step_dev = torch.full((1,), int(self._step), dtype=torch.int32, device=flat.device)
            # (sumsq / tsums: two sets, used alternately by the fused single-replica tail -- a step's optimizer launch clears the
            #  set the NEXT step adds to, so no clearing launch sits on the chain; the unfused path uses set 0)
            self._set = 0
            self._scratch = dict(sumsq=torch.zeros(1, dtype=torch.float64, device=flat.device),
                                 tsums=torch.zeros(len(self.model._params), device=flat.device),
                                 sumsq1=torch.zeros(1, dtype=torch.float64, device=flat.device),
                                 tsums1=torch.zeros(len(self.model._params), device=flat.device),
                                 metric=torch.zeros(2, device=flat.device),
                                 offsets=torch.from_numpy(self.model.static.layout.tensor_offsets).to(flat.device))
            self._publish_state()

    def _publish_state(self):
        L = self.model.static.layout
        for name, p in self.model._params:
            off, shape = L.param_off[name]
            sl = slice(off, off + p.numel())
            st = self.state[p]
            st["step"] = torch.tensor(float(self._step))
            if self.kind == "adam":
                st["exp_avg"] = self._m[sl].view(shape)
                st["exp_avg_sq"] = self._v[sl].view(shape)
            else:
                st["momentum_buffer"] = self._m[sl].view(shape)

    def clip_grad_norm_(self, max_norm):
        """Arms global-norm clipping (torch.nn.utils.clip_grad_norm_ semantics) for the next step();
        the norm is computed on device and never read back."""
        self.max_norm = float(max_norm) if max_norm else 0.0

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise SehipError("FlatOptimizer.step: closures are not supported")
        self._ensure_state()
        model = self.model
        params, grads = model.flat_params, model.flat_grads
        g = self.param_groups[0]
        self._step += 1  # host mirror; the kernel reads the device counter (valid under hipGraph replay)
        s = self._scratch
        if getattr(model, "_tail_done", False) and self.grad_scale == 1.0:
            # the backward pass un-packed the gradients WITH the clipping norm's and the metric's sums and advanced the device step
            # counter (sehip_unpack_grad_sums): one launch updates, derives the logged metrics and clears the other set of accumulators
            model._tail_done, model._tail_sink, model._tail_counted, model._tail_dirty = False, None, False, False
            cur, nxt = ("", "1") if self._set == 0 else ("1", "")
            gfn = getattr(model, "step_guard", None)
            guard = gfn() if gfn is not None else None
            if self.kind == "adam":
                (b1, b2), mode = g["betas"], 0
            else:
                b1, b2, mode = g["momentum"], 0.0, 1
            call("sehip_opt_step_m", ptr(params), ptr(grads), ptr(self._m), ptr(self._v), params.numel(), ptr(s["sumsq" + cur]),
                 self.max_norm, g["lr"], b1, b2, g["eps"], self._step, ptr(self._step_dev), g["weight_decay"], mode, 1.0, ptr(guard),
                 ptr(s["tsums" + cur]), s["tsums"].numel(), ptr(s["metric"]), ptr(s["sumsq" + nxt]), ptr(s["tsums" + nxt]), stream())
            if self._set == 1:                             # this launch has just cleared set 0 (its `next` set)
                self._set0_stale = False
            self._set ^= 1
            self._metric_fresh = True
            self.max_norm = 0.0
            self._unfused_run = 0
            return
        self._note_unfused_step()
        # one launch: step counter, and the accumulators of the clipping norm and of the grad_norm metric cleared
        # guard: device word that, when non-zero, turns this step into a no-op (model.step_guard(): Demucs' hand-off time-out word)
        gfn = getattr(model, "step_guard", None)
        guard = gfn() if gfn is not None else None
        counted = getattr(model, "_tail_counted", False)      # a fused un-pack whose sums were invalidated afterwards (a second, accumulating
        if counted:                                           # backward pass; a data-parallel scale set late) has already counted this step
            model._tail_counted = model._tail_done = False
        call("sehip_opt_begin_g", ptr(self._step_dev), 0 if counted else 1, ptr(s["sumsq"]), ptr(s["tsums"]), s["tsums"].numel(), ptr(guard),
             stream())
        self._tsums_clear = True
        # (this path -- and grad_metric() behind it -- leaves THIS step's sums in set 0: sehip_opt_begin_g clears at the START of a
        #  step only.  A fused tail armed on set 0 afterwards would add to them: _arm_fused_tail clears the set first -- ADVICE r5)
        self._set0_stale = True
        if self.max_norm > 0:
            call("sehip_grad_sumsq_acc", ptr(grads), grads.numel(), ptr(s["sumsq"]), stream())
        if self.kind == "adam":
            b1, b2 = g["betas"]
            mode = 0
        else:
            b1, b2, mode = g["momentum"], 0.0, 1
        call("sehip_opt_step_g", ptr(params), ptr(grads), ptr(self._m), ptr(self._v), params.numel(), ptr(s["sumsq"]),
             self.max_norm, g["lr"], b1, b2, g["eps"], self._step, ptr(self._step_dev), g["weight_decay"], mode, float(self.grad_scale),
             ptr(guard), stream())
        self.max_norm = 0.0

    def grad_metric(self):
        """Device tensor [2]: the reference's sqrt(sum_p (p.grad.sum())^2) (src/solver.py:494-498) and the L2 norm."""
        self._ensure_state()
        s = self._scratch
        if getattr(self, "_metric_fresh", False):          # the fused optimizer launch has written it
            self._metric_fresh = False
            return s["metric"]
        grads = self.model.flat_grads
        offs = self.model.static.layout.tensor_offsets
        fn = "sehip_grad_metric_acc" if getattr(self, "_tsums_clear", False) else "sehip_grad_metric"     # step() has just cleared tsums
        self._tsums_clear = False
        self._set0_stale = True                            # (the per-tensor sums of set 0 now hold this step's values)
        call(fn, ptr(grads), ptr(s["offsets"]), s["tsums"].numel(), int((offs[1:] - offs[:-1]).max()),
             ptr(s["sumsq"]), ptr(s["tsums"]), ptr(s["metric"]), stream())
        if self.grad_scale != 1.0:
            # metric[1] is sqrt(sumsq) of the all-reduced SUM; the mean's norm is 1/world of it (metric[0] already is: the step
            # wrote the scaled gradient back)
            s["metric"][1].mul_(float(self.grad_scale))
        return s["metric"]

    def _note_unfused_step(self):
        """The fused tail (un-pack + clipping norm + metric sums in the backward pass, one optimizer launch) is armed by zero_grad().  A loop
        that could use it but never calls zero_grad() between steps takes the separate launches every time -- correct, ~0.05 ms per DCCRN
        step slower -- and used to do so silently: say so once, at the third such step in a row."""
        import os
        import warnings
        model = self.model
        if (not hasattr(model, "_tail_sink") or getattr(self, "_unfused_warned", False) or os.environ.get("SEHIP_NO_FUSED_TAIL")
                or self.grad_scale != 1.0 or getattr(model, "grad_range_hook", None) is not None or not model.flat_params.is_cuda):
            return
        from ._lib import lib
        if lib().sehip_get_deterministic() or torch.cuda.is_current_stream_capturing():
            return
        self._unfused_run = getattr(self, "_unfused_run", 0) + 1
        if self._unfused_run >= 3:
            self._unfused_warned = True
            warnings.warn("sehip FlatOptimizer: three optimizer steps in a row took the separate un-pack / norm / update launches although the "
                          "fused tail applies to this run.  It is armed by optimizer.zero_grad() BEFORE the backward pass (the order the "
                          "reference's Solver uses, src/solver.py:487-498); a loop that never calls zero_grad() between steps keeps the slower "
                          "launches (same results).")

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=True)
        self.model._grads_live = False
        self._arm_fused_tail()

    def _arm_fused_tail(self):
        """Single replica, default (non-deterministic) schedule, not under stream capture: tell the model where the next backward
        pass may add the clipping norm's sum of squares and the metric's per-tensor sums while it un-packs the gradients
        (model._tail_sink; the DCCRN plan does: csrc/pack.hip unpack_grad_sums_kernel).  Any other situation keeps the separate
        launches of step() / grad_metric()."""
        model = self.model
        if not hasattr(model, "_tail_sink"):
            return
        model._tail_sink = None
        from ._lib import lib
        import os
        capturing = model.flat_params.is_cuda and torch.cuda.is_current_stream_capturing()
        stale0 = getattr(self, "_set0_stale", False) and getattr(self, "_set", 0) == 0 and self._scratch is not None
        if capturing:
            # The repairs below are eager one-off corrections: recorded into a hipGraph they would be replayed with every step (ADVICE
            # r5).  Stale sums can wait for the next eager zero_grad() (a capture never arms the fused tail); a counted step cannot --
            # the captured step() would record "do not count" for every replay.
            if getattr(model, "_tail_counted", False):
                raise SehipError("FlatOptimizer.zero_grad() inside a stream capture after a fused backward pass whose step() never came: "
                                 "call zero_grad() once outside the capture first")
        elif getattr(model, "_tail_counted", False) or getattr(model, "_tail_dirty", False) or stale0:
            self._ensure_state()
            s = self._scratch
            cur = "" if self._set == 0 else "1"
            if getattr(model, "_tail_counted", False):     # a fused backward pass whose step() never came: un-count it
                # (the un-pack kernel does NOT count when the guard word is set -- a hand-off time-out: only a counted step is
                #  un-counted, or the counter could reach 0, where Adam's bias correction divides by 1 - b1^0 = 0)
                gfn = getattr(model, "step_guard", None)
                guard = gfn() if gfn is not None else None
                if guard is None:
                    self._step_dev.sub_(1)
                else:
                    self._step_dev.sub_((guard.reshape(-1)[:1] == 0).to(self._step_dev.dtype))
                model._tail_counted = False
            if getattr(model, "_tail_dirty", False) or stale0:   # ... and its sums (or an unfused step's, left in set 0) are stale
                s["sumsq" + cur].zero_(); s["tsums" + cur].zero_()
                model._tail_dirty = False
                if self._set == 0:
                    self._set0_stale = False
        model._tail_done = False
        if (os.environ.get("SEHIP_NO_FUSED_TAIL") or self.grad_scale != 1.0 or getattr(model, "grad_range_hook", None) is not None
                or not model.flat_params.is_cuda or capturing or lib().sehip_get_deterministic()):
            return
        self._ensure_state()
        s = self._scratch
        cur = "" if self._set == 0 else "1"
        model._tail_sink = (ptr(s["sumsq" + cur]), ptr(s["tsums" + cur]), ptr(s["offsets"]), s["tsums"].numel(), ptr(self._step_dev))

    # ---- checkpoint format of torch.optim ----------------------------------------------------------
    def sync_step(self):
        """Refresh the host mirror of the step counter from the device (after graph replays)."""
        if getattr(self, "_step_dev", None) is not None and self._step_dev.is_cuda:
            self._step = int(self._step_dev.item())
        return self._step

    def state_dict(self):
        self._ensure_state()
        self.sync_step()
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._ensure_state()
        L = self.model.static.layout
        states = state_dict["state"]
        for idx, (name, p) in enumerate(self.model._params):
            st = states.get(idx, states.get(str(idx)))
            if st is None:
                continue
            off, shape = L.param_off[name]
            sl = slice(off, off + p.numel())
            if self.kind == "adam":
                self._m[sl].copy_(st["exp_avg"].reshape(-1))
                self._v[sl].copy_(st["exp_avg_sq"].reshape(-1))
            elif "momentum_buffer" in st and st["momentum_buffer"] is not None:
                self._m[sl].copy_(st["momentum_buffer"].reshape(-1))
            if "step" in st:   # torch.optim.SGD keeps only momentum_buffer
                self._step = int(float(st["step"]))
        self._step_dev.fill_(int(self._step))
        for g, saved in zip(self.param_groups, state_dict["param_groups"]):
            for k in ("lr", "betas", "eps", "momentum", "weight_decay"):
                if k in saved:
                    g[k] = saved[k]
        self._publish_state()
