"""The single-replica tail of the train step in two launches (sehip_unpack_grad_sums + sehip_opt_step_m: un-pack with the clipping
norm's sum of squares and the metric's per-tensor sums, then update + logged metrics + clearing of the next step's accumulators)
against the separate launches (sehip_unpack_grad, sehip_opt_begin, sehip_grad_sumsq, sehip_opt_step, sehip_grad_metric) on the SAME
packed gradient buffer and the same optimizer state: the un-packed gradients must be bit-equal, everything else equal up to the
order of the sums.  HIP vs HIP at the operator level (whole steps differ from run to run by the default schedule's atomics; the
oracle comparison of the whole step -- tests/test_gpu_solver.py -- runs the fused tail by default)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("clip", [0.0, 5.0, 0.05])
@pytest.mark.parametrize("kind,perm", [("adam", False), ("sgd", False), ("adam", True)])
def test_fused_tail_matches_the_separate_launches(clip, kind, perm):
    from sehip.model import DCCRN
    from sehip.optim import FlatOptimizer
    from sehip._lib import call, ptr, stream
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    model = DCCRN(kernel_num=[16, 16, 32, 32, 64, 64], rnn_units=128, length=4000).to(dev).train()
    x = (0.1 * torch.randn(2, 1, 4000)).to(dev)
    model(x).sum().backward()                      # builds the workspace, its tables and the flat gradient buffer
    torch.cuda.synchronize()
    ws = model.workspace(2, 4000)
    n = model.flat_params.numel()
    gen = torch.Generator(device=dev).manual_seed(7)
    ws.gpack.copy_(torch.randn(ws.gpack.shape, device=dev, generator=gen) * 3e-2)
    p0 = model.flat_params.detach().clone()
    res = []
    for fused in (False, True):
        model.flat_params.data.copy_(p0)
        opt = FlatOptimizer(model, lr=3e-3, kind=kind, momentum=0.9 if kind == "sgd" else 0.0)
        opt._ensure_state()
        opt._m.normal_(generator=gen).mul_(1e-2); opt._v.uniform_(1e-6, 1e-4, generator=gen)
        if fused and res:
            opt._m.copy_(res[0]["m0"]); opt._v.copy_(res[0]["v0"])
        m0, v0 = opt._m.clone(), opt._v.clone()
        opt._step = 3; opt._step_dev.fill_(3)
        grads = model.flat_grads
        grads.zero_()
        s = opt._scratch
        if fused and perm:         # rows of every tensor in gather order (sehip_unpack_grad_sums_perm, round 6: what the plans launch)
            assert ws.tb.uperm is not None
            call("sehip_unpack_grad_sums_perm", ptr(ws.gpack), ptr(ws.tb.utab_g), ptr(ws.tb.uperm), n, ptr(grads), ptr(s["offsets"]),
                 s["tsums"].numel(), ptr(s["sumsq"]), ptr(s["tsums"]), ptr(opt._step_dev), None, stream())
            model._tail_done = True
        elif fused:
            call("sehip_unpack_grad_sums", ptr(ws.gpack), ptr(ws.tb.utab), n, ptr(grads), ptr(s["offsets"]), s["tsums"].numel(),
                 ptr(s["sumsq"]), ptr(s["tsums"]), ptr(opt._step_dev), None, stream())
            model._tail_done = True
        else:
            call("sehip_unpack_grad", ptr(ws.gpack), ptr(ws.tb.utab), n, ptr(grads), stream())
            model._tail_done = False
        g_unpacked = grads.clone()
        if clip:
            opt.clip_grad_norm_(clip)
        opt.step()
        metric = opt.grad_metric().clone()
        torch.cuda.synchronize()
        assert opt.sync_step() == 4
        if fused:     # the launch cleared the OTHER set for the next step and left this one's sums in place
            assert float(s["sumsq1"][0]) == 0.0 and float(s["tsums1"].abs().max()) == 0.0 and opt._set == 1
        res.append(dict(g=g_unpacked, p=model.flat_params.detach().clone(), m=opt._m.clone(), v=opt._v.clone(), metric=metric,
                        gc=grads.clone(), m0=m0, v0=v0))
    a, b = res
    assert torch.equal(a["g"], b["g"])
    for key, tol in (("p", 1e-6), ("m", 1e-6), ("v", 1e-6), ("gc", 1e-6)):
        err = float((a[key] - b[key]).norm() / a[key].norm())
        assert err < tol, (key, clip, kind, err)
    # metric[0]: the reference's sqrt(sum_t (sum g_t)^2) of the clipped gradient; metric[1]: the pre-clip L2 norm (the separate launches
    # only take it when clipping is on; the fused un-pack always has it)
    assert abs(float(a["metric"][0]) - float(b["metric"][0])) < 1e-5 * float(a["metric"][0]) and float(b["metric"][0]) > 0
    if clip:
        assert abs(float(a["metric"][1]) - float(b["metric"][1])) < 1e-5 * float(a["metric"][1])
    assert abs(float(b["metric"][1]) - float(a["g"].double().norm())) < 1e-5 * float(b["metric"][1])


def test_accumulation_and_skipped_steps_keep_the_counter_and_the_sums_right():
    """The fused un-pack advances the device step counter and fills the accumulators during backward(); a second, accumulating
    backward pass invalidates its sums (step() then takes them from the final buffer and does not count twice), and a backward
    pass whose step() never comes is un-counted by the next zero_grad()."""
    import numpy as np
    from sehip.model import DCCRN
    from sehip.optim import FlatOptimizer
    from sehip.loss import loss_sisdr
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    model = DCCRN(kernel_num=[16, 16, 32, 32, 64, 64], rnn_units=128, length=4000).to(dev).train()
    opt = FlatOptimizer(model, lr=3e-4)
    g = torch.Generator().manual_seed(4)
    def batch():
        c = (0.1 * torch.randn(2, 1, 4000, generator=g)).to(dev)
        return c + (0.05 * torch.randn(2, 1, 4000, generator=g)).to(dev), c
    offs = model.static.layout.tensor_offsets

    def ref_metric():
        gr = model.flat_grads.double().cpu().numpy()
        return float(np.sqrt(sum(gr[offs[t]:offs[t + 1]].sum() ** 2 for t in range(len(offs) - 1))))
    # two backward passes, one step
    opt.zero_grad()
    for _ in range(2):
        n, c = batch()
        loss_sisdr(model(n), c).backward()
    assert model._tail_done is False
    opt.clip_grad_norm_(5.0)
    opt.step()
    m = opt.grad_metric()
    torch.cuda.synchronize()
    assert opt.sync_step() == 1
    assert abs(float(m[0]) - ref_metric()) < 1e-4 * ref_metric()
    # a backward pass without a step, then a normal (fused) step
    opt.zero_grad()
    n, c = batch()
    loss_sisdr(model(n), c).backward()
    assert model._tail_done is True
    opt.zero_grad()
    n, c = batch()
    loss_sisdr(model(n), c).backward()
    opt.clip_grad_norm_(5.0)
    opt.step()
    m = opt.grad_metric()
    torch.cuda.synchronize()
    assert opt.sync_step() == 2
    assert abs(float(m[0]) - ref_metric()) < 1e-4 * ref_metric()


@pytest.mark.parametrize("which", ["dcunet", "convtasnet", "demucs"])
def test_fused_tail_on_the_other_flat_models(which):
    """DCUnet and ConvTasNet un-pack through sehip_unpack_grad_sums as well, Demucs through sehip_unpack_grad1_sums: the step counter, the clipping norm and the logged
    per-tensor metric of a fused step against the same quantities recomputed from the un-packed gradient buffer; an accumulating
    second pass falls back to the separate launches."""
    import numpy as np
    from sehip.optim import FlatOptimizer
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    if which == "dcunet":
        from sehip.model import DCUnet
        model = DCUnet(data_type=True, model_complexity=8, model_depth=10).to(dev).train()
        make = lambda: torch.randn(2, 1, 257, 33, 2, device=dev)
    elif which == "demucs":       # one-entry un-pack tables + the list of multi-entry parameters (sehip_unpack_grad1_sums / _list_sums)
        from sehip.model import Demucs
        model = Demucs(sources=["a", "b"], audio_channels=1, channels=32, depth=4).to(dev).train()
        make = lambda: 0.3 * torch.randn(2, 1, 9000, device=dev)
    else:
        from sehip.model import ConvTasNet
        model = ConvTasNet(sources=["None", "None"], N=16, L=8, B=16, H=32, P=3, X=3, R=2, audio_channels=1).to(dev).train()
        make = lambda: 0.3 * torch.randn(2, 1, 404, device=dev)
    opt = FlatOptimizer(model, lr=3e-4)
    offs = model.static.layout.tensor_offsets

    def ref():
        gr = model.flat_grads.double().cpu().numpy()
        return (float(np.sqrt(sum(gr[offs[t]:offs[t + 1]].sum() ** 2 for t in range(len(offs) - 1)))), float(np.sqrt((gr ** 2).sum())))
    opt.zero_grad()
    model(make()).pow(2).mean().backward()
    assert model._tail_done is True
    opt.clip_grad_norm_(1e9)
    opt.step()
    m = opt.grad_metric()
    torch.cuda.synchronize()
    metric, total = ref()
    assert opt.sync_step() == 1
    assert abs(float(m[0]) - metric) < 1e-4 * metric + 1e-12
    assert abs(float(m[1]) - total) < 1e-4 * total          # the pre-clip L2 norm the clipping used
    opt.zero_grad()
    for _ in range(2):
        model(make()).pow(2).mean().backward()
    assert model._tail_done is False
    opt.clip_grad_norm_(1e9)
    opt.step()
    m = opt.grad_metric()
    torch.cuda.synchronize()
    metric, _ = ref()
    assert opt.sync_step() == 2
    assert abs(float(m[0]) - metric) < 1e-4 * metric + 1e-12


def test_step_then_zero_grad_order_does_not_add_to_the_previous_steps_sums():
    """The common loop order `backward(); step(); zero_grad()`: step 1 is unfused (nothing armed the tail yet) and leaves
    its sums in set 0; the zero_grad() behind it arms the fused tail on set 0, which must start from zero -- otherwise step 2 clips
    with sqrt(|g1|^2 + |g2|^2) and logs a wrong metric.  Also an unfused step in the middle of fused ones (the eager step after graph
    replays), with the set index at 0 and at 1."""
    import numpy as np
    from sehip.model import DCCRN
    from sehip.optim import FlatOptimizer
    from sehip.loss import loss_sisdr
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = DCCRN(kernel_num=[16, 16, 32, 32, 64, 64], rnn_units=128, length=4000).to(dev).train()
    opt = FlatOptimizer(model, lr=3e-4)
    g = torch.Generator().manual_seed(11)
    offs = model.static.layout.tensor_offsets

    def batch():
        c = (0.1 * torch.randn(2, 1, 4000, generator=g)).to(dev)
        return c + (0.05 * torch.randn(2, 1, 4000, generator=g)).to(dev), c

    def ref():
        gr = model.flat_grads.double().cpu().numpy()
        return (float(np.sqrt(sum(gr[offs[t]:offs[t + 1]].sum() ** 2 for t in range(len(offs) - 1)))), float(np.sqrt((gr ** 2).sum())))
    fused_seen = []
    for it in range(6):
        n, c = batch()
        loss_sisdr(model(n), c).backward()
        if it == 3:                       # an unfused step between fused ones: drop the armed tail's result
            model._tail_done = False
            model._tail_counted = True    # (the un-pack has counted the step; step() must not count again)
        fused_seen.append(bool(model._tail_done))
        torch.cuda.synchronize()
        metric, total = ref()             # the un-packed, unclipped gradient of THIS step
        opt.clip_grad_norm_(1e9)          # (a clip that never bites: the buffer stays comparable; metric[1] is the norm the clip used)
        opt.step()
        m = opt.grad_metric()
        torch.cuda.synchronize()
        assert abs(float(m[1]) - total) < 1e-4 * total, (it, float(m[1]), total)
        assert abs(float(m[0]) - metric) < 1e-4 * metric + 1e-9, (it, float(m[0]), metric)
        assert opt.sync_step() == it + 1
        opt.zero_grad()                   # AFTER the step: arms the next backward pass
    assert fused_seen == [False, True, True, False, True, True]


def test_a_loop_that_never_arms_the_fused_tail_is_told_so_once():
    """the fused tail is armed by zero_grad(); a loop that never calls it between steps silently took the separate
    launches.  It still does (same results) -- and says so once, at the third such step in a row; the Solver's order never hears it."""
    import warnings
    from sehip.model import DCCRN
    from sehip.optim import FlatOptimizer
    from sehip.loss import loss_sisdr
    dev = torch.device("cuda:0")

    def loop(call_zero_grad, steps=5):
        torch.manual_seed(4)
        model = DCCRN(kernel_num=[16, 16, 32, 32, 64, 64], rnn_units=128, length=4000).to(dev).train()
        opt = FlatOptimizer(model, lr=3e-4)
        g = torch.Generator().manual_seed(12)
        c = (0.1 * torch.randn(2, 1, 4000, generator=g)).to(dev)
        n = c + (0.05 * torch.randn(2, 1, 4000, generator=g)).to(dev)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for _ in range(steps):
                if call_zero_grad:
                    opt.zero_grad()
                else:
                    model.flat_grads.zero_()          # (a caller that clears the gradients some other way)
                loss_sisdr(model(n), c).backward()
                opt.clip_grad_norm_(5.0)
                opt.step()
            torch.cuda.synchronize()
        return [x for x in w if "fused tail" in str(x.message)]

    assert len(loop(False)) == 1
    assert len(loop(True)) == 0
