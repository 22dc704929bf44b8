"""GPU: the two stacked complex LSTM layers as ONE persistent launch per direction (csrc/lstm2.hip, round 4) against the two launches per
direction of round 3 (csrc/lstm.hip + the ih2 / dx2 products), on the same model and input.  Reference math:
src/model/dccrn.py:264-302 (NavieComplexLSTM), :170-191 (how DCCRN wires the two layers).  The oracle comparison of the whole LSTM block
at the headline size (T = 323, B = 4 and the B = 32 step) runs on the fused path by default: tests/test_gpu_c1_fullsize.py."""
import os

import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
KEYS = ("P", "h1", "h2", "gates1", "gates2", "c1", "c2", "dz5l", "dpre1_r", "dpre1_i", "dpre2_r", "dpre2_i")


def run_once(fuse, B, N, kernel_num=(16, 16, 32, 32, 64, 64), hook=None, steps=1):
    from sehip.model import DCCRN
    names = ("SEHIP_NO_FUSE_STATS", "SEHIP_NO_FUSE_FINALIZE", "SEHIP_NO_LSTM_FUSE")
    old = {k: os.environ.get(k) for k in names}
    # a bit-reproducible network around the LSTM (see tests/test_gpu_lstm_chunks.py)
    os.environ.update({"SEHIP_NO_FUSE_STATS": "1", "SEHIP_NO_FUSE_FINALIZE": "1"})
    if fuse:
        os.environ.pop("SEHIP_NO_LSTM_FUSE", None)
    else:
        os.environ["SEHIP_NO_LSTM_FUSE"] = "1"
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(3)
        model = DCCRN(rnn_units=128, kernel_num=list(kernel_num), length=N).to(dev).train()
        g = torch.Generator().manual_seed(11)
        x = (0.1 * torch.randn(B, 1, N, generator=g)).to(dev)
        ws = model.workspace(B, N)
        assert ws.lstm_fused == bool(fuse)
        if hook is not None:
            hook(ws)
        for _ in range(steps):
            model.zero_grad()
            for _, prm in model._params:
                prm.grad = None
            model._grads_live = False
            out = model(x)
            out.backward(torch.ones_like(out) * 1e-3)
        torch.cuda.synchronize()
        keep = {k: ws.bufs[k].t.float().cpu().clone() for k in KEYS}
        return keep, model.flat_grads.detach().cpu().clone(), ws, model
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("B,N", [(2, 6000), (3, 4000), (17, 3000), (32, 32000)])
def test_fused_equals_the_two_launches(B, N):
    """Forward, through the model: layer 1 is the same arithmetic in the same order -- bit-equal h1 / gates / c.  Layer 2 sees
    x2 = bf16(h1_a +- h1_b) as ONE operand instead of the two bf16 operands of the ih2 product (one more rounding of its input, 2^-9
    relative): h2 / c2 / the projection equal to a few 1e-3.
    Backward, in isolation (the fused kernel on the UNFUSED run's own records and upstream gradient -- through the whole network the
    0.2 % difference of P moves the decoder's PReLU branches and with them the gradient that arrives here by a few per cent, which
    says nothing about these kernels): layer 2's gate gradients are the same arithmetic -- bit-equal; layer 1 receives dx2 as the fp32
    sum of two partials instead of the bf16 output of the dx2 product: equal to a few 1e-3.
    Ragged batch tiles (B = 2, 3, 17) and the headline shape (B = 32, T = 323)."""
    from sehip import _lib
    kn = (16, 32, 64, 128, 256, 256) if N == 32000 else (16, 16, 32, 32, 64, 64)
    a, ga, wa, _ = run_once(True, B, N, kn, steps=2)        # (two calls: the second runs on granule arrays the first has written)
    b, gb, wb, _ = run_once(False, B, N, kn, steps=2)
    assert int(wa.l2_sync[0]) == 0
    T = wa.T
    for k in ("h1", "gates1", "c1"):
        assert torch.equal(a[k], b[k]), k
    errs = {k: rel_err(a[k], b[k]) for k in ("P", "h2", "gates2", "c2")}
    print(f"fused vs two launches, forward, B={B} T={T}:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["h2"] < 5e-3 and errs["c2"] < 5e-3 and errs["P"] < 5e-3 and errs["gates2"] < 5e-3
    assert rel_err(ga, gb) < 3e-2                        # whole-network parameter gradients (see above: the decoder's kinks)
    # backward in isolation, on the unfused workspace
    lib = _lib.lib()
    names = ("dpre1_r", "dpre1_i", "dpre2_r", "dpre2_i", "dz5l")
    ref = {k: wb.bufs[k].t.float().cpu().clone() for k in names}
    dev = wb.bufs["h1"].t.device
    wb.l2_gran_f = torch.zeros(int(lib.sehip_lstm2_gran_bytes(B, T, 0)) // 8, dtype=torch.int64, device=dev)
    wb.l2_gran_b = torch.zeros(int(lib.sehip_lstm2_gran_bytes(B, T, 1)) // 8, dtype=torch.int64, device=dev)
    wb.l2_sync = torch.zeros(int(lib.sehip_lstm2_sync_bytes()) // 4, dtype=torch.int32, device=dev)
    wb.lstm_fused, wb._l2_cur_epoch = True, 7
    for rep in range(2):                                 # the second call reads granule slots the first has written
        wb._l2_cur_epoch = 7 + rep
        wb._lstm_backward(B, T, 64)
    torch.cuda.synchronize()
    wb.lstm_fused = False
    assert int(wb.l2_sync[0]) == 0
    got = {k: wb.bufs[k].t.float().cpu() for k in names}
    for k in ("dpre2_r", "dpre2_i"):
        assert torch.equal(got[k], ref[k]), k
    berr = {k: rel_err(got[k], ref[k]) for k in ("dpre1_r", "dpre1_i", "dz5l")}
    print(f"fused vs two launches, backward in isolation, B={B} T={T}:", {k: f"{v:.2e}" for k, v in berr.items()})
    assert max(berr.values()) < 6e-3, berr


def test_forced_handoff_timeout_sets_the_guard_and_falls_back(tmp_path):
    """The hand-off waits are bounded: word 1 of the sync block (test hook) makes the first wait of a consumer give up; the sticky
    word 0 is set, the fused optimizer's device-side guard skips the parameter update, the Solver's health check returns this
    workspace to the two launches per direction, and training continues."""
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    from test_gpu_solver import solver_config, make_batch
    cfg = solver_config(tmp_path)
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    solver = Solver(cfg, model, distrib.get_optimizer(cfg.optim, model), distrib.get_loss_function(cfg.optim), device="gpu",
                    writer=ScalarLog())
    noisy, clean = make_batch(31, 2, 4000)
    mix, src = solver._prepare_batch(noisy, clean)
    ws = model.workspace(2, 4000)
    if not ws.lstm_fused:
        pytest.skip("fused LSTM disabled in this environment")
    p0 = model.flat_params.detach().clone()
    ws.l2_sync[1] = -1
    with pytest.warns(UserWarning, match="hand-off wait"):
        solver.train_step(mix, src)
        torch.cuda.synchronize()
        assert int(ws.l2_sync[0]) != 0, "the forced time-out did not fire"
        assert torch.equal(model.flat_params.detach(), p0), "an optimizer step was applied after a hand-off time-out"
        assert int(solver.optimizer._step_dev.item()) == 0
        solver._model_health()
    assert not ws.lstm_fused and int(ws.l2_sync[0]) == 0 and getattr(solver, "lost_steps", 0) == 1
    ws.l2_sync[1] = 0
    loss, _ = solver.train_step(mix, src)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and not torch.equal(model.flat_params.detach(), p0)
    assert int(solver.optimizer._step_dev.item()) == 1


def test_eager_forward_after_a_graph_replay_does_not_accept_the_replays_granules():
    """a hipGraph freezes the granule-tag epoch it was captured with and only the graph clears the granule arrays; an
    eager forward on the same workspace afterwards used to draw the SAME epoch (the counter did not advance under capture), so
    layer 2 accepted the replay's stale h1 granules at once and ran ahead of layer 1 -- silently wrong.  Captured launches now own
    epoch 65535 and eager calls cycle through 1 .. 65534: the eager forward after a replay must equal the unfused launches."""
    from sehip.model import DCCRN
    from sehip import plan
    dev = torch.device("cuda:0")
    B, N = 3, 6000
    g = torch.Generator().manual_seed(5)
    x1 = (0.1 * torch.randn(B, 1, N, generator=g)).to(dev)
    x2 = (0.1 * torch.randn(B, 1, N, generator=g)).to(dev)

    def build(fuse):
        old = os.environ.get("SEHIP_NO_LSTM_FUSE")
        if fuse:
            os.environ.pop("SEHIP_NO_LSTM_FUSE", None)
        else:
            os.environ["SEHIP_NO_LSTM_FUSE"] = "1"
        try:
            torch.manual_seed(3)
            m = DCCRN(rnn_units=128, kernel_num=[16, 16, 32, 32, 64, 64], length=N).to(dev).eval()
            ws = m.workspace(B, N)
            assert ws.lstm_fused == bool(fuse)
            return m, ws
        finally:
            if old is None:
                os.environ.pop("SEHIP_NO_LSTM_FUSE", None)
            else:
                os.environ["SEHIP_NO_LSTM_FUSE"] = old
    ref_model, _ = build(False)
    with torch.no_grad():
        ref1, ref2 = ref_model(x1).clone(), ref_model(x2).clone()
    model, ws = build(True)
    ws.pinned = True
    static_x = x1.clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad():
        with torch.cuda.graph(graph):            # captured on the FRESH workspace: no eager call has advanced its epoch yet
            y_static = model(static_x)
    assert ws.l2_epoch == 0
    graph.replay()
    torch.cuda.synchronize()
    assert rel_err(y_static, ref1) < 2e-2
    with torch.no_grad():
        y_eager = model(x2).clone()              # eager, same workspace, a DIFFERENT input: stale granules would show
    torch.cuda.synchronize()
    assert ws.l2_epoch == 1 and plan.L2_GRAPH_EPOCH == 65535 and int(ws.l2_sync[0]) == 0
    assert rel_err(y_eager, ref2) < 2e-2, rel_err(y_eager, ref2)
    graph.replay()                               # and the graph still works after the eager call
    torch.cuda.synchronize()
    assert rel_err(y_static, ref1) < 2e-2
    # the eager epochs never reach the graphs' one
    ws.l2_epoch = 65533
    assert ws._l2_next_epoch() == 65534 and ws._l2_next_epoch() == 1
