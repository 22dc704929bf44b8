"""dev: fused vs unfused LSTM block: errors per buffer and event timings of the LSTM section alone (no side-stream contention)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_lstm_fused import run_once, KEYS
from util import rel_err

for B, N, kn in ((2, 6000, (16, 16, 32, 32, 64, 64)), (32, 32000, (16, 32, 64, 128, 256, 256))):
    a, ga, wa, ma = run_once(True, B, N, kn, steps=2)
    b, gb, wb, mb = run_once(False, B, N, kn, steps=2)
    print(B, N, "tmo", int(wa.l2_sync[0]), {k: f"{rel_err(a[k], b[k]):.2e}" for k in KEYS}, "grads", f"{rel_err(ga, gb):.2e}")
    # error of dpre2 by time step
    d = (a["dpre2_r"] - b["dpre2_r"]).reshape(B, wa.T, -1).norm(dim=(0, 2)) / (b["dpre2_r"].reshape(B, wa.T, -1).norm(dim=(0, 2)) + 1e-30)
    print("dpre2_r err by t (every 20):", [f"{float(v):.1e}" for v in d[::20]])
    # backward in isolation: the fused kernel on the UNFUSED run's records and upstream gradient
    import torch as _t
    from sehip import _lib
    lib_ = _lib.lib()
    T = wb.T
    ref = {k: wb.bufs[k].t.float().cpu().clone() for k in ("dpre1_r", "dpre1_i", "dpre2_r", "dpre2_i", "dz5l")}
    wb.l2_gran_f = _t.zeros(int(lib_.sehip_lstm2_gran_bytes(B, T, 0)) // 8, dtype=_t.int64, device="cuda")
    wb.l2_gran_b = _t.zeros(int(lib_.sehip_lstm2_gran_bytes(B, T, 1)) // 8, dtype=_t.int64, device="cuda")
    wb.l2_sync = _t.zeros(16, dtype=_t.int32, device="cuda")
    wb.lstm_fused = True
    wb._l2_cur_epoch = 7
    wb._lstm_backward(B, T, 64)
    _t.cuda.synchronize()
    print("isolated bwd:", {k: f"{rel_err(wb.bufs[k].t.float().cpu(), ref[k]):.2e}" for k in ref}, "tmo", int(wb.l2_sync[0]))
    wb.lstm_fused = False
    for name, ws in (("fused", wa), ("two launches", wb)):
        T, h = ws.T, 64
        for fn, label in ((lambda: ws._lstm_forward(B, T, h), "fwd"), (lambda: ws._lstm_backward(B, T, h), "bwd")):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"  {name} {label}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
