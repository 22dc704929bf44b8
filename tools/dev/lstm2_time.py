"""dev: kernel-level timing of the fused LSTM launches (HIP events), incl. the consumer's native speed (granules already valid:
the same epoch twice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sehip._lib import call, ptr, stream
from sehip.model import DCCRN

B, N = 32, 32000
torch.manual_seed(3)
model = DCCRN(rnn_units=128, kernel_num=[16, 32, 64, 128, 256, 256], length=N).cuda().train()
x = (0.1 * torch.randn(B, 1, N)).cuda()
out = model(x); out.backward(torch.ones_like(out) * 1e-3); torch.cuda.synchronize()
ws = model.workspace(B, N)
b, st, tb, T, h = ws.bufs, ws.st, ws.tb, ws.T, 64
wp = tb.wpack.data_ptr()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

ep = [100]
def fwd(new_epoch=True):
    if new_epoch: ep[0] = ep[0] % 60000 + 1
    call("sehip_lstm2_fwd", b["pre1_r"].ptr, b["pre1_i"].ptr, wp + 2 * st.whh_off[1], wp + 2 * st.whh_off[2], wp + 2 * st.wih2_off,
         ws.desc["ih2_r"].bias, B, T, h, b["h1"].ptr, b["gates1"].ptr, b["c1"].ptr, b["h2"].ptr, b["gates2"].ptr, b["c2"].ptr,
         ptr(ws.l2_gran_f), ptr(ws.l2_sync), ep[0], stream())
def bwd(new_epoch=True):
    if new_epoch: ep[0] = ep[0] % 60000 + 1
    call("sehip_lstm2_bwd", b["dxo_r"].ptr, b["dxo_i"].ptr, wp + 2 * st.whhT_off[1], wp + 2 * st.whhT_off[2], wp + 2 * st.wihT2_off,
         b["gates1"].ptr, b["c1"].ptr, b["gates2"].ptr, b["c2"].ptr, B, T, h, b["dpre1_r"].ptr, b["dpre1_i"].ptr, b["dpre2_r"].ptr,
         b["dpre2_i"].ptr, ptr(ws.l2_gran_b), ptr(ws.l2_sync), ep[0], stream())
print("fused fwd, fresh epoch   :", f"{timeit(fwd):.1f} us")
print("fused fwd, same epoch    :", f"{timeit(lambda: fwd(False)):.1f} us   (consumers never wait)")
print("fused bwd, fresh epoch   :", f"{timeit(bwd):.1f} us")
print("fused bwd, same epoch    :", f"{timeit(lambda: bwd(False)):.1f} us")
print("old lstm_fwd layer 1     :", f"{timeit(lambda: ws._lstm_fwd_call(1, 0, T, stream())):.1f} us")
print("old lstm_fwd layer 2     :", f"{timeit(lambda: ws._lstm_fwd_call(2, 0, T, stream())):.1f} us")
print("old lstm_bwd layer 2     :", f"{timeit(lambda: ws._lstm_bwd_call(2, 0, T, stream())):.1f} us")
print("old lstm_bwd layer 1     :", f"{timeit(lambda: ws._lstm_bwd_call(1, 0, T, stream())):.1f} us")
print("gemm_pair ih1            :", f"{timeit(lambda: ws.gemm_pair('ih1_r', 'ih1_i')):.1f} us")
print("gemm_pair ih2            :", f"{timeit(lambda: ws.gemm_pair('ih2_r', 'ih2_i')):.1f} us")
print("gemm_pair dx2            :", f"{timeit(lambda: ws.gemm_pair('dx2_r', 'dx2_i')):.1f} us")
print("tmo", int(ws.l2_sync[0]))
