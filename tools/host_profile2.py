"""Host cost of the backward pass when it is called directly (autograd's device thread hides it from cProfile in host_profile.py):
cProfile over ws.backward() + the optimizer tail, main thread.  GPU box only."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

solver, model, mixture, sources = bench.build_for_profile()
for _ in range(5):
    solver.train_step(mixture, sources)
torch.cuda.synchronize()
ws = model.workspace(mixture.shape[0], mixture.shape[-1])
g = torch.randn(ws.B, ws.length, device=mixture.device) * 1e-3
N = 30
t0 = time.time()
for _ in range(N):
    ws.forward(mixture.reshape(ws.B, -1), model._flat, model._bflat, model._nbt, training=True)
t1 = time.time()
torch.cuda.synchronize()
pr = cProfile.Profile()
t2 = time.time()
pr.enable()
for _ in range(N):
    ws.backward(g, model._flat, model.flat_grads)
pr.disable()
t3 = time.time()
torch.cuda.synchronize()
print(f"forward host {1e3 * (t1 - t0) / N:.3f} ms, backward host {1e3 * (t3 - t2) / N:.3f} ms per call (profiler on)")
t2 = time.time()
for _ in range(N):
    ws.backward(g, model._flat, model.flat_grads)
t3 = time.time()
torch.cuda.synchronize()
print(f"backward host {1e3 * (t3 - t2) / N:.3f} ms per call (profiler off)")
t2 = time.time()
for _ in range(N):
    solver.train_step(mixture, sources)
t3 = time.time()
torch.cuda.synchronize()
print(f"train_step host {1e3 * (t3 - t2) / N:.3f} ms per call")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
