"""Where the HOST time of one training step goes (cProfile over N eager steps; GPU box only).
    python tools/host_profile.py [--steps 100]"""
import argparse, cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (re-uses the bench's model / solver construction)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    a = ap.parse_args()
    solver, model, mixture, sources = bench.build_for_profile()
    for _ in range(5):
        solver.train_step(mixture, sources)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    for _ in range(a.steps):
        solver.train_step(mixture, sources)
    pr.disable()
    t1 = time.time()
    torch.cuda.synchronize()
    print(f"host {1e3 * (t1 - t0) / a.steps:.2f} ms/step (with profiler), total {1e3 * (time.time() - t0) / a.steps:.2f}")
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
    st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
