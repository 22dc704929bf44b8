#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/second of DCCRN training (16 kHz, 2-s clips, batch 32 per GPU) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = Solver.train_step on one pre-staged synthetic batch: forward (STFT -> encoder -> complex LSTM -> decoder ->
mask -> iSTFT), SI-SNR loss, full backward, gradient all-reduce (N>1), global-norm clip and Adam -- nothing skipped.
Prints ONE JSON line (rank 0).  Also measured in the same run: the dominant kernel's roofline fraction (HIP events
around repeated launches of that exact kernel on the launch stream) and the CPU baseline (the oracle's fp32 PyTorch
restatement of the same step, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SR, CLIP_S, BATCH = 16000, 2.0, 32
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def make_batch(b, n, seed, device):
    g = torch.Generator().manual_seed(seed)
    clean = 0.1 * torch.randn(b, 1, 1, n, generator=g)
    noisy = clean[:, 0] + 0.05 * torch.randn(b, 1, n, generator=g)
    return noisy.to(device), clean.to(device)


def bench_config(length):
    from sehip.utils import dict2obj
    return dict2obj({
        "seed": 10, "root": None, "ha": None,
        "model": {"name": "dccrn", "audio_channels": 1, "num_spk": 1, "length": length,
                  "kernel_num": [16, 32, 64, 128, 256, 256], "rnn_units": 128, "masking_mode": "E"},
        "optim": {"optim": "adam", "lr": 3e-4, "beta1": 0.9, "beta2": 0.999, "loss": "si-sdr", "clip_grad": 5, "pit": False,
                  "load": False},
        "dset": {"name": "synthetic"},
        "solver": {"epochs": 1, "save_checkpoint_interval": 1000000, "all_steps": True, "total_steps": 0, "patience": 0,
                   "root": "/tmp/sehip_bench", "resume": None, "preloaded_model": None, "log_interval": 1000000,
                   "validation": {"interval": 1000000, "metric": "loss", "total_steps": 0}, "test": {"interval": 1000000}},
    })


def gemm_roofline(ws, reps=10):
    """Times every product launch of the step separately (HIP events on the launch stream, `reps` back-to-back launches
    each), groups them by the kernel instantiation libsehip picked, and returns the per-class table: launches per step,
    average launch duration (comparable with rocprofv3 --stats AverageNs of that symbol) and algorithmic TFLOP/s."""
    import ctypes as C
    from sehip._lib import call, stream, lib
    per = {}
    for name, d in ws.desc.items():
        if not name.endswith(".wg") and not d.W:
            continue  # recurrent-weight gradients exist only as wgrad launches
        fn = "sehip_wgrad" if name.endswith(".wg") else "sehip_gemm"
        flops = 2.0 * d.M * d.N * _real_k(ws, name)
        call(fn, C.byref(d), stream())
        kname = lib().sehip_last_kernel().decode()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call(fn, C.byref(d), stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        r = per.setdefault(kname, {"kernel": kname, "launches": 0, "ms": 0.0, "gflop": 0.0, "layers": []})
        r["launches"] += 1; r["ms"] += ms; r["gflop"] += flops / 1e9; r["layers"].append(name)
    rows = sorted(per.values(), key=lambda r: -r["ms"])
    for r in rows:
        r["avg_us"] = r["ms"] / r["launches"] * 1e3
        r["tflops"] = r["gflop"] / r["ms"]
    return rows


def _real_k(ws, name):
    s = ws.st.specs[name[:-3] if name.endswith(".wg") else name]
    return int((s.widx[0] >= 0).sum()) if s.N > 0 else s.K


def cpu_baseline_worker():
    """Child process: the oracle (fp32 PyTorch-CPU restatement of the reference step) on a bounded sample:
    B=4 clips, 1 warm-up + up to 5 timed steps.  Prints one JSON object."""
    from oracle import dccrn_oracle as O
    # torch CPU ops stop scaling (and then slow down) beyond a few dozen threads on these small tensors
    cores = min(torch.get_num_threads(), len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    cfg = O.DCCRNConfig(length=int(SR * CLIP_S))
    p = O.init_params(cfg, seed=10)
    adam = O.AdamState({k: v for k, v in p.items() if O.is_trainable(k)}, lr=3e-4)
    b = 4
    noisy, clean = make_batch(b, int(SR * CLIP_S), 0, "cpu")
    bases = O.stft_bases(cfg.win_len, cfg.fft_len)
    O.train_step(p, noisy, clean[:, 0], cfg, adam, clip_grad=5.0, bases=bases)
    t0 = time.time()
    n = 0
    while n < 5 and time.time() - t0 < 20.0:
        O.train_step(p, noisy, clean[:, 0], cfg, adam, clip_grad=5.0, bases=bases)
        n += 1
    dt = (time.time() - t0) / n
    print(json.dumps({"value": b * CLIP_S / dt, "unit": "audio-s/s", "cores": cores, "kind": "port",
                      "sample": f"{n} timed train steps of B={b} x 2-s clips (fp32 oracle, torch CPU, {cores} threads), "
                                f"{dt:.2f} s/step"}))


def cpu_baseline(timeout_s=180):
    """Runs the worker as a child process (own thread pool, hard timeout) so a slow host cannot stall the bench."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True,
                           text=True, timeout=timeout_s, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout / failure: report it, do not fake a number
        return {"value": None, "unit": "audio-s/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
                "sample": f"cpu baseline did not finish: {type(e).__name__}"}


def note(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as two captured hipGraphs instead of launching "
                    "every kernel from Python (measured slower: graph replay serialises the side-stream weight gradients)")
    ap.add_argument("--h2d", action="store_true", help="stage every batch from pinned host memory inside the timed region "
                    "(the PCIe-inclusive rate quoted in DESIGN.md; never the headline value)")
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker()
        return

    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    rank, world, local = distrib.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    n = int(SR * CLIP_S)
    cfg = bench_config(n)
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    opt = distrib.get_optimizer(cfg.optim, model)
    solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), device="gpu", writer=ScalarLog())
    dev = solver.device
    noisy, clean = make_batch(args.batch, n, rank, dev)
    mixture, sources = solver._prepare_batch(noisy, clean)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"model built, batch staged on {dev}; warm-up {args.warmup} steps")
    args.eager = not args.graph
    step_fn = solver.train_step if args.eager else solver.train_step_graphed
    # the dependent chain of the step runs on a high-priority stream; the weight gradients (side stream, default
    # priority) then only take the CUs the chain leaves idle
    lo, hi = torch.cuda.Stream.priority_range()
    main_stream = torch.cuda.Stream(device=dev, priority=hi) if os.environ.get("SEHIP_BENCH_PRIO", "1") == "1" else torch.cuda.current_stream()
    main_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main_stream)
    if args.h2d:
        h_mix, h_src = mixture.cpu().pin_memory(), sources.cpu().pin_memory()

    def stage():
        if args.h2d:
            mixture.copy_(h_mix, non_blocking=True)
            sources.copy_(h_src, non_blocking=True)

    for _ in range(args.warmup):
        stage()
        step_fn(mixture, sources)
    sync()
    note(f"timing {args.steps} steps ({'eager launches' if args.eager else 'hipGraph replay'})")
    t0 = time.time()
    for _ in range(args.steps):
        stage()
        loss, metric = step_fn(mixture, sources)
    sync()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / args.steps * 1e3
    note(f"{ms:.2f} ms/step")
    value = world * args.batch * CLIP_S * args.steps / dt

    out = {
        "metric": "audio-sec/sec training, DCCRN 16kHz 2s bs32", "value": value, "unit": "audio-s/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "DCCRN (kernel_num 16-32-64-128-256-256, complex LSTM 128, mask E) train step, 16 kHz 2-s "
                               "clips, SI-SNR, Adam 3e-4, clip 5", "per_gpu_batch": args.batch,
                   "global_batch": args.batch * world, "samples_per_clip": n, "parallelism": f"dp{world}",
                   "launch": "eager" if args.eager else "hipGraph", "inputs": "pinned host -> HBM every step" if args.h2d else "resident in HBM"},
        "final_loss": float(loss),
    }
    if rank == 0 and not args.no_roofline:
        ws = model.workspace(args.batch, n)
        note("per-kernel roofline pass")
        rows = gemm_roofline(ws)
        top = rows[0]
        total_ms = sum(r["ms"] for r in rows)
        total_gf = sum(r["gflop"] for r in rows)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")  # PMC passes (FETCH_SIZE / WRITE_SIZE), tools/collect_traffic.sh
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(top["kernel"], {}).get("hbm_bytes_per_launch")
        out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["tflops"], "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": top["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic,
                           "launches_per_step": top["launches"], "avg_launch_us": top["avg_us"],
                           "gflop_per_launch": top["gflop"] / top["launches"],
                           "all_product_kernels": {"ms_per_step": total_ms, "tflops": total_gf / total_ms,
                                                   "frac": total_gf / total_ms / PEAK_BF16_TFLOPS}}
        out["step_tflops"] = 45.96e9 * args.batch / (ms * 1e-3) / 1e12
        out["kernel_classes"] = [{"kernel": r["kernel"], "launches": r["launches"], "avg_us": round(r["avg_us"], 1),
                                  "tflops": round(r["tflops"], 1)} for r in rows[:10]]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (oracle)")
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
