#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/second of DCCRN training (16 kHz, 2-s clips, batch 32 per GPU) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... -- or
     plain: without WORLD_SIZE in the environment bench.py starts the N ranks itself as child processes and relays rank 0's line;
     fewer than N visible devices is a named error, "needs N devices")

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


HEADLINE_KERNEL_NUM = [16, 32, 64, 128, 256, 256]
KERNEL_NUM = list(HEADLINE_KERNEL_NUM)      # --kernel-num replaces it (a side measurement: other widths of the same network)
RNN_UNITS = [128]                           # --rnn-units replaces it (a side measurement too: 256 = the DCCRN paper's complex LSTM)


def bench_config(length):
    from sehip.utils import dict2obj
    return dict2obj({
        "seed": 10, "root": None, "ha": None,
        "model": {"name": "dccrn", "audio_channels": 1, "num_spk": 1, "length": length,
                  "kernel_num": list(KERNEL_NUM), "rnn_units": RNN_UNITS[0], "masking_mode": "E"},
        "optim": {"optim": "adam", "lr": 3e-4, "beta1": 0.9, "beta2": 0.999, "loss": "si-sdr", "clip_grad": 5, "pit": False,
                  "load": False},
        "dset": {"name": "synthetic"},
        "solver": {"epochs": 1, "save_checkpoint_interval": 1000000, "all_steps": True, "total_steps": 0, "patience": 0,
                   "root": "/tmp/sehip_bench", "resume": None, "preloaded_model": None, "log_interval": 1000000,
                   "validation": {"interval": 1000000, "metric": "loss", "total_steps": 0}, "test": {"interval": 1000000}},
    })


def dcunet_config(complexity=45):
    """BASELINE config C2: DCUnet-10 (the reference has no depth 16: SURVEY section 0.1), complex, n_fft 512 / hop 128, mse in
    the STFT domain (src/solver.py:454-480), Adam 3e-4, clip 5."""
    from sehip.utils import dict2obj
    cfg = bench_config(0)
    d = dict2obj({"name": "dcunet", "audio_channels": 1, "num_spk": 1, "n_fft": 512, "hop_length": 128, "win_length": 512,
                  "center": True, "model_complexity": complexity, "model_depth": 10, "data_type": True, "padding_mode": "zeros"})
    cfg.model = d
    cfg.optim.loss = "mse"
    return cfg


def convtasnet_config():
    """BASELINE config C4: ConvTasNet (N128 L40 B128 H256 P3 X7 R2, gLN), 2-speaker separation, SI-SNR, 8 kHz 4-s clips, batch 32."""
    from sehip.utils import dict2obj
    cfg = bench_config(0)
    cfg.model = dict2obj({"name": "conv-tasnet", "audio_channels": 1, "num_spk": 2, "sources": ["None", "None"], "skip": False,
                          "sample_rate": 8000, "segment": 4})
    return cfg


def demucs_config():
    """BASELINE config C3: Demucs denoiser (channels 64, depth 6, the constructor defaults), 48 kHz stereo 2-s clips, one source, batch 16."""
    from sehip.utils import dict2obj
    cfg = bench_config(0)
    cfg.model = dict2obj({"name": "demucs", "audio_channels": 2, "num_spk": 1, "sources": ["clean"], "samplerate": 48000, "segment": 2})
    return cfg


def gemm_roofline(ws, reps=5):
    """Times every product launch of the step separately, groups them by the kernel instantiation libsehip picked, and
    returns the per-class table: launches per step, average launch duration and algorithmic TFLOP/s.
    Each timing is ONE launch between two HIP events on the launch stream, after a 320 MB write has pushed the launch's
    inputs out of the 256 MB Infinity Cache: back-to-back repetitions of one descriptor (round 1) re-read 21-42 MB inputs
    from that cache and came out ~5 % below the rocprofv3 AverageNs of the same symbol in the serial step."""
    import ctypes as C
    from sehip._lib import call, stream, lib
    flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device=ws.device)
    per = {}
    # the launches as the step issues them where the workspace says so (DCCRN: pairs through sehip_gemm_pair / sehip_wgrad_pair, the
    # LSTM weight gradients as their one grouped launch); every descriptor singly otherwise
    if hasattr(ws, "launch_units"):
        units = ws.launch_units()
    else:
        units = []
        for name, d in ws.desc.items():
            if not name.endswith(".wg") and not d.W:
                continue  # recurrent-weight gradients exist only as wgrad launches
            fn = "sehip_wgrad" if name.endswith(".wg") else "sehip_gemm"
            if name.endswith(".wg") and hasattr(ws, "_launch_wgrad"):
                # (Demucs: the launch the step makes -- the streaming dense-row kernel where the workspace bound one)
                units.append((name, [name], lambda st, nm=name[:-3]: ws._launch_wgrad(nm, st)))
            else:
                units.append((name, [name], lambda st, fn=fn, d=d: call(fn, C.byref(d), st)))
    queue = list(units)
    while queue:
        name, members, launch = queue.pop(0)
        flops = sum(2.0 * ws.desc[m].M * _weight_entries(ws, m) for m in members)   # algorithmic: padding channels / padded K columns do not count
        launch(stream())
        kname = lib().sehip_last_kernel().decode()
        if len(members) == 2 and not any(t in kname for t in ("pair", "_stream_kernel", "conv_small2")):
            # the library ran this pair as its two launches (e.g. the deep decoder weight gradients): time and label each on its own
            for m in members:
                fn = "sehip_wgrad" if m.endswith(".wg") else "sehip_gemm"
                queue.append((m, [m], lambda st, fn=fn, m=m: call(fn, C.byref(ws.desc[m]), st)))
            continue
        if kname.startswith("conv_gemm_v3_kernel<"):
            # one class per (taps, row stride): the instantiations by rows per frame / tile rows / tile columns are the same code
            # (csrc/conv3.hip), picked per layer shape -- the successor of round 2's conv_gemm_v2_kernel<taps, ...> classes
            a = kname[len("conv_gemm_v3_kernel<"):].split(",")
            kname = f"conv_gemm_v3_kernel<{a[0].strip()}, {a[1].strip()}, *>"
        elif kname.startswith("conv_gemm_v3_pair_kernel<"):
            kname = "conv_gemm_v3_pair_kernel<*>"        # both output-row parities (3 + 2 taps) of a layer in one launch
        torch.cuda.synchronize()
        times = []
        for _ in range(reps):
            flush.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(stream())
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        ms = sorted(times)[len(times) // 2]
        r = per.setdefault(kname, {"kernel": kname, "launches": 0, "ms": 0.0, "gflop": 0.0, "layers": []})
        r["launches"] += 1; r["ms"] += ms; r["gflop"] += flops / 1e9; r["layers"].append(name)
    rows = sorted(per.values(), key=lambda r: -r["ms"])
    for r in rows:
        r["avg_us"] = r["ms"] / r["launches"] * 1e3
        r["tflops"] = r["gflop"] / r["ms"]
    return rows


def cbn_roofline(ws, model, reps=3):
    """HBM roofline entry of the ComplexBatchNorm class (DCCRN): every streaming pass of every layer timed as ONE launch between
    HIP events after the same 320 MB cache flush as the products.  Algorithmic bytes per pass = tensor bytes x (stats 1,
    apply 2, backward reduce 2, backward apply 3); the per-channel finalize launches move kilobytes and are left out of the
    byte count but not of the time."""
    from sehip._lib import call, ptr, stream
    flush = torch.empty(80 * 1024 * 1024, dtype=torch.float32, device=ws.device)
    params = model._flat
    L = ws.st.layout
    cfg = ws.st.cfg
    b = ws.bufs
    layers = [(f"encoder.{i}.", cfg.kernel_num[i + 1] // 2, b[f"y{i}"], b[f"z{i}"], b[f"dye{i}"], b[f"dye{i}"], 0) for i in range(6)]
    layers += [(f"decoder.{j}.", cfg.kernel_num[5 - j] // 2, b[f"yd{j}"], b[f"zd{j}"], b[f"dzd{j}"], b[f"dyd{j}"], 1) for j in range(5)]
    # bracket overhead of two events around one launch (these passes take 10-35 us, the bracket costs several): an
    # empty-sized fill timed the same way, minus the ~2 us such a kernel runs, is subtracted from every measurement
    one = torch.zeros(64, device=ws.device)
    empt = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        one.fill_(1.0)
        e1.record()
        torch.cuda.synchronize()
        empt.append(e0.elapsed_time(e1))
    bracket_ms = max(0.0, sorted(empt)[len(empt) // 2] - 0.002)
    tot_ms, tot_bytes, n = 0.0, 0.0, 0
    for pre, cr, y, z, dz, dy, tfirst in layers:
        rows = y.t.numel() // (2 * cr)
        tbytes = y.t.numel() * 2
        pp = lambda k: params.data_ptr() + 4 * L.param_off[pre + k][0]
        coef = ws.bn_coef[pre]
        passes = [
            (lambda: call("sehip_cbn_stats", y.ptr, rows, cr, ptr(ws.bn_acc), stream()), 1),
            (lambda: call("sehip_cbn_apply", y.ptr, ptr(coef), pp("2.weight"), rows, cr, z.ptr, stream()), 2),
            (lambda: call("sehip_cbn_bwd_reduce", dz.ptr, None, y.ptr, ptr(coef), pp("2.weight"), rows, cr, y.F, y.Tst, tfirst,
                          ptr(ws.bn_acc), stream()), 2),
            (lambda: call("sehip_cbn_bwd_apply", dz.ptr, None, y.ptr, ptr(coef), ptr(ws.bn_bcoef), pp("2.weight"), rows, cr, y.F,
                          y.Tst, tfirst, dy.ptr, stream()), 3),
        ]
        for fn, mult in passes:
            if mult == 1 and (pre in ws.st.fused_stats or pre in getattr(ws, "fused_small", ())):
                continue          # this layer's sums come out of the convolution's epilogue: no stats pass in the step
            times = []
            for _ in range(reps):
                flush.fill_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            tot_ms += max(sorted(times)[len(times) // 2] - bracket_ms, 1e-3)
            tot_bytes += mult * tbytes
            n += 1
    return {"bound": "hbm", "kernel": "cbn_apply / cbn_bwd_reduce / cbn_bwd_apply (all layers; cbn_stats where a layer still runs it)", "achieved": tot_bytes / tot_ms / 1e6,
            "peak": 8000.0, "unit": "GB/s", "frac": tot_bytes / tot_ms / 1e6 / 8000.0, "traffic": None, "launches_per_step": n,
            "ms_per_step": tot_ms, "algorithmic_bytes_per_step": tot_bytes, "event_bracket_us_subtracted": bracket_ms * 1e3}


def _weight_entries(ws, name):
    key = name[:-3] if name.endswith(".wg") else name
    st = getattr(ws, "st", None)
    if st is not None and hasattr(st, "prods"):       # Demucs: dense weights (channel counts are multiples of 16, K = taps x channels);
        d = ws.desc[name]                             # its packing tables are freed after the build
        return int(d.N) * int(d.K)
    specs = ws.pl.specs if hasattr(ws, "pl") else ws.st.specs
    return int((specs[key].widx >= 0).sum())


# ---- the other three BASELINE configurations: CPU baseline + parity on a stated SUB-batch of the same synthetic batch ----------
SUB_BATCH = {"dcunet": 4, "convtasnet": 4, "demucs": 2}      # clips of the CPU leg / parity block (full batches: 64 / 32 / 16)
SUB_STEPS = 3                                                # 1 warm-up + 2 timed oracle train steps
TOL_GENERIC = {"dcunet": {"out_rel": 3e-2, "loss_rel": 3e-2}, "convtasnet": {"out_rel": 3e-2, "dloss_db": 0.15},
               "demucs": {"out_rel": 3e-2, "dloss_db": 0.15}}


def workload_batch(workload, b, rank, device):
    """(noisy [B, C, n], clean [B, S, C, n]) exactly as main() builds them for the timed run."""
    n = {"dcunet": 32768, "convtasnet": 32000, "demucs": 96000}[workload]
    noisy, clean = make_batch(b, n, rank, device)
    if workload == "dcunet":      # unit-variance clips (randn(64, 1, 32768)), not the 0.1-scale of C1
        noisy, clean = 10.0 * noisy, 10.0 * clean
    if workload == "demucs":
        n2, c2 = make_batch(b, n, rank + 1000, device)
        noisy = torch.cat([noisy, 0.8 * clean[:, 0] + (n2 - c2[:, 0])], dim=1)
        clean = torch.cat([clean, 0.8 * clean], dim=2)
    if workload == "convtasnet":
        clean = torch.cat([clean, noisy.unsqueeze(1) - clean], dim=1)
    return noisy, clean


def oracle_problem(workload, p, noisy, clean):
    """forward(params, stats_out) -> estimate, loss function, model-domain target and the trainable keys of the fp32 CPU oracle."""
    import torch.nn.functional as F
    from oracle import dccrn_oracle as O
    if workload == "dcunet":
        from oracle import dcunet_oracle as D, stft_oracle as S
        x = torch.from_numpy(S.stft_custom(noisy.numpy(), 512, 128, 512))
        tgt = torch.from_numpy(S.stft_custom(clean[:, 0].numpy(), 512, 128, 512))
        fwd = lambda prm, st: D.dcunet_forward(prm, x, model_complexity=45, model_depth=10, training=True, stats_out=st)
        return fwd, F.mse_loss, tgt, [k for k in p if D.is_trainable(k)]
    if workload == "convtasnet":
        from oracle import convtasnet_oracle as CT
        fwd = lambda prm, st: CT.convtasnet_forward(prm, noisy, audio_channels=1)
        return fwd, O.loss_sisdr, clean, [k for k in p if CT.is_trainable(k)]
    from oracle import demucs_oracle as DM
    cfg = DM.DemucsConfig(sources=["clean"], audio_channels=2)
    fwd = lambda prm, st: DM.demucs_forward(prm, noisy, cfg)
    return fwd, O.loss_sisdr, clean, [k for k in p if DM.is_trainable(k)]


def cpu_worker_generic(workload, state_path, out_path, threads):
    """Child process: the workload's fp32 CPU oracle on the first SUB_BATCH clips of the bench batch, from the same initial weights:
    training-mode forward (step-0 estimate), SUB_STEPS train steps (loss, backward, clip 5, Adam 3e-4), forward again."""
    torch.set_num_threads(threads)
    b = SUB_BATCH[workload]
    p = torch.load(state_path)
    full = {"dcunet": 64, "convtasnet": 32, "demucs": 16}[workload]
    noisy, clean = workload_batch(workload, full, 0, "cpu")       # the bench batch (same generator sequence), then its first clips
    noisy, clean = noisy[:b].contiguous(), clean[:b].contiguous()
    fwd, loss_fn, tgt, names = oracle_problem(workload, p, noisy, clean)
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    opt = torch.optim.Adam([leaves[k] for k in names], lr=3e-4)
    with torch.no_grad():
        est0 = fwd(work, None)
    losses, times = [], []
    for _ in range(SUB_STEPS):
        t0 = time.time()
        stats = {}
        loss = loss_fn(fwd(work, stats), tgt)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([leaves[k] for k in names], 5.0)
        opt.step()
        work.update(stats)
        times.append(time.time() - t0)
        losses.append(float(loss.detach()))
    with torch.no_grad():
        estk = fwd(work, None)
    torch.save({"est0": est0, "estk": estk, "losses": losses}, out_path)
    dt = sum(times[1:]) / len(times[1:])
    clip_s = {"dcunet": 32768 / SR, "convtasnet": 4.0, "demucs": 2.0}[workload]
    print(json.dumps({"value": b * clip_s / dt, "unit": "audio-s/s", "cores": threads, "kind": "port",
                      "host_cores": len(os.sched_getaffinity(0)), "s_per_step": dt,
                      "sample": f"{len(times) - 1} timed train steps after 1 warm-up on the first {b} clips of the bench batch "
                                f"(fp32 oracle of this network, torch CPU, {threads} threads), {dt:.2f} s/step; a sub-batch because the "
                                f"full batch does not fit the bench's CPU time bound"}))


def hip_parity_generic(solver, model, noisy, clean, workload):
    """The HIP side on the same sub-batch, from the freshly initialised weights."""
    b = SUB_BATCH[workload]
    mix, src = solver._prepare_batch(noisy[:b], clean[:b])
    model.train()
    with torch.no_grad():
        est0 = model(mix).cpu()
    losses = []
    for _ in range(SUB_STEPS):
        loss, _ = solver.train_step(mix, src)
        losses.append(float(loss))
    with torch.no_grad():
        estk = model(mix).cpu()
    return {"est0": est0, "estk": estk, "losses": losses}


def parity_generic(hip, cpu_path, workload):
    cpu = torch.load(cpu_path)
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    tol = TOL_GENERIC[workload]
    out = {"against": f"fp32 CPU oracle of this network (oracle/, pinned to the reference by tests/golden), identical initial weights and "
                      f"the first {SUB_BATCH[workload]} clips of the bench batch", "batch": SUB_BATCH[workload], "train_steps": SUB_STEPS,
           "loss_hip": hip["losses"], "loss_cpu": cpu["losses"], "output_step0_rel": rel(hip["est0"], cpu["est0"]),
           f"output_after_{SUB_STEPS}_steps_rel": rel(hip["estk"], cpu["estk"]), "tolerance": tol}
    if "dloss_db" in tol:
        out["max_abs_dloss_db"] = max(abs(a - b) for a, b in zip(hip["losses"], cpu["losses"]))
        ok = out["max_abs_dloss_db"] < tol["dloss_db"]
    else:
        out["max_loss_rel"] = max(abs(a - b) / abs(b) for a, b in zip(hip["losses"], cpu["losses"]))
        ok = out["max_loss_rel"] < tol["loss_rel"]
    out["pass"] = bool(ok and out["output_step0_rel"] < tol["out_rel"] and out[f"output_after_{SUB_STEPS}_steps_rel"] < 2 * tol["out_rel"])
    return out


def activation_bytes(workload, ws, batch):
    """A = bytes of the layer-boundary activations of one forward pass in their stored types; the compulsory-traffic model of
    SURVEY section 8d (forward write + read 2 A, skip re-reads, backward read-saved + write-grad + read-grad 3 A => 5.5 A per step)
    gives the algorithmic HBM bytes of a train step."""
    if workload == "convtasnet":
        cfg, M, K = ws.st.cfg, ws.M, ws.K
        nb = len(ws.st.blocks)
        per_frame = cfg.N * 4 + cfg.N * 2 + cfg.B * 2 + nb * (4 * cfg.H + cfg.B) * 2 + cfg.C * cfg.N * 2
        return M * K * per_frame + 2 * M * cfg.C * ws.T * 4
    if workload == "dcunet":     # packed input spectrum, every pre-BatchNorm tensor and every BatchNorm + LeakyReLU output (bf16), the
        #                           fp32 input / output spectra [B, 1, 257, 257, 2]
        fwd = [k for k in ws.bufs if k == "x0" or k.startswith(("ye", "ze", "yd", "zd"))]
        return sum(ws.bufs[k].t.numel() * ws.bufs[k].t.element_size() for k in fwd) + 2 * ws.out.numel() * 4
    if workload == "demucs":     # forward buffers = every workspace buffer that is not the gradient twin ("d" + name) of another
        names = set(ws.bufs)
        fwd = [k for k in names if not (k.rsplit(".", 1)[-1].startswith("d") and
                                        (k.rsplit(".", 1)[0] + "." if "." in k else "") + k.rsplit(".", 1)[-1][1:] in names)]
        return sum(ws.bufs[k].t.numel() * ws.bufs[k].t.element_size() for k in fwd)
    return 0


PARITY_STEPS = 4           # 1 warm-up + 3 timed oracle steps; the HIP side runs the same 4
TOL_DLOSS_DB, TOL_WAVE_REL = 0.15, 3e-2


def snr_db(x, ref):
    return float(10.0 * torch.log10(ref.double().pow(2).sum() / ((x.double() - ref.double()).pow(2).sum() + 1e-30)))


def cpu_baseline_worker(state_path, out_path, batch, threads):
    """Child process: the oracle (fp32 PyTorch-CPU restatement of the reference step, pinned to the reference by
    tests/golden) on the SAME B=`batch` synthetic batch and the SAME initial weights as the GPU run, on `threads` host
    threads: forward (step-0 waveform), 1 warm-up + 3 timed train steps, forward again.  Prints one JSON object and
    leaves waveforms / losses in `out_path` for the parity block."""
    from oracle import dccrn_oracle as O
    torch.set_num_threads(threads)
    n = int(SR * CLIP_S)
    cfg = O.DCCRNConfig(length=n)
    p = torch.load(state_path)
    adam = O.AdamState({k: v for k, v in p.items() if O.is_trainable(k)}, lr=3e-4)
    noisy, clean = make_batch(batch, n, 0, "cpu")
    bases = O.stft_bases(cfg.win_len, cfg.fft_len)
    with torch.no_grad():
        est0 = O.dccrn_forward(p, noisy, cfg, training=True, bases=bases)
    losses, times = [], []
    for s in range(PARITY_STEPS):
        t0 = time.time()
        loss, _metric, _g = O.train_step(p, noisy, clean[:, 0], cfg, adam, clip_grad=5.0, bases=bases)
        times.append(time.time() - t0)
        losses.append(loss)
    with torch.no_grad():
        estk = O.dccrn_forward(p, noisy, cfg, training=True, bases=bases)
    torch.save({"est0": est0, "estk": estk, "losses": losses}, out_path)
    dt = sum(times[1:]) / len(times[1:])
    print(json.dumps({"value": batch * CLIP_S / dt, "unit": "audio-s/s", "cores": threads, "kind": "port",
                      "host_cores": len(os.sched_getaffinity(0)), "s_per_step": dt,
                      "sample": f"{len(times) - 1} timed train steps after 1 warm-up of the same B={batch} x 2-s batch "
                                f"(fp32 oracle, torch CPU, {threads} threads), {dt:.2f} s/step"}))


def cpu_baseline(state_path, out_path, batch, threads, timeout_s=900, workload="dccrn"):
    """Runs the worker as a child process (own thread pool, hard timeout) so a slow host cannot stall the bench."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", state_path, out_path,
                            "--batch", str(batch), "--cpu-threads", str(threads), "--workload", workload], capture_output=True,
                           text=True, timeout=timeout_s, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout / failure: report it, do not fake a number
        return {"value": None, "unit": "audio-s/s", "cores": threads, "kind": "port",
                "sample": f"cpu baseline did not finish: {type(e).__name__}"}


def hip_parity_run(solver, model, mixture, sources):
    """The HIP side of the parity block, from the freshly initialised weights: step-0 waveform, PARITY_STEPS train steps on
    the same batch (losses), waveform afterwards.  The CPU worker repeats exactly this with the oracle."""
    model.train()
    bsave, nsave = model._bflat.clone(), model._nbt.clone()
    with torch.no_grad():
        est0 = model(mixture).cpu()
    model._bflat.copy_(bsave); model._nbt.copy_(nsave)      # a training-mode forward advances the running statistics
    losses = []
    for _ in range(PARITY_STEPS):
        loss, _ = solver.train_step(mixture, sources)
        losses.append(float(loss))
    bsave, nsave = model._bflat.clone(), model._nbt.clone()
    with torch.no_grad():
        estk = model(mixture).cpu()
    model._bflat.copy_(bsave); model._nbt.copy_(nsave)
    return {"est0": est0, "estk": estk, "losses": losses}


def parity_block(hip, cpu_path, batch):
    cpu = torch.load(cpu_path)
    def wave(a, b):
        return {"max_abs": float((a - b).abs().max()), "rel": float((a - b).norm() / b.norm()), "snr_db": snr_db(a, b)}
    dl = [abs(a - b) for a, b in zip(hip["losses"], cpu["losses"])]
    w0, wk = wave(hip["est0"], cpu["est0"]), wave(hip["estk"], cpu["estk"])
    return {"against": "fp32 CPU oracle (oracle/dccrn_oracle.py, pinned to the reference by tests/golden), identical initial "
                       "weights and (noisy, clean) batch", "batch": batch, "train_steps": PARITY_STEPS,
            "loss_hip": [round(v, 5) for v in hip["losses"]], "loss_cpu": [round(v, 5) for v in cpu["losses"]],
            "max_abs_dloss_db": max(dl), "waveform_step0": w0, f"waveform_after_{PARITY_STEPS}_steps": wk,
            "tolerance": {"dloss_db": TOL_DLOSS_DB, "waveform_rel": TOL_WAVE_REL,
                          "why": "bf16 activation storage / bf16 MFMA operands with fp32 accumulation vs fp32"},
            "pass": bool(max(dl) < TOL_DLOSS_DB and w0["rel"] < TOL_WAVE_REL and wk["rel"] < 2 * TOL_WAVE_REL)}


def build_for_profile(batch=BATCH):
    """Model, solver and one staged batch of the headline workload (tools/host_profile.py)."""
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    n = int(SR * CLIP_S)
    cfg = bench_config(n)
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    opt = distrib.get_optimizer(cfg.optim, model)
    solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), device="gpu", writer=ScalarLog())
    noisy, clean = make_batch(batch, n, 0, solver.device)
    mixture, sources = solver._prepare_batch(noisy, clean)
    return solver, model, mixture, sources


def measure_traffic(workload, batch, timeout_s=420):
    """HBM bytes per kernel class of THIS build, measured in this run: two child processes of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only beside them, as MI355X_MICROARCH.md's HBM
    section prescribes) over 1 warm-up + 2 train steps.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the
    bytes of wide (16 B per lane) coalesced reads, which is what these kernels issue: the read side is doubled.  Returns
    {class name: {"launches_per_step", "hbm_bytes_per_launch"}, "_whole_step": {...}} or None (no rocprofv3 / a pass failed)."""
    import collections, csv, glob, re, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None
    child_steps, child_warmup = 2, 1
    steps = child_steps + child_warmup          # every step of the child is profiled (launches_per_step = dispatches / steps)
    tmp = tempfile.mkdtemp(prefix="sehip_pmc_", dir="/tmp")
    acc = {"fetch": collections.defaultdict(list), "write": collections.defaultdict(list)}
    try:
        for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            out = os.path.join(tmp, kind)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--workload", workload, "--batch", str(batch), "--steps", str(child_steps),
                   "--warmup", str(child_warmup),
                   "--no-cpu-baseline", "--no-roofline", "--no-parity", "--no-traffic"]
            env = {**os.environ, "TMPDIR": "/tmp"}
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                note(f"traffic pass {counter} failed (rc {r.returncode}): {r.stderr[-300:]}")
                return None
            for row in csv.DictReader(open(files[0])):
                m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)", row["Kernel_Name"])
                acc[kind][m.group(1) if m else row["Kernel_Name"]].append(float(row["Counter_Value"]))
    except Exception as e:  # timeout etc.: report, never fake
        note(f"traffic measurement failed: {type(e).__name__}: {e}")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # FETCH_SIZE is doubled for every class (MI355X_MICROARCH.md: gfx950 tallies the 128-byte requests of wide coalesced reads at 64
    # bytes).  That is calibrated for 16-byte-per-lane streaming reads and LDS-DMA -- what the convolution, BatchNorm, STFT and
    # optimizer kernels issue; for the few kernels that read narrower (finalize / reduce / scalar kernels, < 1 % of the step's bytes)
    # it is an UPPER bound.  The raw counters are kept beside the corrected figure.
    res, lib, raw_f, raw_w = {}, 0.0, 0.0, 0.0
    for k in sorted(set(acc["fetch"]) | set(acc["write"])):
        f, w = acc["fetch"].get(k, []), acc["write"].get(k, [])
        fk, wk = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
        res[k] = {"launches_per_step": round(max(len(f), len(w)) / steps, 2), "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                  "raw_fetch_bytes_per_launch": fk * 1024.0, "raw_write_bytes_per_launch": wk * 1024.0}
        if not k.startswith(("at::", "__amd")):      # the library's own kernels (torch fills / copies of the allocations excluded)
            lib += (2.0 * sum(f) + sum(w)) * 1024.0
            raw_f += sum(f) * 1024.0
            raw_w += sum(w) * 1024.0
    res["_whole_step"] = {"library_kernels_hbm_bytes_per_step": lib / steps, "steps_profiled": steps,
                          "raw_fetch_bytes_per_step": raw_f / steps, "raw_write_bytes_per_step": raw_w / steps}
    return res


def class_traffic(tj, kernel):
    """bytes per launch of a kernel (class) from a traffic table: exact name, or the launch-weighted mean over the members of a class
    written as 'name<a, b, *>'"""
    if tj is None:
        return None
    v = tj.get(kernel, {}).get("hbm_bytes_per_launch") if isinstance(tj.get(kernel), dict) else None
    if v is None and kernel.endswith("*>"):
        pre = kernel[:-2]
        mem = [x for k, x in tj.items() if k.startswith(pre) and isinstance(x, dict) and "hbm_bytes_per_launch" in x]
        nl = sum(x["launches_per_step"] for x in mem)
        if nl > 0:
            v = sum(x["hbm_bytes_per_launch"] * x["launches_per_step"] for x in mem) / nl
    return v


def live_communicator_report(rank, world, dev, ms_rank):
    """What the LIVE communicator says about the N > 1 run, so that the line proves N ranks on N devices instead of repeating what the
    launcher was asked for: the backend, its own rank count (ncclCommCount through sehip_comm_info on the
    sehip-rccl path; torch.distributed's process group otherwise, checked by an all-reduce of ones ON THE DEVICE), every rank's
    device (index, name, PCI bus id) gathered over the communicator, and the per-rank step time (min / max beside the MAX the
    throughput uses).  Collective: every rank calls it; rank 0 gets the dict."""
    import torch.distributed as dist
    from sehip import distrib
    ones = torch.ones(1, device=dev, dtype=torch.float32)
    direct = distrib.direct_comm()
    info = {}
    if direct is not None:
        direct.all_reduce_(ones).wait()
        n, r, d = direct.info()
        info.update({"backend": "sehip-rccl (sehip_comm_init / sehip_allreduce_f32; control plane " + dist.get_backend() + ")",
                     "nranks": n, "comm_rank": r, "comm_device": d})
    else:
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        info.update({"backend": dist.get_backend() + " (torch.distributed; nccl = RCCL on ROCm)", "nranks": dist.get_world_size()})
    torch.cuda.synchronize()
    info["allreduce_of_ones"] = float(ones[0])              # == nranks iff every rank took part in a device collective
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": int(dev.index if dev.index is not None else torch.cuda.current_device()), "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")), "ms_per_step": round(ms_rank, 4)}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    info["ranks"] = allr
    ms = [a["ms_per_step"] for a in allr]
    info["ms_per_step_min"], info["ms_per_step_max"] = min(ms), max(ms)
    info["distinct_devices"] = len({(a["pci_bus_id"], a["uuid"], a["device"]) for a in allr})
    return info


class NeedsDevices(RuntimeError):
    """`--gpus N` asked for more devices than this node shows."""


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks ourselves (one process per GPU, the data-parallel
    replacement of the reference's single-process nn.DataParallel, src/solver.py:144-145) as CHILD processes of
    `python -m torch.distributed.run`, relay rank 0's JSON line and return the launcher's exit code.  This parent never touches the
    GPU (torch.cuda.device_count() does not initialise it on this image), and nothing is exec'ed from a process that did."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and "SEHIP_LOCAL_DEVICE" not in os.environ:     # (test hook: several ranks on one device over gloo)
        raise NeedsDevices(f"bench.py --gpus {n}: needs {n} devices, this node shows {have}")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if r.returncode != 0 or not lines:
        print(f"bench.py --gpus {n}: the ranks failed (launcher exit code {r.returncode})", file=sys.stderr)
        return r.returncode or 1
    print(lines[-1], flush=True)
    return 0


def note(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default: 32 for dccrn, 64 for dcunet)")
    ap.add_argument("--workload", choices=("dccrn", "dcunet", "convtasnet", "demucs"), default="dccrn",
                    help="dccrn = BASELINE configs[1], the headline metric; dcunet = configs[2] (DCUnet-10, STFT-domain mse, B=64); "
                         "convtasnet = configs[4] (2-speaker separation, 8 kHz 4-s clips, B=32 per GPU); demucs = configs[3] (48 kHz stereo "
                         "2-s clips, B=16 per GPU)")
    ap.add_argument("--kernel-num", default="", help="DCCRN only: six comma-separated channel counts instead of the headline's "
                    "16,32,64,128,256,256 (e.g. the reference YAML's commented 'paper' widths 32,64,128,128,256,256, "
                    "src/conf/config.yaml:86-88).  A side measurement: implies --no-roofline --no-cpu-baseline --no-traffic, and the "
                    "line's config.workload names the widths")
    ap.add_argument("--rnn-units", type=int, default=128, help="DCCRN only: 256 = the complex LSTM of the DCCRN paper (hidden 128 per "
                    "nn.LSTM; one launch per layer and direction).  A side measurement like --kernel-num")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as two captured hipGraphs instead of launching "
                    "every kernel from Python (measured slower: graph replay serialises the side-stream weight gradients)")
    ap.add_argument("--h2d", action="store_true", help="stage every batch from pinned host memory inside the timed region "
                    "(the PCIe-inclusive rate quoted in DESIGN.md; never the headline value)")
    ap.add_argument("--h2d-overlap", action="store_true", help="as --h2d, but the NEXT batch is copied on a stream of its own while the "
                    "step runs (two device buffers): what sehip.solver.DevicePrefetcher does for the Solver's epoch loop")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes, "
                    "~1 min); the line then carries the tracked profiles/ figure with the commit it was taken at")
    ap.add_argument("--no-parity", action="store_true", help="skip the HIP-vs-oracle parity block (it needs the CPU baseline leg)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the CPU baseline (0 = min(cores of the affinity mask, 32): on the 256-core host of the GPU "
                         "box torch's CPU ops with 256 threads did not finish ONE B=32 step in 15 minutes, 32 threads take ~5 s)")
    ap.add_argument("--cpu-baseline-worker", nargs=2, metavar=("STATE", "OUT"))
    args = ap.parse_args()
    if args.kernel_num:
        widths = [int(v) for v in args.kernel_num.split(",")]
        if args.workload != "dccrn" or len(widths) != 6:
            sys.exit("--kernel-num: six channel counts, DCCRN only")
        KERNEL_NUM[:] = widths
        if widths != HEADLINE_KERNEL_NUM:
            args.no_roofline = args.no_cpu_baseline = args.no_traffic = True
    if args.rnn_units != 128:
        if args.workload != "dccrn":
            sys.exit("--rnn-units: DCCRN only")
        RNN_UNITS[0] = args.rnn_units
        args.no_roofline = args.no_cpu_baseline = args.no_traffic = True
    dcu = args.workload == "dcunet"
    ctn = args.workload == "convtasnet"
    dmx = args.workload == "demucs"
    if not args.batch:
        args.batch = 64 if dcu else 16 if dmx else BATCH
    if args.cpu_baseline_worker:
        threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 32)
        if args.workload == "dccrn":
            cpu_baseline_worker(args.cpu_baseline_worker[0], args.cpu_baseline_worker[1], args.batch, threads)
        else:
            cpu_worker_generic(args.workload, args.cpu_baseline_worker[0], args.cpu_baseline_worker[1], threads)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # plain `python bench.py --gpus N`: this process only starts the ranks

    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    rank, world, local = distrib.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    n = 32768 if dcu else int(SR * CLIP_S)        # C2: 257 frames (= 1 mod 32, the depth-10 constraint) at hop 128
    clip_s = n / SR
    cfg = dcunet_config() if dcu else bench_config(n)
    if ctn:
        n, clip_s, cfg = 32000, 4.0, convtasnet_config()       # 4 s at 8 kHz
    if dmx:
        n, clip_s, cfg = 96000, 2.0, demucs_config()           # 2 s at 48 kHz
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    opt = distrib.get_optimizer(cfg.optim, model)
    solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), device="gpu", writer=ScalarLog())
    dev = solver.device
    if dcu or ctn or dmx:
        # dcunet: unit-variance clips; demucs: stereo, the second channel an attenuated, differently-noised copy, one source [B, 1, 2, N];
        # convtasnet: two sources per clip whose sum is the mixture (sources [B, S, 1, N] stay 4-D for this model, src/solver.py:443-452)
        noisy, clean = workload_batch(args.workload, args.batch, rank, dev)
    else:
        noisy, clean = make_batch(args.batch, n, rank, dev)
    mixture, sources = solver._prepare_batch(noisy, clean)
    if dcu:
        # the reference transforms mixture and sources inside every step (src/solver.py:454-458): timed with the step
        base_step = (lambda fn: (lambda mix_, src_: fn(*solver._prepare_batch(noisy, clean))))
    else:
        base_step = lambda fn: fn

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    do_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    hip_par = None
    if do_cpu:
        import tempfile
        tmpd = tempfile.mkdtemp(prefix="sehip_bench_")
        state_path, cpu_out = os.path.join(tmpd, "state0.pt"), os.path.join(tmpd, "cpu.pt")
        torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()
                    if not k.startswith(("stft.", "istft.", "encoders.", "decoders."))}, state_path)
        if not args.no_parity:
            note(f"parity leg on the GPU: step-0 output + train steps from the initial weights")
            hip_par = (hip_parity_generic(solver, model, noisy, clean, args.workload) if (dcu or ctn or dmx)
                       else hip_parity_run(solver, model, mixture, sources))
    note(f"model built, batch staged on {dev}; warm-up {args.warmup} steps")
    args.eager = not args.graph
    step_fn = base_step(solver.train_step if args.eager else solver.train_step_graphed)
    # the dependent chain of the step runs on a high-priority stream; the weight gradients (side stream, default
    # priority) then only take the CUs the chain leaves idle
    lo, hi = torch.cuda.Stream.priority_range()
    main_stream = torch.cuda.Stream(device=dev, priority=hi) if os.environ.get("SEHIP_BENCH_PRIO", "1") == "1" else torch.cuda.current_stream()
    main_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main_stream)
    if args.h2d or args.h2d_overlap:
        h_mix, h_src = mixture.cpu().pin_memory(), sources.cpu().pin_memory()
    if args.h2d_overlap:
        bufs = [(mixture, sources), (mixture.clone(), sources.clone())]
        cstream = torch.cuda.Stream(device=dev)
        done = [None, None]                 # event after the last step that read buffer i
        state = {"k": 0}

        def copy_into(i):
            with torch.cuda.stream(cstream):
                if done[i] is not None:
                    cstream.wait_event(done[i])
                bufs[i][0].copy_(h_mix, non_blocking=True)
                bufs[i][1].copy_(h_src, non_blocking=True)
        copy_into(0)

    def one_step():
        if args.h2d_overlap:
            i = state["k"] & 1
            state["k"] += 1
            torch.cuda.current_stream().wait_stream(cstream)      # batch k has landed
            copy_into(i ^ 1)                                      # batch k + 1 travels while step k computes
            out = step_fn(*bufs[i])
            done[i] = torch.cuda.Event()
            done[i].record()
            return out
        if args.h2d:
            mixture.copy_(h_mix, non_blocking=True)
            sources.copy_(h_src, non_blocking=True)
        return step_fn(mixture, sources)

    for _ in range(args.warmup):
        one_step()
    sync()
    note(f"timing {args.steps} steps ({'eager launches' if args.eager else 'hipGraph replay'})")
    t0 = time.time()
    for _ in range(args.steps):
        loss, metric = one_step()
    t_enq = time.time() - t0            # host time to enqueue the steps; close to dt = the launches, not the GPU, set the pace
    sync()
    dt = time.time() - t0
    note(f"host enqueue {t_enq / args.steps * 1e3:.2f} ms/step of {dt / args.steps * 1e3:.2f}")
    rccl_info = None
    if world > 1:
        dt_rank = dt
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        rccl_info = live_communicator_report(rank, world, dev, dt_rank / args.steps * 1e3)
    ms = dt / args.steps * 1e3
    note(f"{ms:.2f} ms/step")
    if dmx:
        model.workspace(args.batch, n).check_lstm_handoffs()    # the timed steps are valid only if no hand-off spin gave up
    value = world * args.batch * clip_s * args.steps / dt

    out = {
        "metric": ("audio-sec/sec training, DCUnet-10 16kHz 2.048s bs64" if dcu else "audio-sec/sec training, ConvTasNet 8kHz 4s bs32"
                   if ctn else "audio-sec/sec training, Demucs 48kHz stereo 2s bs16" if dmx else "audio-sec/sec training, DCCRN 16kHz 2s bs32"), "value": value, "unit": "audio-s/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": ("DCUnet-10 (complex, model_complexity 45 -> 31/62 channels, mask E) train step on [64,1,257,257,2] "
                                "spectra: stft_custom of mixture and sources, mse in the STFT domain, Adam 3e-4, clip 5" if dcu else
                                "ConvTasNet (N128 L40 B128 H256 P3 X7 R2, gLN, relu mask) 2-speaker separation train step, 8 kHz 4-s clips, "
                                "SI-SNR, Adam 3e-4, clip 5" if ctn else
                                "Demucs (channels 64, depth 6, DConv with BLSTM + LocalState from layer 4, x2 resampling; 133.7 M parameters) "
                                "denoising train step, 48 kHz stereo 2-s clips, SI-SNR, Adam 3e-4, clip 5" if dmx else
                                f"DCCRN (kernel_num {'-'.join(str(v) for v in KERNEL_NUM)}, complex LSTM {RNN_UNITS[0]}, mask E) train step, 16 kHz 2-s "
                                "clips, SI-SNR, Adam 3e-4, clip 5"), "per_gpu_batch": args.batch,
                   "global_batch": args.batch * world, "samples_per_clip": n, "parallelism": f"dp{world}",
                   "launch": "eager" if args.eager else "hipGraph", "inputs": "pinned host -> HBM every step, overlapped with the previous step" if args.h2d_overlap else
                   "pinned host -> HBM every step" if args.h2d else "resident in HBM"},
        "final_loss": float(loss),
    }
    if rccl_info is not None:
        out["rccl"] = rccl_info
    if rank == 0 and not args.no_roofline:
        ws = model.workspace(args.batch, 257, 257) if dcu else model.workspace(args.batch, n)
        note("per-kernel roofline pass")
        rows = gemm_roofline(ws)
        top = rows[0]
        total_ms = sum(r["ms"] for r in rows)
        total_gf = sum(r["gflop"] for r in rows)
        # roofline.traffic: measured in THIS run when rocprofv3 is there (a tracked file would silently go stale);
        # otherwise the newest tracked table of this workload, with the commit it was taken at in the line
        traffic, traffic_src, tj = None, None, None
        if not args.no_traffic and world == 1:
            note("PMC traffic passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, child processes)")
            tj = measure_traffic(args.workload, args.batch)
            if tj is not None:
                traffic_src = {"how": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over 1 + 2 train steps, "
                                      "class mean per launch, FETCH_SIZE doubled (gfx950 wide-read correction)",
                               "whole_step_bytes": tj["_whole_step"]["library_kernels_hbm_bytes_per_step"],
                               "raw_counters": {"fetch_bytes_per_step": tj["_whole_step"]["raw_fetch_bytes_per_step"],
                                                "write_bytes_per_step": tj["_whole_step"]["raw_write_bytes_per_step"],
                                                "note": "FETCH_SIZE x 2 + WRITE_SIZE = whole_step_bytes; the factor is calibrated for 16-byte-per-lane "
                                                        "reads, an upper bound for narrower ones"}}
        if tj is None:
            suffix = "" if args.workload == "dccrn" else f"_{args.workload}"
            for rnd in ("r5", "r4", "r3", "r2", "r1"):
                tpath = os.path.join(ROOT, "profiles", f"{rnd}_traffic{suffix}.json")
                if os.path.exists(tpath):
                    tj = json.load(open(tpath))
                    traffic_src = {"how": "tracked file, NOT measured in this run", "file": os.path.relpath(tpath, ROOT),
                                   "commit": tj.get("_whole_step", {}).get("commit")}
                    break
        traffic = class_traffic(tj, top["kernel"])
        out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["tflops"], "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": top["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                           "launches_per_step": top["launches"], "avg_launch_us": top["avg_us"],
                           "gflop_per_launch": top["gflop"] / top["launches"],
                           "all_product_kernels": {"ms_per_step": total_ms, "tflops": total_gf / total_ms,
                                                   "frac": total_gf / total_ms / PEAK_BF16_TFLOPS}}
        if ctn or dmx or dcu:
            # second entry, HBM: the whole step against the compulsory-traffic model (these two networks are bound by their
            # activation streams: normalisation / activation / depthwise kernels, not by the products)
            A = activation_bytes(args.workload, ws, args.batch)
            out["roofline_hbm"] = {"bound": "hbm", "scope": "whole train step", "achieved": 5.5 * A / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                   "unit": "GB/s", "frac": 5.5 * A / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                   "traffic": (tj or {}).get("_whole_step", {}).get("library_kernels_hbm_bytes_per_step"),
                                   "algorithmic_bytes_per_step": 5.5 * A,
                                   "model": "5.5 x the bytes of the layer-boundary activations of one forward pass in their stored "
                                            "types (SURVEY section 8d: forward write + read, backward read-saved + write-grad + read-grad)"}
        if not dcu and not ctn and not dmx:
            out["roofline_hbm"] = cbn_roofline(ws, model)   # second entry: the largest HBM-bound class of the step
            if tj is not None:   # PMC bytes of the same passes over the real step (finalize launches excluded)
                tb = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for k, v in tj.items()
                         if k.startswith("cbn_") and "finalize" not in k and isinstance(v, dict) and "hbm_bytes_per_launch" in v)
                out["roofline_hbm"]["traffic"] = tb or None
        # algorithmic FLOPs per clip-step (SURVEY section 8d / BASELINE.md): DCCRN 45.96 GF, DCUnet-10 111.9 GF (fwd + bwd)
        out["step_tflops"] = ((111.9e9 if dcu else 3 * 3.2e9 if ctn else 45.96e9) * args.batch if not dmx else total_gf * 1e9) / (ms * 1e-3) / 1e12
        out["kernel_classes"] = [{"kernel": r["kernel"], "launches": r["launches"], "avg_us": round(r["avg_us"], 1),
                                  "tflops": round(r["tflops"], 1)} for r in rows[:10]]
    if do_cpu:
        threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 32)
        note(f"cpu baseline + parity reference (oracle, B={args.batch}, {threads} threads)")
        out["cpu_baseline"] = cpu_baseline(state_path, cpu_out, args.batch, threads, workload=args.workload)
        if hip_par is not None and os.path.exists(cpu_out):
            out["parity"] = (parity_generic(hip_par, cpu_out, args.workload) if (dcu or ctn or dmx)
                             else parity_block(hip_par, cpu_out, args.batch))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
