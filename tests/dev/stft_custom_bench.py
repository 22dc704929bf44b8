#!/usr/bin/env python3
"""stft_custom / istft_custom at the DCUNet configuration (BASELINE configs[2]: 64 x 32768 samples, 512/128/512):
HIP-event time per call, achieved HBM rate against the algorithmic bytes, and the numpy oracle on the host beside it."""
import json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sehip.evaluate import stft_custom, istft_custom
from oracle import stft_oracle as S

cfg = types.SimpleNamespace(n_fft=512, hop_length=128, win_length=512, center=True)
B, N = 64, 32768
x = torch.randn(B, 1, N, device="cuda")


def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

s = stft_custom(x, cfg)
us_f = timed(lambda: stft_custom(x, cfg))
us_i = timed(lambda: istft_custom(s, N, cfg))
bytes_f = x.numel() * 4 + s.numel() * 4
bytes_i = s.numel() * 4 + x.numel() * 4            # algorithmic: spectrum in, waveform out (the frame scratch is extra traffic)
xs = x[:4].cpu().numpy()
t0 = time.time(); so = S.stft_custom(xs, 512, 128, 512); t1 = time.time(); S.istft_custom(so, N, 512, 128, 512); t2 = time.time()
print(json.dumps({
    "workload": f"stft_custom / istft_custom, {B} x {N} samples, 512/128/512 -> {tuple(s.shape)}",
    "stft_us": us_f, "stft_GBps": bytes_f / us_f / 1e3, "stft_frac_of_8TBps": bytes_f / us_f / 1e3 / 8000,
    "istft_us": us_i, "istft_GBps": bytes_i / us_i / 1e3, "istft_frac_of_8TBps": bytes_i / us_i / 1e3 / 8000,
    "cpu_oracle": {"kind": "port", "cores": 1, "sample": "4 clips", "stft_us_per_clip": (t1 - t0) / 4 * 1e6, "istft_us_per_clip": (t2 - t1) / 4 * 1e6},
    "gpu_us_per_clip": {"stft": us_f / B, "istft": us_i / B}}))
