#!/usr/bin/env python3
"""Prints a per-kernel table (ms per step) from a rocprofv3 --kernel-trace --stats run of bench.py."""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 13
f = glob.glob(f"{d}/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r['Name'][:64]:64s} calls/step {int(r['Calls'])/steps:6.1f} ms/step {float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
print("GPU-busy ms per step:", round(tot / 1e6 / steps, 3))
