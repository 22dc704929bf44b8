#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --stats csv: name (cut), calls, total us, average us."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
for r in rows[:n]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']) / div:7.1f} {float(r['TotalDurationNs']) / 1e3 / div:9.1f} us {float(r['AverageNs']) / 1e3:8.1f} us")
