#!/usr/bin/env python3
"""rocprofv3 --pmc ... --kernel-trace --output-format csv  ->  per-kernel-class means of every collected counter.
    python tools/pmc_summary.py <dir> [substring filter]"""
import collections, csv, glob, re, sys
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"]
        if flt in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(44), "n".rjust(5), " ".join(c[-18:].rjust(18) for c in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", acc[k].get(names[0], [0])))):
    n = max(len(v) for v in acc[k].values())
    print(k[:44].ljust(44), str(n).rjust(5), " ".join((f"{sum(acc[k][c]) / len(acc[k][c]):.4g}" if c in acc[k] else "-").rjust(18) for c in names))
