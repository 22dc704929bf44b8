#!/usr/bin/env python3
"""Timeline of one train step from a rocprofv3 kernel trace: per queue busy time, and the gaps on the chain's queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step starts at every stft_fwd_kernel (or the kernel named in SEHIP_TRACE_START, e.g. dmx_prep_kernel for Demucs)
import os
first = os.environ.get("SEHIP_TRACE_START", "stft_fwd_kernel")
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(first)]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
a, b = starts[which], starts[which + 1]
step = rows[a:b]
t0 = step[0]["s"]
print("step wall us", (rows[b]["s"] - t0) / 1e3, "kernels", len(step))
byq = collections.defaultdict(list)
for r in step: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs) / 1e3
    print("queue", q, "kernels", len(rs), "busy us", round(busy, 1))
mainq = max(byq, key=lambda q: len(byq[q]))
rs = byq[mainq]
prev_e = rs[0]["s"]
gaps = []
verbose = len(sys.argv) > 3
for r in rs:
    gap = (r["s"] - prev_e) / 1e3
    gaps.append((gap, r["Kernel_Name"][:50]))
    if verbose:
        print(f"{(r['s']-t0)/1e3:9.1f} gap {gap:7.1f} dur {(r['e']-r['s'])/1e3:8.1f}  {r['Kernel_Name'][:70]}")
    prev_e = max(prev_e, r["e"])
print("main queue", mainq, "sum of gaps us", round(sum(g for g, _ in gaps if g > 0), 1))
big = sorted(gaps, reverse=True)[:25]
for g, n in big: print(f"  gap {g:7.1f} us before {n}")
per = collections.defaultdict(lambda: [0, 0.0])
for r in rs:
    k = r["Kernel_Name"].split("(")[0][:48]
    per[k][0] += 1; per[k][1] += (r["e"] - r["s"]) / 1e3
print("main queue, time by kernel (us):")
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:22]: print(f"  {t:9.1f}  x{n:4d}  {k}")

# SEHIP_TRACE_TIMELINE=1: every kernel of the step on every queue in start order (queue tag, start, duration) -- who overlaps whom
if os.environ.get("SEHIP_TRACE_TIMELINE"):
    qn = {q: i for i, q in enumerate(sorted(byq, key=lambda q: -len(byq[q])))}
    print("timeline (us from the step's first kernel): queue start dur name")
    for r in step:
        print(f"  q{qn[r['Queue_Id']]} {(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f}  {r['Kernel_Name'].replace('void ', '')[:64]}")
