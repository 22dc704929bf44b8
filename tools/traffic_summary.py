#!/usr/bin/env python3
"""<dir>/pmc_{fetch,write}/... (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --steps 2 --warmup 1`) -> JSON on stdout:
HBM bytes per launch of every kernel CLASS (mean over all its dispatches of the real step, not one hand-picked layer) and the
whole-step total.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B per lane)
coalesced reads (MI355X_MICROARCH.md, HBM section), which is what these kernels issue, so the read side is doubled.

    python tools/traffic_summary.py gpurun_out/r2 3      # 3 = train steps in the profiled run (warm-up + timed)
"""
import collections, csv, glob, json, re, sys

root, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
acc = {"fetch": collections.defaultdict(list), "write": collections.defaultdict(list)}
for kind in ("fetch", "write"):
    files = glob.glob(f"{root}/pmc_{kind}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)", name)
        acc[kind][m.group(1) if m else name].append(float(r["Counter_Value"]))
res, tot_f, tot_w, lib_f, lib_w = {}, 0.0, 0.0, 0.0, 0.0
for k in sorted(set(acc["fetch"]) | set(acc["write"])):
    f, w = acc["fetch"].get(k, []), acc["write"].get(k, [])
    fk = sum(f) / len(f) if f else 0.0
    wk = sum(w) / len(w) if w else 0.0
    tot_f += sum(f); tot_w += sum(w)
    if not k.startswith(("at::", "__amd")):      # the library's own kernels (torch fill / copy kernels of the allocations excluded)
        lib_f += sum(f); lib_w += sum(w)
    res[k] = {"launches_per_step": round(max(len(f), len(w)) / steps, 2), "fetch_size_kib": fk, "write_size_kib": wk,
              "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
out = {"_whole_step": {"steps_profiled": steps, "fetch_size_kib_per_step": tot_f / steps, "write_size_kib_per_step": tot_w / steps,
                       "hbm_bytes_per_step": (2.0 * tot_f + tot_w) * 1024.0 / steps,
                       "library_kernels_hbm_bytes_per_step": (2.0 * lib_f + lib_w) * 1024.0 / steps,
                       "algorithmic_bytes_per_step": 80.7e6 * 32,
                       "note": "all dispatches of the profiled run / steps (model construction and the first-step allocations "
                               "included: an upper bound); FETCH_SIZE doubled (gfx950 wide-read correction)"}}
out.update(res)
print(json.dumps(out, indent=1))
