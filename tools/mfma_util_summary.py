#!/usr/bin/env python3
"""<dir>/pmc_mfma{A,B}/ (rocprofv3 --pmc passes over `bench.py --steps 2 --warmup 1`, SEHIP_NO_SIDE_STREAM=1 so that every kernel is
alone on the GPU) -> JSON on stdout: per kernel class the MFMA instruction count, MFMA busy cycles, the SQ busy cycles and the
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8
XCDs, MI355X_MICROARCH.md 'DVFS give-back').  Passes: A = SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_LDS, B = GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY.
    python tools/mfma_util_summary.py gpurun_out/r3 <commit>"""
import collections, csv, glob, json, re, sys

root = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/pmc_mfma*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"]
        if k.startswith(("at::", "__amd")):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_meta": {"commit": commit, "command": "rocprofv3 --pmc <pass> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                                               "--no-roofline (SEHIP_NO_SIDE_STREAM=1)",
                 "mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
                 "note": "class = kernel symbol incl. template arguments; means over all dispatches of the class"}}
for k, c in acc.items():
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    row = {"dispatches": max(len(v) for v in c.values())}
    row.update({n: mean[n] for n in sorted(mean)})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean.get("GRBM_GUI_ACTIVE", 0) > 0:
        row["mfma_util"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * mean["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_INSTS_MFMA" in mean and mean["SQ_INSTS_MFMA"] > 0 and "SQ_INSTS_VALU" in mean:
        row["valu_per_mfma"] = mean["SQ_INSTS_VALU"] / mean["SQ_INSTS_MFMA"]
    out[k] = row
order = sorted((k for k in out if k != "_meta"), key=lambda k: -out[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * out[k]["dispatches"])
print(json.dumps({"_meta": out["_meta"], **{k: out[k] for k in order}}, indent=1))
