#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kres.py conv3.hip [filter-substring] [-DNAME ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech-enhancement-pytorch_amd", "sehip", "csrc")
src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-D")), "")
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
cmd = ["hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-fno-finite-math-only",
       "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", "/dev/null"] + defs
r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
cur, rows = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, q in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and flt not in dem: continue
    print(f"{dem[:70]:70s} vgpr {q.get('VGPRs')} agpr {q.get('AGPRs')} sgpr {q.get('TotalSGPRs')} scratch {q.get('ScratchSize [bytes/lane]')} "
          f"spill s{q.get('SGPRs Spill')} v{q.get('VGPRs Spill')} occ {q.get('Occupancy [waves/SIMD]')}")
