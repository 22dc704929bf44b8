#!/usr/bin/env python3
"""Instruction mix of one kernel of a csrc file, split at its first / last MFMA: prologue, main loop, epilogue.
    python tools/asm_mix.py conv3.hip '_Z19conv_gemm_v3_kernelILi5ELi2ELi8ELi8ELi4ELi2ELi0E' [-DNAME ...]"""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech-enhancement-pytorch_amd", "sehip", "csrc")
src, key = sys.argv[1], sys.argv[2]
defs = [a for a in sys.argv[3:] if a.startswith("-D")]
out = f"/tmp/{src}.s"
subprocess.run(["hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-fno-finite-math-only",
                "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)] + defs, capture_output=True, cwd="/tmp")
s = open(out).read().split("\n")
i = next(k for k, l in enumerate(s) if l.startswith(key) and l.rstrip().split(":")[0].startswith(key) and ":" in l)
j = i
while ".end_amdhsa_kernel" not in s[j] and "s_endpgm" not in s[j] or j == i:
    j += 1
    if j >= len(s) - 1: break
# up to the last s_endpgm of the function
k = j
while k < len(s) - 1 and not s[k].startswith(".Lfunc_end"): k += 1
body = s[i:k]
isn = lambda l: l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")
mf = [n for n, l in enumerate(body) if "v_mfma" in l]
print("kernel", s[i].split(":")[0], "instructions", sum(map(isn, body)), "mfma", len(mf))
for tag, part in (("prologue", body[:mf[0]]), ("loop", body[mf[0]:mf[-1] + 1]), ("epilogue", body[mf[-1] + 1:])):
    c = collections.Counter(l.split()[0] for l in part if isn(l))
    print(f"{tag:9s} {sum(c.values()):6d}  " + " ".join(f"{a}:{b}" for a, b in c.most_common(22)))
