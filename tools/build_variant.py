#!/usr/bin/env python3
"""A second libsehip for same-box A/B runs: ONE source recompiled with extra -D flags, linked with the product build's other objects.

    python tools/build_variant.py <tag> <source.hip>[,<source2.hip>...] [-DNAME[=V] ...]   ->  tools/_var_<tag>.so

The result is git-ignored (tools/_*), travels to the GPU box with the snapshot, and is selected per process with SEHIP_LIB or
tools/gemm_variants.py's `lib:tools/_var_<tag>.so`.  The product library is built first if it is stale."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-enhancement-pytorch_amd", "sehip")
sys.path.insert(0, PKG)
import build as B


def main():
    tag, srcs = sys.argv[1], sys.argv[2].split(",")
    defs = sys.argv[3:]
    # `name.hip=/path/to/other.hip`: compile that file in place of csrc/name.hip (e.g. `git show HEAD:.../conv3.hip > /tmp/x.hip`)
    alt = {q.split("=")[0]: q.split("=")[1] for q in srcs if "=" in q}
    srcs = [q.split("=")[0] for q in srcs]
    B.build(verbose=False)
    objs = []
    for name in B.sources():
        obj = os.path.join(B.OBJ, name + ".o")
        if name in srcs:
            obj = os.path.join(B.OBJ, f"_var_{tag}_{name}.o")
            cmd = ["hipcc", "-x", "hip"] + B.FLAGS + defs + ["-I", B.CSRC, "-c", alt.get(name, os.path.join(B.CSRC, name)), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
            if r.stderr.strip():
                sys.stderr.write(r.stderr)
        objs.append(obj)
    missing = [s for s in srcs if s not in B.sources()]
    if missing:
        sys.exit(f"no such source(s) in csrc/: {missing}")
    out = os.path.join(ROOT, "tools", f"_var_{tag}.so")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(f"link failed:\n{r.stdout}\n{r.stderr}")
    print(out)


if __name__ == "__main__":
    main()
