"""Builds libsehip.so (hipcc, gfx950 only) in-tree next to this file.

    python speech-enhancement-pytorch_amd/sehip/build.py [--force]

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsehip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast", "-fno-finite-math-only"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(src, obj):
    if not os.path.exists(obj):
        return True
    m = os.path.getmtime(obj)
    # (the public header too: sehip_gemm_desc is passed to kernels by value, a stale object would read another layout)
    deps = [src] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "sehip.h"))
    return any(os.path.getmtime(d) > m for d in deps)


def _compile(name, force):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name + ".o")
    if force or _stale(src, obj):
        cmd = ["hipcc", "-x", "hip"] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj, True
    return obj, False


def build(force=False, verbose=True, tools=False):
    """tools=True (--tools): also compiles the timing-ablation instantiations (SEHIP_C3_ABL, SEHIP_C3_W8, SEHIP_W3_ABL, SEHIP_L2_ABL:
    wrong results by design) that the product library does not contain."""
    if tools and "-DSEHIP_TOOLS_BUILD" not in FLAGS:
        FLAGS.append("-DSEHIP_TOOLS_BUILD")
        force = True
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(lambda n: _compile(n, force), sources()))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[sehip] built {LIB} from {len(objs)} objects")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tools="--tools" in sys.argv)
