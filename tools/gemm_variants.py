#!/usr/bin/env python3
"""A/B of launch variants of single products on ONE GPU box: the tuning variables are read once per process, so each variant
runs in a child process on the same device, in the order given.  Usage: SEHIP_NAMES=enc3.fwd,dec0.dg python
tools/gemm_variants.py base SEHIP_CW_SPLITS=16 lib:tools/_prev.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd")); sys.path.insert(0, ROOT)
    import torch, ctypes as C
    from sehip.model import DCCRN
    from sehip._lib import call, stream
    names = sys.argv[2:]
    dev = torch.device("cuda:0"); torch.manual_seed(0)
    model = DCCRN(length=32000).to(dev).train()
    x = (0.1 * torch.randn(32, 1, 32000)).to(dev)
    out = model(x); out.backward(torch.randn_like(out) * 1e-3); torch.cuda.synchronize()
    ws = model.workspace(32, 32000)
    res = {}
    for name in names:
        fn = "sehip_wgrad" if name.endswith(".wg") else "sehip_gemm"
        d = ws.desc[name]
        for _ in range(3): call(fn, C.byref(d), stream())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call(fn, C.byref(d), stream())
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
    print("RES " + json.dumps(res))
else:
    names = os.environ.get("SEHIP_NAMES", "").split(",") if os.environ.get("SEHIP_NAMES") else ["dec1.dg", "dec0.fwd0", "enc5.fwd", "dec0.fwd0.wg", "enc4.fwd.wg", "enc1.fwd", "enc2.fwd", "enc1.dg0", "enc2.dg0", "dec3.fwd0", "dec4.fwd0", "dec5.fwd0", "dec4.dg", "enc0.fwd", "dec5.dg", "enc1.fwd.wg", "enc2.fwd.wg", "dec4.fwd0.wg", "dec3.fwd0.wg", "dec5.fwd0.wg", "enc0.fwd.wg"]
    # a variant is "base", "lib:<path to another libsehip.so>" or NAME=VALUE (any SEHIP_* tuning variable)
    for flags in sys.argv[1:]:
        env = dict(os.environ)
        if flags.startswith("lib:"):
            env["SEHIP_LIB"] = os.path.join(ROOT, flags[4:])
        elif "=" in flags:
            k, v = flags.split("=", 1)
            env[k] = v
        r = subprocess.run([sys.executable, __file__, "--child"] + names, env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES ")]
        print(flags, {k: round(v * 1e3) for k, v in json.loads(line[0][4:]).items()} if line else r.stderr[-500:])
