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
        if "+" in name:          # "dec0.fwd0+dec0.fwd1": the pair through sehip_gemm_pair (one launch where the library merges them)
            na, nb = name.split("+")
            da, db = ws.desc[na], ws.desc[nb]
            run = lambda: call("sehip_gemm_pair", C.byref(da), C.byref(db), stream())
            d = da
        else:
            d = ws.desc[name]
            run = lambda: call(fn, C.byref(d), stream())
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
        if "+" in name: continue
        if os.environ.get("SEHIP_VARIANT_SUMS") and fn == "sehip_gemm":      # digest of what ONE call writes (outputs + fused sums)
            B = 32
            for q in (0, 1):
                t = d.dst[q]
                if not t.ptr: continue
                n = B * t.T * t.F * t.C
                buf = (C.c_uint32 if t.is_f32 else C.c_uint16) * n
                # a view on the arena through the CUDA array interface
                class V: pass
                v = V(); v.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4" if t.is_f32 else "<i2", "data": (t.ptr, False), "version": 2}
                x_ = torch.as_tensor(v, device=dev)
                x_ = x_ if t.is_f32 else x_.view(torch.bfloat16)
                x_.zero_()
            if d.stats:
                v = V(); v.__cuda_array_interface__ = {"shape": (8 * 5 * d.stats_cr,), "typestr": "<f4", "data": (d.stats, False), "version": 2}
                st_ = torch.as_tensor(v, device=dev); st_.zero_()
            call(fn, C.byref(d), stream()); torch.cuda.synchronize()
            sums = []
            for q in (0, 1):
                t = d.dst[q]
                if not t.ptr: continue
                v = V(); v.__cuda_array_interface__ = {"shape": (B * t.T * t.F * t.C,), "typestr": "<f4" if t.is_f32 else "<i2", "data": (t.ptr, False), "version": 2}
                x_ = torch.as_tensor(v, device=dev)
                x_ = (x_ if t.is_f32 else x_.view(torch.bfloat16)).double()
                w_ = torch.arange(x_.numel(), device=dev, dtype=torch.float64) % 977 + 1
                sums += [float(x_.abs().sum()), float((x_ * w_).sum())]
            if d.stats:
                sums += [float(st_.double().abs().sum())]
            res[name + ".sum"] = sums
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
        if line:
            out = json.loads(line[0][4:])
            print(flags, {k: round(v * 1e3) for k, v in out.items() if not k.endswith(".sum")})
            for k, v in out.items():
                if k.endswith(".sum"): print("   ", k, ["%.6e" % q for q in v])
        else:
            print(flags, r.stderr[-1500:])
