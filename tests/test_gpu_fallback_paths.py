"""GPU: the specialised kernels (LDS-patch conv GEMMs, small-channel / 2-channel kernels, paired launches, LSTM chunk pipeline)
against the table-gathered generic kernels that remain the fallback for shapes they do not take.  The tuning variables are
read once per process, so each configuration runs in a child process."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, torch
sys.path.insert(0, os.path.join(%(root)r, "speech-enhancement-pytorch_amd")); sys.path.insert(0, %(root)r)
from sehip.model import DCCRN
dev = torch.device("cuda:0")
torch.manual_seed(21)
model = DCCRN(rnn_units=128, kernel_num=[16, 32, 64, 128, 128, 128], length=6000).to(dev).train()
g = torch.Generator().manual_seed(22)
x = (0.1 * torch.randn(4, 1, 6000, generator=g)).to(dev)
out = model(x)
out.backward(1e-3 * torch.ones_like(out))
torch.cuda.synchronize()
named = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
torch.save({"out": out.detach().cpu(), "grads": model.flat_grads.cpu(), "named": named}, sys.argv[1])
"""


def run(env_extra):
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "r.pt")
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return torch.load(path)


def test_specialised_kernels_match_generic_fallback():
    fast = run({})
    slow = run({"SEHIP_NO_PATCH": "1", "SEHIP_NO_NARROW": "1", "SEHIP_NO_PAIR": "1", "SEHIP_NO_BM64": "1", "SEHIP_LSTM_CHUNK": "0",
                "SEHIP_NO_SIDE_STREAM": "1", "SEHIP_NO_WGRAD_GROUP": "1", "SEHIP_NO_FUSE_STATS": "1"})
    # same bf16 rounding points, different fp32 summation order: outputs agree far inside the bf16 noise of the chain
    eo = float((fast["out"] - slow["out"]).norm() / slow["out"].norm())
    eg = float((fast["grads"] - slow["grads"]).norm() / slow["grads"].norm())
    assert eo < 2e-2, eo
    assert eg < 5e-2, eg


def test_conv_wgrad_v3_all_layer_classes_and_both_flushes_match_conv_wgrad():
    """conv_wgrad_v3_kernel (csrc/wgrad3.hip: LDS-DMA stages, one CU per workgroup, partial arrays + reduction) takes the 5-tap encoder
    layers by default; here every class it is built for (3- / 2-tap decoder products too), one workgroup per CU, with the store
    flush and with the atomic flush, against conv_wgrad_kernel on the same operands (enc3 / enc4 / enc5, dec0 / dec1 of this model
    qualify: >= 64 input channels, 128 output columns, 16 / 8 / 4 rows per frame)."""
    old, old2 = run({"SEHIP_NO_WGRAD_V3": "1"}), run({"SEHIP_NO_WGRAD_V3": "1"})
    convs = [n for n in old["named"] if n.endswith("conv.weight") and old["named"][n].dim() == 4]
    assert len(convs) >= 20

    def worst(a, b):
        return max(float((a["named"][n] - b["named"][n]).norm() / b["named"][n].norm()) for n in convs)

    # the forward pass's BatchNorm sums are fp32 atomics: two runs of the SAME build differ by the bf16 roundings that flips -- the
    # yardstick for "same operands, different summation order"
    noise = worst(old2, old)
    bound = max(5 * noise, 5e-3)
    for extra in ({}, {"SEHIP_W3_ATOMIC_FLUSH": "1"}, {"SEHIP_W3_NB": "3", "SEHIP_W3_WGS": "96"}):
        new = run(dict({"SEHIP_W3_CLASSES": "7", "SEHIP_W3_WGS": "256"}, **extra))
        assert worst(new, old) < bound, (extra, worst(new, old), noise)      # every convolution weight tensor on its own
    assert worst(run({}), old) < bound


def test_batchnorm_pass_variants_agree():
    """The ComplexBatchNorm layer as the step runs it (sums out of the convolution epilogues of all widths, the apply pass finalizing
    them) against the separate passes (cbn_stats + cbn_finalize + cbn_apply) and against the opt-in two-launch backward pass
    (sehip_cbn_bwd_fused): same tensors, different summation orders -- every parameter gradient on its own, yardstick = two runs of
    the default."""
    a, b = run({}), run({})
    # (the PReLU slope gradients are scalars that nearly cancel under this upstream gradient -- two runs of ONE build differ by 16 % --
    #  and are checked against the oracle layer by layer in test_gpu_ops_local.py: left out here)
    names = [n for n in a["named"] if float(a["named"][n].norm()) > 0 and a["named"][n].numel() > 1]
    assert len(names) > 100

    def err(x, y, n):
        return float((x["named"][n] - y["named"][n]).norm() / y["named"][n].norm())

    noise = {n: err(b, a, n) for n in names}
    # a different rounding of the batch statistics moves bf16 activations: a few times the atomics' own noise.  (Round 6: the floor was 3e-2 and
    # the suite failed here ONCE in ~30 runs; measured over 8 x 3 comparisons on one box: the worst tensor is the 16-element decoder.4.1.Wri /
    # .Br of the separate-passes variant at 2.0e-2 ... 2.5e-2 with a run-to-run noise of 3e-3 ... 7e-3 -- 0.8 of the old floor.)
    floor = 6e-2
    for env in ({"SEHIP_NO_FUSE_STATS": "1", "SEHIP_NO_FUSE_FINALIZE": "1"}, {"SEHIP_NO_FUSE_STATS32": "1"}, {"SEHIP_FUSE_BWD_FINALIZE": "1"}):
        other = run(env)
        assert float((other["out"] - a["out"]).norm() / a["out"].norm()) < 1e-2, env
        for n in names:
            assert err(other, a, n) < max(5 * noise[n], floor), (env, n, err(other, a, n), noise[n])


CHILD_DCU = r"""
import os, sys, torch
sys.path.insert(0, os.path.join(%(root)r, "speech-enhancement-pytorch_amd")); sys.path.insert(0, %(root)r)
from sehip.model import DCUnet
dev = torch.device("cuda:0")
torch.manual_seed(31)
model = DCUnet(data_type=True, model_complexity=45, model_depth=10).to(dev).train()
g = torch.Generator().manual_seed(32)
x = (0.1 * torch.randn(2, 1, 257, 65, 2, generator=g)).to(dev)
out = model(x)
out.backward(1e-2 * out.detach())
torch.cuda.synchronize()
torch.save({"out": out.detach().cpu(), "grads": model.flat_grads.cpu()}, sys.argv[1])
"""


def test_dcunet_patch_weight_gradient_matches_generic():
    """conv_wgrad2_kernel (LDS patch, all taps per workgroup; frames of 129 / 128 rows at this size) against the table-gathered
    wgrad_kernel on the full-width DCUnet-10: same operands, different fp32 summation order."""
    def run_dcu(env_extra):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "r.pt")
            env = dict(os.environ)
            env.update(env_extra)
            r = subprocess.run([sys.executable, "-c", CHILD_DCU % {"root": ROOT}, path], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            return torch.load(path)
    fast, slow = run_dcu({}), run_dcu({"SEHIP_NO_WGRAD2": "1"})
    assert float((fast["out"] - slow["out"]).abs().max()) == 0.0            # the forward pass is untouched
    d = (fast["grads"] - slow["grads"]).norm() / slow["grads"].norm()
    assert 0.0 < float(d) < 5e-3, float(d)                                   # not bit-identical: it really is the other kernel


CHILD_DMX = r"""
import os, sys, torch
sys.path.insert(0, os.path.join(%(root)r, "speech-enhancement-pytorch_amd")); sys.path.insert(0, %(root)r)
from sehip.model import Demucs
dev = torch.device("cuda:0")
torch.manual_seed(41)
model = Demucs(sources=["a", "b"], audio_channels=2, channels=32, depth=4, norm_starts=2, dconv_lstm=2, dconv_attn=2).to(dev).train()
with torch.no_grad():
    for name, prm in model.named_parameters():
        if name.endswith(".scale"):
            prm.fill_(0.3)
g = torch.Generator().manual_seed(42)
x = (0.3 * torch.randn(2, 2, 16000, generator=g)).to(dev)
out = model(x)
out.backward(1e-3 * torch.ones_like(out))
torch.cuda.synchronize()
model.workspace(2, 16000).check_lstm_handoffs()
torch.save({"out": out.detach().cpu(), "grads": model.flat_grads.cpu()}, sys.argv[1])
"""


def test_demucs_persistent_lstm_and_second_stream_match_the_plain_schedule():
    """Demucs with a chunked and an unchunked BLSTM level: the persistent hand-off LSTM kernels + second-stream packing / weight
    gradients against one launch per time step on a single stream (SEHIP_DMX_LSTM_STEPS, SEHIP_NO_SIDE_STREAM)."""
    def run_dmx(env_extra):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "r.pt")
            env = dict(os.environ)
            env.update(env_extra)
            r = subprocess.run([sys.executable, "-c", CHILD_DMX % {"root": ROOT}, path], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            return torch.load(path)
    fast = run_dmx({})
    slow = run_dmx({"SEHIP_DMX_LSTM_STEPS": "1", "SEHIP_NO_SIDE_STREAM": "1"})
    eo = float((fast["out"] - slow["out"]).norm() / slow["out"].norm())
    eg = float((fast["grads"] - slow["grads"]).norm() / slow["grads"].norm())
    assert eo < 2e-2, eo        # same arithmetic; the fp64 / fp32 atomics' order flips bf16 roundings from run to run
    assert eg < 5e-2, eg
