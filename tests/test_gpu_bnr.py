"""GPU: the ComplexBatchNorm backward REDUCE pass computed inside the streaming launch that produces the layer's activation gradient
(sehip_gemm_desc.bnr_*, csrc/convt.hip: decoder 5 / 4 input gradients at the headline widths) against the separate pass it replaces
(sehip_cbn_bwd_reduce).  Same arithmetic on the same bf16 values, per-workgroup partial rows instead of per-block ones: the sums the
finalize pass forms from them -- and therefore every parameter gradient of the network -- must agree to summation-order noise.
Reference math: src/model/dccrn.py:457-634 (ComplexBatchNorm), torch.nn.PReLU; against the oracle the whole step is checked on the
fused path (the default) by tests/test_gpu_c1_fullsize.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_once(fuse, B, N, monkeypatch):
    from sehip.model import DCCRN
    from sehip.utils import set_deterministic
    if fuse:
        monkeypatch.delenv("SEHIP_NO_BNR", raising=False)
    else:
        monkeypatch.setenv("SEHIP_NO_BNR", "1")
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    set_deterministic(True)            # no fp32 atomics upstream: the two runs see the same gradients to the last bit
    try:
        model = DCCRN(rnn_units=128, kernel_num=[16, 32, 64, 128, 256, 256], length=N).to(dev).train()
        model.set_deterministic(True)
        g = torch.Generator().manual_seed(17)
        x = (0.1 * torch.randn(B, 1, N, generator=g)).to(dev)
        ws = model.workspace(B, N)
        out = model(x)
        out.backward(torch.ones_like(out) * 1e-3)
        torch.cuda.synchronize()
        L = model.static.layout
        gflat = model.flat_grads.detach().cpu().clone()
        grads = {}
        for name in L.param_names:
            off, shape = L.param_off[name]
            grads[name] = gflat[off:off + int(np.prod(shape))].reshape(shape)
        return grads, ws
    finally:
        set_deterministic(False)


@pytest.mark.parametrize("B,N", [(3, 16000), (16, 8000)])
def test_reduce_inside_the_producer_equals_the_separate_pass(B, N, monkeypatch):
    ga, wa = run_once(True, B, N, monkeypatch)
    gb, wb = run_once(False, B, N, monkeypatch)
    assert set(wa.bnr_rows) == {"decoder.3.", "decoder.4."} and not wb.bnr_rows
    # decoder 4's sums are formed from bit-identical inputs in both runs (nothing upstream of them depends on the choice): only the
    # grouping of the fp32 partial sums differs (one row per workgroup of ~40 frames instead of per block of rows; the rows are then
    # added in double precision by the same kernel)
    for k in ("1.Wrr", "1.Wri", "1.Wii", "1.Br", "1.Bi", "2.weight"):
        a, b = ga["decoder.4." + k], gb["decoder.4." + k]
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-30), ("decoder.4." + k, float((a - b).abs().max()), float(b.abs().max()))
    # everything downstream sees those 1e-6-level differences through bf16 stores and PReLU kinks (an occasional element flips by one
    # bf16 ulp: DESIGN section 2): the same computation to that noise.  Tensors whose gradient is analytically zero -- a convolution
    # bias in front of a BatchNorm -- hold rounding residue only and are left out; every other tensor is measured against the larger
    # of its own largest entry and 1e-3 of the network's
    top = max(float(v.abs().max()) for v in gb.values())
    worst, where = 0.0, None
    for n in ga:
        if n.endswith("_conv.bias") and not n.startswith("decoder.5."):
            continue
        scale = max(float(gb[n].abs().max()), 1e-3 * top)
        r = float((ga[n] - gb[n]).abs().max()) / scale
        if r > worst:
            worst, where = r, n
    assert worst < 3e-2, (worst, where)


def test_library_reports_no_rows_for_products_it_does_not_fuse():
    import ctypes as C
    from sehip import _lib
    from sehip.plan import CGemmDesc
    d = CGemmDesc()                                    # nothing described: not a streaming product
    assert _lib.lib().sehip_bnr_rows(C.byref(d), None) == 0
