"""GPU: the first encoder layer's ComplexBatchNorm + PReLU backward apply pass computed INSIDE its weight gradient
(sehip_gemm_desc.bn_dz ..., csrc/gemm.hip narrow_wgrad_mfma_kernel<5, 2, true>) against the two launches it replaces (sehip_cbn_bwd_apply
storing dOut, then the weight gradient reading it).  Same arithmetic in the same order on the same bf16 inputs, dOut rounded to bf16
at the same point: the two schedules must agree to the last few bits.  Reference math: src/model/dccrn.py:139-167 (the layer),
:457-634 (ComplexBatchNorm), torch.nn.PReLU; against the oracle the whole step is checked on the fused path (the default) by
tests/test_gpu_c1_fullsize.py and tests/test_gpu_ops_local.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_once(fuse, B, N, kernel_num, monkeypatch):
    """One forward + backward on the DETERMINISTIC schedule (sehip_set_deterministic: no fp32 atomics anywhere), so that the two runs see
    the same upstream gradient to the last bit and differ only in where the apply arithmetic runs."""
    from sehip import plan
    from sehip.model import DCCRN
    from sehip.utils import set_deterministic
    monkeypatch.setattr(plan, "FUSE_ENC0_BN_WGRAD", bool(fuse))
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    set_deterministic(True)
    try:
        model = DCCRN(rnn_units=128, kernel_num=list(kernel_num), length=N).to(dev).train()
        model.set_deterministic(True)
        g = torch.Generator().manual_seed(13)
        x = (0.1 * torch.randn(B, 1, N, generator=g)).to(dev)
        ws = model.workspace(B, N)
        assert ws.enc0_bn_in_wgrad == bool(fuse)
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


@pytest.mark.parametrize("B,N,kn", [(2, 6000, (16, 16, 32, 32, 64, 64)), (3, 16000, (16, 32, 64, 128, 256, 256)),
                                   (16, 8000, (16, 32, 64, 64, 128, 128))])
def test_apply_inside_the_weight_gradient_equals_the_two_launches(B, N, kn, monkeypatch):
    ga, wa = run_once(True, B, N, kn, monkeypatch)
    gb, wb = run_once(False, B, N, kn, monkeypatch)
    # what the fused kernel produces: the layer's convolution weights and bias (the BatchNorm / PReLU parameter gradients come from
    # the reduce + finalize launches, which both schedules run)
    names = [n for n in ga if n.startswith("encoder.0.0.")]
    assert names
    for n in names:
        a, b = ga[n], gb[n]
        scale = float(b.abs().max()) + 1e-30
        # fp32 sums of bf16 products in the same order; only the compiler's choice of fused multiply-adds inside the apply
        # arithmetic may differ between the two kernels: one bf16 ulp of an occasional dOut element
        assert float((a - b).abs().max()) <= 2e-3 * scale, (n, float((a - b).abs().max()), scale)
    # every other gradient of the network does not depend on the choice at all
    for n in ga:
        if n not in names:
            assert torch.equal(ga[n], gb[n]), n


def test_library_refuses_the_operands_where_no_kernel_applies_them():
    """sehip_wgrad with bn_dz set on a product the narrow MFMA kernel does not take: a named error, not a silent plain weight gradient."""
    import ctypes as C
    from sehip import _lib
    from sehip.plan import CGemmDesc
    dev = torch.device("cuda:0")
    d = CGemmDesc()
    buf = torch.zeros(1 << 16, dtype=torch.bfloat16, device=dev)
    f32 = torch.zeros(1 << 12, dtype=torch.float32, device=dev)
    tab = torch.zeros(1 << 10, dtype=torch.int32, device=dev)
    d.src[0].ptr, d.src[0].T, d.src[0].F, d.src[0].C, d.src[0].tlo, d.src[0].thi = buf.data_ptr(), 4, 8, 64, 0, 4
    d.dst[0].ptr, d.dst[0].T, d.dst[0].F, d.dst[0].C, d.dst[0].fmul = buf.data_ptr(), 4, 8, 128, 1
    d.ktab, d.ntab, d.W, d.dW = tab.data_ptr(), tab.data_ptr(), buf.data_ptr(), f32.data_ptr()
    d.M, d.N, d.Npad, d.K, d.TT, d.J, d.fmul = 32, 128, 128, 64, 4, 8, 1
    d.bn_dz, d.bn_y, d.bn_coef, d.bn_bcoef, d.bn_slope = buf.data_ptr(), buf.data_ptr(), f32.data_ptr(), f32.data_ptr(), f32.data_ptr()
    rc = _lib.lib().sehip_wgrad(C.byref(d), None)
    assert rc != 0
    assert b"bn_dz" in _lib.lib().sehip_last_error()
