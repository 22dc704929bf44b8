"""The LSTM weight gradients as ONE streaming dense-row launch (csrc/dtw.hip, sehip_wgrad_dense_group) against the fp32 products of the
same operands computed with torch on the device (the operands ARE bf16 tensors: the reference is exact up to summation order), at the
headline shape (B = 32, T = 323: 10 336 rows, not a multiple of the 64-row stage; the recurrent products read h[t - 1] with a zero
at t = 0; layer 2's input is two tensors; layer 1's input is four strided pieces of the encoder output) and at a small ragged one.
HIP vs torch on the device, not an oracle test: the oracle comparisons of these gradients are tests/test_gpu_c1_fullsize.py and
tests/test_gpu_solver.py, which run through this launch by default."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, n):
    from sehip.model import DCCRN
    from sehip._lib import call, stream
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = DCCRN(length=n).to(dev).train()
    x = (0.1 * torch.randn(B, 1, n)).to(dev)
    out = model(x)
    out.backward(torch.randn_like(out) * 1e-2)
    torch.cuda.synchronize()
    ws = model.workspace(B, n)
    names = ws._lstm_wgrad_names((2, 1))
    buf, k, total, dense = ws._wgrad_group_handle(names)
    assert dense is not None, "the LSTM products must qualify for the dense-row launch"
    T, h = ws.T, ws.st.cfg.hid
    M = B * T
    ws.gpack.zero_()
    call("sehip_wgrad_dense_group", dense[0].data_ptr(), k, C.cast(dense[1], C.c_void_p), ws._dtw_scratch.data_ptr(), stream())
    torch.cuda.synchronize()
    got = ws.gpack.clone()
    worst = 0.0
    for nm in names:
        d = ws.desc[nm + ".wg"]
        s = ws.st.specs[nm]
        N, K = d.Npad, d.K
        off = (d.dW - ws.gpack.data_ptr()) // 4
        dw = got[off:off + N * K].view(N, K).double()
        # dOut: the destination buffer's columns; A: gathered by the spec's sources
        gname = s.dsts[0][0] if s.kind == "wgrad_only" else ws._grad_buffer_of(s.dsts[0][0])     # (the .wg twin reads the gradient buffer)
        gb = ws.bufs[gname].t.reshape(M, -1).double()
        if nm.startswith("hh"):
            layer, combo = int(nm[2]), int(nm[4])
            l = combo & 1
            g = gb[:, l * 4 * h:(l + 1) * 4 * h]
            hbuf = ws.bufs[f"h{layer}_{combo}"].t.reshape(B, T, h).double()
            a = torch.zeros_like(hbuf)
            a[:, 1:] = hbuf[:, :-1]
            a = a.reshape(M, h)
        elif nm.startswith("ih1"):
            q = 0 if nm.endswith("r") else 1
            g = gb
            z5 = ws.bufs["z5"].t.reshape(M, 4, -1).double()
            cp = z5.shape[-1] // 2
            a = z5[:, :, q * cp:(q + 1) * cp].reshape(M, 4 * cp)
        else:
            g = gb
            a = torch.cat([ws.bufs[b].t.reshape(M, h).double() for b, _ in s.srcs], 1)
        ref = g.t() @ a
        err = float((dw - ref).norm() / ref.norm())
        worst = max(worst, err)
        assert err < 2e-5, (nm, err)
        if d.dbias:                                     # b_ih + b_hh: the column sums of dOut
            boff = (d.dbias - ws.gpack.data_ptr()) // 4
            db = got[boff:boff + N].double()
            refb = g.sum(0)
            errb = float((db - refb).norm() / refb.norm())
            assert errb < 2e-5, (nm, "dbias", errb)
    return worst


def test_lstm_weight_gradients_dense_group_headline_shape():
    _run(32, 32000)


def test_lstm_weight_gradients_dense_group_small_ragged():
    _run(3, 8000)
