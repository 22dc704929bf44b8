"""GPU: the complex LSTM layers run as a chunk pipeline on two streams (sehip/plan.py).  Chunking must not change a single
bit: the same kernels resume from the h / c records (forward) and the carried (dc, dh) state (backward)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def run_once(chunk, B, N, units=128):
    from sehip.model import DCCRN
    old = os.environ.get("SEHIP_LSTM_CHUNK")
    os.environ["SEHIP_LSTM_CHUNK"] = str(chunk)
    # bit equality needs a bit-reproducible network around the LSTM: the BatchNorm sums that the convolution epilogues take with
    # fp32 atomics (and the backward sums that meet in a few rows the same way) vary in the last bit from run to run, the separate passes do not
    # (and the comparison is between the chunked and the whole-sequence launches of the SAME per-layer kernels: the fused two-layer
    #  kernel of round 4, csrc/lstm2.hip, rounds layer 2's input once more and has its own test, tests/test_gpu_lstm_fused.py)
    fused = {k: os.environ.get(k) for k in ("SEHIP_NO_FUSE_STATS", "SEHIP_NO_FUSE_FINALIZE", "SEHIP_NO_LSTM_FUSE")}
    os.environ.update({k: "1" for k in fused})
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(3)
        model = DCCRN(rnn_units=units, kernel_num=[16, 16, 32, 32, 64, 64], length=N).to(dev).train()
        g = torch.Generator().manual_seed(11)
        x = (0.1 * torch.randn(B, 1, N, generator=g)).to(dev)
        out = model(x)
        out.backward(torch.ones_like(out) * 1e-3)
        torch.cuda.synchronize()
        ws = model.workspace(B, N)
        keep = {k: ws.bufs[k].t.clone() for k in ("P", "h1", "h2", "dz5l", "dpre1_r", "dpre2_i")}
        return keep, model.flat_grads.clone(), len(ws.lstm_chunks)
    finally:
        for k, v in fused.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if old is None:
            os.environ.pop("SEHIP_LSTM_CHUNK", None)
        else:
            os.environ["SEHIP_LSTM_CHUNK"] = old


@pytest.mark.parametrize("B,N,chunk,units", [(3, 6000, 16, 128), (17, 3000, 7, 128), (2, 6000, 40, 128), (5, 6000, 16, 256)])
def test_chunked_equals_whole_sequence(B, N, chunk, units):
    """(units = 256: the hidden-128 instance of the same kernels -- 512 threads, the carried state 4 H floats per thread pair)"""
    whole, gw, n1 = run_once(0, B, N, units)
    parts, gp, n2 = run_once(chunk, B, N, units)
    assert n1 == 1 and n2 > 1
    for k in whole:
        assert torch.equal(whole[k], parts[k]), k
    # the weight gradients accumulate with fp32 atomics (order varies run to run): equal up to that
    assert float((gw - gp).abs().max()) <= 1e-3 * float(gw.abs().max())
