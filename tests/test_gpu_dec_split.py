"""The opt-in split of the deep decoders' forward products (SEHIP_DEC_SPLIT=1, sehip.plan.DCCRNStatic._maybe_split_decoder: the
skip-connection half on the weight-gradient stream under the LSTM, the main half adding the bf16 partial tile BEFORE it takes the
fused BatchNorm sums -- conv_gemm_v3's residual-before-staging path, csrc/conv3.hip) against the default two-source products on the
same weights and inputs: the same network up to one more bf16 rounding of a partial sum, and against the fp32 oracle with the bound
of the default path.  (Measured slower in the step -- the LSTM loses more beside the skip halves than the main halves gain -- and
therefore not the default; this test keeps the kernel path alive and correct.)"""
import os

import pytest
import torch

from oracle import dccrn_oracle as O
from util import rel_err

pytestmark = pytest.mark.gpu


def _run(kw, split, B, seed=0):
    from sehip.model import DCCRN
    from sehip.loss import loss_sisdr
    old = os.environ.get("SEHIP_DEC_SPLIT")
    if split:
        os.environ["SEHIP_DEC_SPLIT"] = "1"
    else:
        os.environ.pop("SEHIP_DEC_SPLIT", None)
    try:
        torch.manual_seed(seed)
        model = DCCRN(**kw).cuda().train()
        g = torch.Generator().manual_seed(1)
        clean = 0.1 * torch.randn(B, 1, kw["length"], generator=g)
        noisy = clean + 0.05 * torch.randn(B, 1, kw["length"], generator=g)
        est = model(noisy.cuda())
        loss = loss_sisdr(est, clean.cuda())
        loss.backward()
        torch.cuda.synchronize()
        ws = model.workspace(B, kw["length"])
        assert bool(model.static.dec_split) == split
        return dict(model=model, ws=ws, est=est.detach().cpu(), loss=float(loss), grads=model.flat_grads.detach().cpu().clone(),
                    noisy=noisy, clean=clean, sd={k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    finally:
        if old is None:
            os.environ.pop("SEHIP_DEC_SPLIT", None)
        else:
            os.environ["SEHIP_DEC_SPLIT"] = old


@pytest.mark.parametrize("kw,B", [(dict(kernel_num=[16, 16, 32, 32, 64, 64], rnn_units=128, length=4000), 2),
                                  (dict(kernel_num=[16, 32, 64, 128, 256, 256], rnn_units=128, length=8000), 3)])
def test_split_decoder_products_match_the_two_source_products(kw, B):
    a, b = _run(kw, False, B), _run(kw, True, B)
    assert b["model"].static.dec_split == ([0] if kw["kernel_num"][5] == 64 else [0, 1, 2])
    for j in b["model"].static.dec_split:                    # the layer's stored pre-BatchNorm tensor: one more rounding of a partial sum
        ya, yb = a["ws"].bufs[f"yd{j}"].t.float().cpu(), b["ws"].bufs[f"yd{j}"].t.float().cpu()
        e = rel_err(yb, ya)
        print(f"decoder {j}: split vs two-source pre-BatchNorm tensor rel {e:.2e}")
        assert e < 7e-3, (j, e)          # (two independent roundings of the stored tensor, 1.66e-3 each, + the partial tile's)
    e_est, e_grad = rel_err(b["est"], a["est"]), rel_err(b["grads"], a["grads"])
    print(f"split vs default: waveform rel {e_est:.2e}, |loss a - loss b| {abs(a['loss'] - b['loss']):.4f} dB, gradient rel {e_grad:.2e}")
    # The loss bound.  A random-initialised network's estimate is almost orthogonal to the target: SI-SDR = -45 dB here, i.e. the
    # target component is 5.6e-3 of the estimate, and a 3.5e-3 relative change of the waveform moves the loss by tenths of a dB.  The
    # DEFAULT path alone does so from run to run (its BatchNorm sums are fp32 atomics: 44.70 / 44.75 / 44.82 dB for the same seed over
    # three processes, gpurun_out trail of round 6), so 0.05 dB -- this test's first bound -- failed in 3 of 8 runs with |delta| 0.12-0.17 dB
    # and gradient rel 2.5e-2 ... 3.3e-2.  Bounds now: 4 x that jitter; the tight statements are the stored tensors above and the
    # op-local tests of the kernel path (tests/test_gpu_ops_local.py).
    assert e_est < 8e-3 and abs(a["loss"] - b["loss"]) < 0.5 and e_grad < 8e-2
    for k in a["sd"]:                                        # running statistics of the decoders' BatchNorm layers (the fused sums)
        if k.startswith("decoder.") and k.endswith(("RMr", "RMi", "RVrr", "RVii")):
            assert rel_err(b["sd"][k].float(), a["sd"][k].float()) < 5e-3, k
    # and against the fp32 oracle, with the bound of the default path (tests/test_gpu_dccrn_plan.py)
    p = {k: v for k, v in a["sd"].items() if not k.startswith(("stft.", "istft."))}
    for k in p:
        if k.endswith(("RMr", "RMi", "RVri")):
            p[k] = torch.zeros_like(p[k])
        if k.endswith(("RVrr", "RVii")):
            p[k] = torch.ones_like(p[k])
    ref = O.dccrn_forward(p, b["noisy"], O.DCCRNConfig(**kw), training=True)
    assert rel_err(b["est"], ref) < 3e-2
