"""GPU: the drop-in module against the oracle over the shapes and options the reference accepts -- batch 1, batches that are
not a multiple of any tile, short / odd-length clips, clips longer than `length` (truncated like the reference's
[..., :length]), the three masking modes, narrow and wide channel configurations, eval mode."""
import pytest
import torch

from oracle import dccrn_oracle as O
from util import rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # (batch, samples, ctor kwargs)
    (1, 1600, dict(kernel_num=[16, 16, 16, 16, 16, 16], length=1600)),
    (5, 6000, dict(kernel_num=[16, 32, 32, 32, 64, 64], length=6000)),
    (2, 4100, dict(kernel_num=[16, 16, 32, 32, 64, 64], length=4000)),          # input longer than `length`
    (2, 4000, dict(kernel_num=[16, 16, 32, 32, 64, 64], length=16384)),         # `length` longer than the synthesis
    (3, 3777, dict(kernel_num=[32, 32, 64, 64, 128, 128], length=3777)),        # odd length, wide layers
    (2, 4000, dict(kernel_num=[16, 16, 32, 32, 64, 64], length=4000, masking_mode="C")),
    (2, 4000, dict(kernel_num=[16, 16, 32, 32, 64, 64], length=4000, masking_mode="R")),
    (17, 2000, dict(kernel_num=[16, 16, 16, 32, 32, 32], length=2000)),         # more than one LSTM batch tile, ragged
    (1, 100, dict(kernel_num=[16, 16, 16, 16, 16, 16], length=100)),            # 4 frames (the reference pads by win-hop)
    # other frame geometries of the constructor (win_len, win_inc; src/model/dccrn.py:14-15): half overlap, a window as long as the transform
    (2, 4000, dict(kernel_num=[16, 16, 32, 32, 64, 64], length=4000, win_len=320, win_inc=160)),
    (3, 5000, dict(kernel_num=[16, 16, 32, 32, 64, 64], length=5000, win_len=512, win_inc=128, win_type="hamming")),
]


@pytest.mark.parametrize("b,n,kw", CASES)
def test_forward_loss_backward_vs_oracle(b, n, kw):
    from sehip.model import DCCRN
    from sehip.loss import loss_sisdr
    dev = torch.device("cuda:0")
    torch.manual_seed(b * 1000 + n)
    model = DCCRN(rnn_units=128, **kw).to(dev).train()
    g = torch.Generator().manual_seed(n)
    clean = 0.1 * torch.randn(b, 1, n, generator=g)
    noisy = clean + 0.05 * torch.randn(b, 1, n, generator=g)
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith(("stft.", "istft."))}
    est = model(noisy.to(dev))
    cfg = O.DCCRNConfig(rnn_units=128, **kw)
    names = [k for k in p if O.is_trainable(k)]
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    work = dict(p); work.update(leaves)
    ref = O.dccrn_forward(work, noisy, cfg, training=True, sim=O.Bf16Sim)
    assert est.shape == ref.shape, (est.shape, ref.shape)
    assert rel_err(est.detach().cpu(), ref.detach()) < 3e-2
    tgt = clean[..., :ref.shape[-1]]
    if tgt.shape[-1] < ref.shape[-1]:
        tgt = torch.nn.functional.pad(tgt, [0, ref.shape[-1] - tgt.shape[-1]])
    loss = loss_sisdr(est, tgt.to(dev))
    loss_ref = O.loss_sisdr(ref, tgt)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 0.15
    loss.backward()
    grads_ref = torch.autograd.grad(loss_ref, [leaves[k] for k in names])
    gn = torch.sqrt(sum((x.double() ** 2).sum() for x in grads_ref))
    got = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
    err = torch.sqrt(sum(((got[k].double() - gr.double()) ** 2).sum() for k, gr in zip(names, grads_ref)))
    assert float(err / gn) < 0.25, float(err / gn)   # whole-chain bf16 bound (see test_gpu_dccrn_plan.py)
    # eval mode uses the running statistics the training forward just updated
    model.eval()
    with torch.no_grad():
        out_eval = model(noisy.to(dev))
    p2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith(("stft.", "istft."))}
    ref_eval = O.dccrn_forward(p2, noisy, cfg, training=False, sim=O.Bf16Sim)
    assert rel_err(out_eval.cpu(), ref_eval) < 3e-2


def test_rejects_what_is_not_built():
    from sehip.model import DCCRN
    from sehip import SehipError
    for bad in (dict(rnn_units=80), dict(kernel_num=[8, 16, 32, 64, 128, 128]), dict(kernel_size=3), dict(use_clstm=False, rnn_units=256),
                dict(win_type="no-such-window")):       # (any scipy.signal.get_window name / None is built since round 6)
        with pytest.raises(SehipError):
            DCCRN(**bad)
    dev = torch.device("cuda:0")
    m = DCCRN(length=4000).to(dev)
    with pytest.raises(SehipError):
        m(torch.zeros(1, 1, 4000))                 # CPU tensor: there is no CPU path
