"""CPU: checkpoint interchange of the fused optimizer with torch.optim (src/solver.py:233-279 restores optimizer state
through load_state_dict; src/distrib.py:244-261 builds torch.optim.SGD / Adam)."""
import torch


def _tiny_model():
    from sehip.model import DCCRN
    torch.manual_seed(0)
    return DCCRN(kernel_num=[16, 16, 16, 16, 16, 16], length=1600)


def _fill_grads(params, seed):
    g = torch.Generator().manual_seed(seed)
    for p in params:
        p.grad = 0.01 * torch.randn(p.shape, generator=g)


def test_sgd_momentum_state_dict_loads_into_flat_optimizer():
    from sehip.optim import FlatOptimizer
    model = _tiny_model()
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    ref = torch.optim.SGD(ref_params, lr=0.01, momentum=0.9)
    _fill_grads(ref_params, 1)
    ref.step()
    sd = ref.state_dict()
    assert "step" not in sd["state"][0]                     # the case that used to raise KeyError
    opt = FlatOptimizer(model, lr=0.5, kind="sgd", momentum=0.0)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 0.01 and opt.param_groups[0]["momentum"] == 0.9
    L = model.static.layout
    for (name, p), rp in zip(model._params, ref_params):
        off, _ = L.param_off[name]
        assert torch.equal(opt._m[off:off + p.numel()].view(p.shape), ref.state[rp]["momentum_buffer"])
    out = opt.state_dict()                                   # and back out in torch.optim's format
    assert torch.equal(out["state"][0]["momentum_buffer"], sd["state"][0]["momentum_buffer"])


def test_adam_state_dict_round_trip():
    from sehip.optim import FlatOptimizer
    model = _tiny_model()
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    ref = torch.optim.Adam(ref_params, lr=3e-4)
    for s in range(3):
        _fill_grads(ref_params, 10 + s)
        ref.step()
    opt = FlatOptimizer(model, lr=1.0, kind="adam")
    opt.load_state_dict(ref.state_dict())
    assert opt._step == 3
    L = model.static.layout
    name, p = model._params[5]
    off, _ = L.param_off[name]
    assert torch.equal(opt._v[off:off + p.numel()].view(p.shape), ref.state[ref_params[5]]["exp_avg_sq"])
