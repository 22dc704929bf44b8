"""CPU: ConvTasNet module schema against the reference's checkpoint keys, the gradient un-packing table, rejections."""
import numpy as np
import pytest
import torch

from util import load_golden


def test_schema_and_unpack_table():
    from sehip.model import ConvTasNet
    from sehip import SehipError, distrib, utils
    g = load_golden("convtasnet_tiny.npz")
    ref = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    m = ConvTasNet(sources=["None", "None"], N=16, L=8, B=16, H=32, P=3, X=3, R=2, audio_channels=1)
    sd = m.state_dict()
    assert list(sd) == list(ref) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in ref.items())
    assert [n for n, _ in m.named_parameters()] == list(ref)              # optimizer state indices interchange
    m.load_state_dict(ref)
    st = m.static
    L = st.layout
    real = np.concatenate([L.index_array(n).reshape(-1) for n in L.param_names])
    assert (st.utab[real, 0] >= 0).all() and (st.utab[:, 1:] == -1).all()   # exactly one packed-gradient entry per parameter
    assert len(np.unique(st.utab[real, 0])) == len(real)
    # xavier_normal_ also hits the [1, C, 1] LayerNorm tensors in the reference (src/model/conv_tasnet.py:132-134)
    m2 = ConvTasNet(sources=["None", "None"], audio_channels=1)
    assert float(m2.state_dict()["separator.network.0.gamma"].std()) > 0.05
    with pytest.raises(SehipError):
        m(torch.zeros(1, 1, 400))                                           # CPU tensor: no fallback
    with pytest.raises(SehipError):
        ConvTasNet(sources=["None"], skip=True)
    opt = distrib.get_optimizer(utils.dict2obj({"optim": "adam", "lr": 3e-4, "beta1": 0.9, "beta2": 0.999}), m)
    assert len(opt.state_dict()["state"]) == 60
