"""Worker of tests/test_gpu_demucs.py::test_handoff_timeout_on_one_rank_skips_the_step_on_every_rank (launched by
torch.distributed.run, two ranks sharing cuda:0 over gloo): a small Demucs with BLSTM layers; the hand-off time-out is forced on
RANK 1 ONLY (test word 61 of the sync block).  Every rank saves what it saw: parameters / step counter after the poisoned step,
the fall-back flag after the Solver's health check, parameters after the next (healthy) step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-enhancement-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def main(out_dir):
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    from test_gpu_demucs import c3_config, SMALL
    cfg = c3_config(out_dir)
    for k, v in dict(SMALL, sources=["clean"]).items():
        setattr(cfg.model, k, v)
    torch.manual_seed(3)
    model = distrib.get_model(cfg.model)
    solver = Solver(cfg, model, distrib.get_optimizer(cfg.optim, model), distrib.get_loss_function(cfg.optim), device="gpu",
                    writer=ScalarLog())
    r = solver.rank
    g = torch.Generator().manual_seed(5)
    clean = 0.1 * torch.randn(4, 1, 2, 8000, generator=g)
    mix = clean[:, 0] + 0.05 * torch.randn(4, 2, 8000, generator=g)
    mx, sr = solver._prepare_batch(mix[2 * r:2 * r + 2], clean[2 * r:2 * r + 2])
    p0 = model.flat_params.detach().clone()
    ws = model.workspace(2, 8000)
    if r == 1:
        ws.lstm_sync[61] = -1            # the first hand-off wait of every persistent launch on THIS rank times out
    solver.train_step(mx, sr)
    torch.cuda.synchronize()
    out = {"guard_after_step": int(ws.lstm_sync[60]), "unchanged": bool(torch.equal(model.flat_params.detach(), p0)),
           "step_dev": int(solver.optimizer._step_dev.item())}
    solver._model_health()
    out["per_step_after_health"] = bool(model.static.lstm_per_step)
    out["lost_steps"] = getattr(solver, "lost_steps", 0)
    loss, _ = solver.train_step(mx, sr)
    torch.cuda.synchronize()
    out["params"] = model.flat_params.detach().cpu()
    out["moved"] = bool(not torch.equal(model.flat_params.detach(), p0))
    out["step_dev_2"] = int(solver.optimizer._step_dev.item())
    out["loss"] = float(loss)
    torch.save(out, os.path.join(out_dir, f"guard_r{r}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
