"""GPU: the drop-in module / fused optimizer / Solver loop against two oracle train steps (same weights, same batches)."""
import numpy as np
import pytest
import torch

from oracle import dccrn_oracle as O
from util import rel_err, max_abs

pytestmark = pytest.mark.gpu
KW = dict(kernel_num=[16, 16, 32, 32, 64, 64], rnn_units=128, length=4000)


def make_batch(seed, b, n):
    g = torch.Generator().manual_seed(seed)
    clean = 0.1 * torch.randn(b, 1, 1, n, generator=g)
    noisy = clean[:, 0] + 0.05 * torch.randn(b, 1, n, generator=g)
    return noisy, clean


def solver_config(tmp, clip=5):
    from sehip.utils import dict2obj
    return dict2obj({
        "seed": 10, "root": None, "ha": None,
        "model": dict(name="dccrn", audio_channels=1, num_spk=1, **KW),
        "optim": {"optim": "adam", "lr": 3e-4, "beta1": 0.9, "beta2": 0.999, "loss": "si-sdr", "clip_grad": clip,
                  "pit": False, "load": True},
        "dset": {"name": "synthetic"},
        "solver": {"epochs": 1, "save_checkpoint_interval": 1, "all_steps": True, "total_steps": 0, "patience": 0,
                   "root": str(tmp), "resume": None, "preloaded_model": None,
                   "validation": {"interval": 1, "metric": "loss", "total_steps": 0}, "test": {"interval": 1}},
    })


def test_two_solver_steps_match_oracle(tmp_path):
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    cfg = solver_config(tmp_path)
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    p = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("stft.", "istft."))}
    opt = distrib.get_optimizer(cfg.optim, model)
    batches = []
    for s in range(2):
        noisy, clean = make_batch(100 + s, 2, 4000)
        batches.append((noisy, clean, [None], [None], ["x"], [s]))
    log = ScalarLog()
    solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), train_dataloader=batches,
                    validation_dataloader=[batches[0]], device="gpu", writer=log)
    solver.train()
    losses = [v for (t, v, _s) in log.scalars if t == "Train/Loss_step"]
    gnorms = [v for (t, v, _s) in log.scalars if t == "Train/grad_norm_step"]

    cfg_o = O.DCCRNConfig(**KW)
    adam = O.AdamState({k: v for k, v in p.items() if O.is_trainable(k)}, lr=3e-4)
    for s in range(2):
        loss, metric, _ = O.train_step(p, batches[s][0], batches[s][1][:, 0], cfg_o, adam, clip_grad=5)
        assert abs(loss - losses[s]) < 0.1, (s, loss, losses[s])          # dB; bf16 activations vs fp32 oracle
        assert abs(metric - gnorms[s]) < 0.15 * abs(metric) + 1e-3, (s, metric, gnorms[s])
    sd = {k: v.cpu() for k, v in solver.model.state_dict().items()}
    for k, v in p.items():
        if v.dtype == torch.int64:
            assert int(sd[k]) == int(v), k
        elif O.is_trainable(k):
            # two Adam steps move every weight by at most ~2*lr; direction agreement is what is checked
            assert max_abs(sd[k], v) < 1.3e-3, (k, max_abs(sd[k], v))
        elif k.endswith(("RMr", "RMi", "RVri")):
            # running means / cross-covariances are ~1e3x smaller than the channel scale: absolute bound
            assert max_abs(sd[k], v) < 3e-3, (k, max_abs(sd[k], v))
        else:
            assert rel_err(sd[k], v) < 3e-2, k
    # checkpoint files and keys (src/solver.py:295-341)
    ck = list((solver.checkpoints_dir).glob("*"))
    names = sorted(f.name for f in ck)
    assert "latest_model.tar" in names and "best_model.tar" in names and "state.json" in names
    tar = torch.load(solver.checkpoints_dir / "latest_model.tar", weights_only=False)
    assert sorted(tar.keys()) == ["best_score", "epoch", "model", "optimizer"]
    assert len(tar["model"]) == 204
    # resume into a fresh model + optimizer
    cfg2 = solver_config(tmp_path)
    cfg2.solver.resume = str(solver.root_dir)
    model2 = distrib.get_model(cfg2.model)
    opt2 = distrib.get_optimizer(cfg2.optim, model2)
    s2 = Solver(cfg2, model2, opt2, distrib.get_loss_function(cfg2.optim), train_dataloader=batches,
                validation_dataloader=[batches[0]], device="gpu", writer=ScalarLog())
    assert torch.equal(s2.model.flat_params.cpu(), solver.model.flat_params.cpu())
    assert opt2._step == 2 and torch.equal(opt2._m.cpu(), opt._m.cpu())


def test_eval_forward_and_cpu_rejection():
    from sehip.model import DCCRN
    from sehip import SehipError
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = DCCRN(**KW).to(dev)
    noisy, _ = make_batch(5, 2, 4000)
    model.train()
    with torch.no_grad():
        model(noisy.to(dev))  # updates running statistics
    model.eval()
    with torch.no_grad():
        out = model(noisy.to(dev))
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith(("stft.", "istft."))}
    ref = O.dccrn_forward(p, noisy, O.DCCRNConfig(**KW), training=False)
    assert rel_err(out.cpu(), ref) < 3e-2
    with pytest.raises(SehipError):
        model(noisy)  # CPU tensor: no fallback


def test_graph_replay_matches_eager_steps(tmp_path, monkeypatch):
    """Three optimisation steps through the captured hipGraphs == three eager steps (same kernels, same order)."""
    # (bit-equal first losses need a bit-reproducible forward pass: BatchNorm sums from the separate passes, not from the fp32
    #  atomics of the convolution epilogues)
    monkeypatch.setenv("SEHIP_NO_FUSE_STATS", "1")
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    batches = [make_batch(300 + s, 2, 4000) for s in range(3)]
    finals = []
    for use_graph in (False, True):
        cfg = solver_config(tmp_path)
        torch.manual_seed(cfg.seed)
        model = distrib.get_model(cfg.model)
        opt = distrib.get_optimizer(cfg.optim, model)
        solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), device="gpu", writer=ScalarLog())
        losses = []
        for noisy, clean in batches:
            mix, src = solver._prepare_batch(noisy, clean)
            fn = solver.train_step_graphed if use_graph else solver.train_step
            loss, metric = fn(mix, src)
            losses.append((float(loss), float(metric[0])))
        finals.append((losses, solver.model.flat_params.cpu().clone(), opt.sync_step()))
    (l0, p0, s0), (l1, p1, s1) = finals
    assert s0 == s1 == 3
    # step 0 starts from identical weights: forward + backward are the same kernels -> same loss / metric
    assert abs(l0[0][0] - l1[0][0]) < 1e-5 and abs(l0[0][1] - l1[0][1]) < 1e-3 * abs(l0[0][1])
    # later steps: the fp32 atomics of the weight-gradient kernels make gradients differ in the last bits from run to run
    # (eager-vs-eager too); Adam's first steps turn that into +-lr moves of the parameters whose true gradient is zero
    # (conv biases in front of a BatchNorm), hence the loose bounds
    for a, b in zip(l0[1:], l1[1:]):
        assert abs(a[0] - b[0]) < 0.03 and abs(a[1] - b[1]) < 1e-2 * max(1.0, abs(a[1]))
    assert max_abs(p0, p1) < 3 * 2 * 3e-4


def test_bench_two_ranks_on_one_gpu():
    """The N>1 path of bench.py / Solver (rank-0 broadcast, flat-gradient all-reduce between the two hipGraphs, max-over-
    ranks timing) with two processes sharing cuda:0 over gloo (RCCL refuses two ranks on one device)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SEHIP_DIST_BACKEND="gloo", SEHIP_LOCAL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "4", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["value"] > 0
    assert out["final_loss"] == out["final_loss"]  # not NaN


def test_deferred_logging_under_graph_replay(tmp_path):
    """use_graph + log_interval 2: every deferred Train/Loss_step entry is its own step's loss (the graphed step returns
    static output tensors that the next replay overwrites), i.e. the same sequence an eager, per-step-logged run gives."""
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    batches = []
    for s in range(4):
        noisy, clean = make_batch(400 + s, 2, 4000)
        batches.append((noisy, clean, [None], [None], ["x"], [s]))
    logged = []
    for use_graph, interval in ((False, 1), (True, 2)):
        cfg = solver_config(tmp_path)
        cfg.solver.use_graph, cfg.solver.log_interval = use_graph, interval
        cfg.solver.save_checkpoint_interval = 1000
        torch.manual_seed(cfg.seed)
        model = distrib.get_model(cfg.model)
        opt = distrib.get_optimizer(cfg.optim, model)
        log = ScalarLog()
        solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), train_dataloader=batches,
                        validation_dataloader=[batches[0]], device="gpu", writer=log)
        solver._run_one_epoch(0, 1, train=True)
        logged.append(([v for (t, v, _s) in log.scalars if t == "Train/Loss_step"], solver.score["loss"]))
    (eager, mean_e), (graph, mean_g) = logged
    assert len(eager) == len(graph) == 4
    assert len(set(round(v, 4) for v in graph)) == 4, graph            # four different batches -> four different losses
    for a, b in zip(eager, graph):
        assert abs(a - b) < 0.05, (eager, graph)
    assert abs(mean_e - mean_g) < 0.05


def test_device_prefetcher_yields_the_loader_in_order_and_stops_at_the_step_cap():
    """sehip.solver.DevicePrefetcher (the epoch loop's one-batch-ahead host -> HBM staging): same batches, same order, tensors on the
    device, non-tensors untouched, and with a step cap it never pulls a batch the epoch will not use."""
    from sehip.solver import DevicePrefetcher
    dev = torch.device("cuda:0")
    pulled = []

    def loader():
        for s in range(6):
            pulled.append(s)
            noisy, clean = make_batch(500 + s, 2, 4000)
            yield (noisy.pin_memory(), clean, [None], ["name"], s)

    ref = [make_batch(500 + s, 2, 4000) for s in range(6)]
    got = list(DevicePrefetcher(loader(), dev, limit=4))
    assert len(got) == 4 and pulled == [0, 1, 2, 3]
    for s, batch in enumerate(got):
        assert batch[0].is_cuda and batch[1].is_cuda and batch[2] == [None] and batch[3] == ["name"] and batch[4] == s
        assert torch.equal(batch[0].cpu(), ref[s][0]) and torch.equal(batch[1].cpu(), ref[s][1])
    pulled.clear()
    assert len(list(DevicePrefetcher(loader(), dev))) == 6 and pulled == list(range(6))


def test_workspace_generation_and_lru():
    from sehip.model import DCCRN
    from sehip import SehipError
    from sehip.loss import loss_sisdr
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    model = DCCRN(**KW).to(dev).train()
    noisy, clean = make_batch(7, 2, 4000)
    a = model(noisy.to(dev))
    b = model(0.5 * noisy.to(dev))        # same shape: overwrites the activations `a` would need
    with pytest.raises(SehipError):
        loss_sisdr(a, clean[:, 0].to(dev)).backward()
    loss_sisdr(b, clean[:, 0].to(dev)).backward()    # the live one is fine
    assert float(model.flat_grads.norm()) > 0
    # ragged validation lengths do not grow the cache without bound
    model.eval()
    with torch.no_grad():
        for n in (3000, 3100, 3200, 3300, 3400, 3500):
            model(torch.zeros(1, 1, n, device=dev))
    assert len(model._ws) <= model._ws_cap


def test_two_ranks_equal_one_rank_with_per_replica_batchnorm(tmp_path, monkeypatch):
    """Data parallel = DataParallel's semantics (src/solver.py:144-145): two ranks with 2 clips each give the parameters of ONE
    process that runs the two halves separately (per-replica BatchNorm statistics), averages the gradients, clips and steps.
    Exercises the overlapped two-range all-reduce (decoder / LSTM range first) and the 1/world folded into the optimizer.
    Two processes share cuda:0 over gloo (RCCL refuses two ranks on one device)."""
    import os, socket, subprocess, sys
    from sehip import distrib
    from sehip.solver import Solver, ScalarLog
    # the comparison is between processes: BatchNorm sums from the separate passes (the convolution epilogues' fp32 atomics differ
    # in the last bit from run to run, and the network amplifies that beyond the bound below) -- here and in the workers
    monkeypatch.setenv("SEHIP_NO_FUSE_STATS", "1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.pt")
    env = dict(os.environ, SEHIP_DIST_BACKEND="gloo", SEHIP_LOCAL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dp_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = torch.load(out)
    # the same thing in one process
    cfg = solver_config(tmp_path)
    torch.manual_seed(cfg.seed)
    model = distrib.get_model(cfg.model)
    opt = distrib.get_optimizer(cfg.optim, model)
    solver = Solver(cfg, model, opt, distrib.get_loss_function(cfg.optim), device="gpu", writer=ScalarLog())
    noisy, clean = make_batch(900, 4, 4000)
    model.train()
    opt.zero_grad()
    bsave, nsave = model._bflat.clone(), model._nbt.clone()
    for r_ in range(2):
        mix, src = solver._prepare_batch(noisy[2 * r_:2 * r_ + 2], clean[2 * r_:2 * r_ + 2])
        solver.loss_function(model(mix), src).backward()          # second call accumulates into flat_grads
        if r_ == 0:
            model._bflat.copy_(bsave); model._nbt.copy_(nsave)    # rank 1 starts from the same running statistics
    opt.grad_scale = 0.5
    opt.clip_grad_norm_(cfg.optim.clip_grad)
    opt.step()
    torch.cuda.synchronize()
    ref_g, ref_p = model.flat_grads.cpu(), model.flat_params.cpu()
    assert rel_err(got["grads"], ref_g) < 2e-3                     # fp32 atomics of the weight gradients: run-to-run noise only
    assert max_abs(got["params"], ref_p) < 2.1 * 3e-4 and float((got["params"] - ref_p).abs().mean()) < 0.05 * 3e-4


def test_rccl_entry_points_of_the_c_abi_single_rank():
    """sehip_comm_unique_id / sehip_comm_init / sehip_allreduce_f32 / sehip_comm_destroy (include/sehip.h): a one-rank communicator on
    this GPU, an in-place SUM of a range of a flat buffer on a side stream (identity at world 1), and sehip.distrib.DirectComm
    around them.  (RCCL refuses two ranks on ONE device, so the 2-rank path of this backend cannot run on a 1-GPU box; the
    torch.distributed path is what the 2-rank tests above exercise.)"""
    from sehip import distrib
    c = distrib.DirectComm(0, 1, torch.device("cuda:0"))
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    c.all_reduce_(x, 100, 900, side).wait()
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    c.close()
