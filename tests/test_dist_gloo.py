"""CPU, world_size 2 over gloo: the data-parallel helpers (rank-0 broadcast of the flat parameters, ONE all-reduce of
the flat gradient buffer followed by 1/world) reproduce the global-batch mean gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sehip import distrib
    r, w, l = distrib.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)
    params = torch.randn(1000)
    distrib.broadcast_parameters(params)
    grads = torch.full((1000,), float(rank + 1)) + torch.arange(1000) * 1e-3
    distrib.allreduce_gradients(grads)
    # a rank-local verdict every rank must act on together (model.check_health(): a hand-off time-out seen on ONE rank switches the
    # launch path of ALL of them, or their collective sequences diverge -- ADVICE r4): the OR over the ranks
    flags = [distrib.global_flag(rank == 1), distrib.global_flag(False), distrib.global_flag(True)]
    guard = torch.tensor([7 if rank == 1 else 0], dtype=torch.int32)
    distrib.allreduce_step_guard(guard)                             # the device step guard of the optimizer: MAX over the ranks
    torch.save({"params": params, "grads": grads, "flags": flags, "guard": int(guard[0])}, f"{out}/r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mean_and_broadcast(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["params"], b["params"])                    # rank 0's weights everywhere
    want = 1.5 + torch.arange(1000) * 1e-3                          # mean of the two ranks' gradients
    assert torch.allclose(a["grads"], want) and torch.equal(a["grads"], b["grads"])
    assert a["flags"] == b["flags"] == [True, False, True]
    assert a["guard"] == b["guard"] == 7


def test_global_flag_without_a_process_group_is_the_local_value():
    from sehip import distrib
    assert distrib.global_flag(True) is True and distrib.global_flag(False) is False
