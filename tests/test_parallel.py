"""Data-parallel reducer with world_size 2 over gloo on CPU (no GPU): summing the two ranks'
gradient arenas and scaling by 1/2 reproduces the single-process gradient of the mean loss over
the concatenated batch; parameters broadcast from rank 0; scalar partial sums add up."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeArena:
    """Stands in for runtime.arena.ParamArena: the reducer only touches .flat and .grad."""

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.flat = torch.randn(n, generator=g)
        self.grad = torch.zeros(n)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "probnmn-clevr_amd"))
    from probnmn import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    # a tiny "model": y = x @ w.T + b, loss = per-example squared error, mean over the shard
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 5, generator=g)
    Y = torch.randn(8, 3, generator=g)
    arena = _FakeArena(15, seed=rank)  # different initial weights per rank on purpose
    loose = torch.nn.Parameter(torch.randn(3, generator=torch.Generator().manual_seed(10 + rank)))
    # two more loose tensors that take the bucketed path (one collective for all small gradients)
    scale = torch.nn.Parameter(torch.randn(3, generator=torch.Generator().manual_seed(20 + rank)))
    shift = torch.nn.Parameter(torch.randn(1, generator=torch.Generator().manual_seed(30 + rank)))
    parallel.broadcast_parameters([arena], [loose, scale, shift])
    w = arena.flat.view(3, 5).clone().requires_grad_(True)
    early = parallel.EarlyReducer([])          # no parameters registered: everything reduced at the end
    # the loose parameter's all-reduce starts inside backward; the arena travels as two pieces (engine.grad_pieces:
    # floats 5..15 are final first, 0..5 last) plus whatever no piece covers (nothing here)
    early2 = parallel.EarlyReducer([loose], pieces=[(arena, 5, 15), (arena, 0, 5)])
    shard = slice(rank * 4, rank * 4 + 4)
    loss = (((X[shard] @ w.T + loose) * scale + shift - Y[shard]) ** 2).sum(1).mean()
    early2.arm()  # (what a trainer does right before its backward)
    loss.backward()
    arena.grad.copy_(w.grad.reshape(-1))
    assert id(loose) in early2._pending  # the hook fired during backward; taken by all_reduce_gradients below
    if rank == 0:
        # only THIS rank announces piece 0 during "backward" (the other's shard had no rows for it): rank 1 issues
        # the same collective from all_reduce_gradients, in the same slot order
        early2.piece_ready(0)
        assert ("piece", 0) in early2._pending and ("piece", 1) not in early2._pending
    else:
        early2.piece_ready(1)  # out of order: held back until the slots before it have been started
        assert ("piece", 1) not in early2._pending
    parallel.all_reduce_gradients([arena], [loose, scale, shift], early=early2)
    sums = parallel.all_reduce_scalars(torch.tensor([float(rank + 1), 4.0]))
    # loss terms are means over data-dependent subsets: shard sizes 3 and 5 here.  The trainers weight
    # each local mean by n_local * world / n_global so that the AVERAGED gradient is the global mean's.
    from probnmn.modules.elbo import Reinforce
    from probnmn.trainers.joint_training import _dp_weight

    values = torch.arange(8.0)
    mine = values[:3] if rank == 0 else values[3:]
    weighted = _dp_weight(mine.numel(), torch.device("cpu")) * mine.mean()
    dist.all_reduce(weighted)
    global_mean = weighted / world
    reinforce = Reinforce(baseline_decay=0.5)
    reinforce(torch.ones_like(mine), mine)  # the baseline moves by decay * GLOBAL mean of (reward - baseline)
    if rank == 0:
        torch.save({"w": arena.flat.clone(), "gw": arena.grad.clone(), "gb": loose.grad.clone(),
                    "b": loose.detach().clone(), "sums": sums, "scale": scale.detach().clone(),
                    "shift": shift.detach().clone(), "gscale": scale.grad.clone(), "gshift": shift.grad.clone(),
                    "global_mean": float(global_mean), "baseline": float(reinforce._reinforce_baseline)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # single-process reference on the full batch with rank 0's (broadcast) parameters
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 5, generator=g)
    Y = torch.randn(8, 3, generator=g)
    w = got["w"].view(3, 5).clone().requires_grad_(True)
    b = got["b"].clone().requires_grad_(True)
    scale, shift = got["scale"].clone().requires_grad_(True), got["shift"].clone().requires_grad_(True)
    loss = (((X @ w.T + b) * scale + shift - Y) ** 2).sum(1).mean()
    loss.backward()
    torch.testing.assert_close(got["gscale"], scale.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["gshift"], shift.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["gw"], w.grad.reshape(-1), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["gb"], b.grad, rtol=1e-6, atol=1e-6)
    assert got["sums"].tolist() == [3.0, 8.0]
    assert abs(got["global_mean"] - 3.5) < 1e-6  # mean of 0..7, from shards of 3 and 5
    assert abs(got["baseline"] - 0.5 * 3.5) < 1e-6
    # broadcast really took rank 0's values
    torch.testing.assert_close(got["w"], _FakeArena(15, seed=0).flat)


def test_single_process_is_a_noop():
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "probnmn-clevr_amd"))
    from probnmn import parallel

    a = _FakeArena(4, 0)
    a.grad.fill_(2.0)
    parallel.all_reduce_gradients([a])
    assert parallel.world() == 1 and a.grad.tolist() == [2.0] * 4


def _or_worker(rank, world, port, out_path):
    import sys

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "probnmn-clevr_amd"))
    from probnmn import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mask = np.zeros(7, bool)
    mask[[0, 2] if rank == 0 else [2, 5]] = True
    got = parallel.host_or(mask)
    again = parallel.host_or(np.zeros(7, bool))
    torch.save((got.tolist(), again.tolist(), mask.tolist()), out_path + str(rank))
    dist.destroy_process_group()


def test_host_or_over_ranks(tmp_path):
    """``parallel.host_or``: which trunk parameters received a gradient on ANY rank (ClampAdam's per-parameter step counts)."""
    out = str(tmp_path / "o")
    mp.spawn(_or_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        got, again, mine = torch.load(out + str(r))
        assert got == [True, False, True, False, False, True, False]
        assert again == [False] * 7
        assert mine == ([True, False, True, False, False, False, False] if r == 0 else [False, False, True, False, False, True, False])
