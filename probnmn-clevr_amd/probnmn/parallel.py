"""Data parallelism: one process per GPU, full replica, gradient all-reduce over RCCL/xGMI.

The reference wraps models in ``nn.DataParallel`` (single process, parameter broadcast and
gradient reduce through device 0 every step -- reference: probnmn/trainers/_trainer.py:94-100).
Here every rank owns a replica and a shard of the batch; after backward the gradient *arenas*
(one contiguous buffer per model, ``probnmn.runtime.arena``) are summed with a single
``all_reduce`` each -- for the NMN that is one 257 MB collective instead of ~110 small ones,
which is what a point-to-point xGMI fabric wants (per-link bound; few, large messages) -- then
scaled by 1/world so that equal-sized shards reproduce the single-process mean loss gradient.
The element-wise clamp happens AFTER the reduce, as in the reference, where the clamp sees the
whole-batch gradient (joint_training_trainer.py:181-188).
"""
from typing import Iterable, Sequence

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_gradients(arenas: Sequence, loose_params: Iterable[torch.nn.Parameter] = (), average: bool = True) -> None:
    n = world()
    if n == 1:
        return
    scale = 1.0 / n if average else 1.0
    handles = []
    for a in arenas:
        handles.append((dist.all_reduce(a.grad, op=dist.ReduceOp.SUM, async_op=True), a.grad))
    for p in loose_params:
        if p.grad is not None:
            handles.append((dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True), p.grad))
    for h, g in handles:
        h.wait()
        if scale != 1.0:
            g.mul_(scale)


def all_reduce_scalars(values: torch.Tensor) -> torch.Tensor:
    """Sum a small tensor of per-rank partial sums / counts (REINFORCE baseline, metrics)."""
    if world() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values


def broadcast_parameters(arenas: Sequence, loose_params: Iterable[torch.nn.Parameter] = (), src: int = 0) -> None:
    if world() == 1:
        return
    for a in arenas:
        dist.broadcast(a.flat, src)
    for p in loose_params:
        dist.broadcast(p.data, src)
