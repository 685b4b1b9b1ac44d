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


class EarlyReducer:
    """Starts the all-reduce of chosen (large, loose) parameters the moment autograd has finished
    their gradient, so that the collective overlaps the rest of backward.  For the NMN that is
    ``classifier.4.weight``: 205 MB of the 257 MB gradient payload, final right after the
    classifier's backward -- before the whole module-program / stem backward runs."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self._pending = {}
        self._hooks = []
        for p in params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._fire))

    def _fire(self, p: torch.nn.Parameter) -> None:
        if world() > 1 and p.grad is not None:
            self._pending[id(p)] = (dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True), p.grad)

    def take(self, p: torch.nn.Parameter):
        return self._pending.pop(id(p), None)

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


SMALL_BUCKET_BYTES = 8 << 20


def all_reduce_gradients(arenas: Sequence, loose_params: Iterable[torch.nn.Parameter] = (), average: bool = True,
                         early: "EarlyReducer" = None) -> None:
    n = world()
    if n == 1:
        return
    scale = 1.0 / n if average else 1.0
    handles = []
    for a in arenas:
        handles.append((dist.all_reduce(a.grad, op=dist.ReduceOp.SUM, async_op=True), a.grad))
    small = []  # the seq2seq models have ~40 tensors of a few hundred KB: one bucket, one collective
    for p in loose_params:
        if p.grad is None:
            continue
        started = early.take(p) if early is not None else None
        if started is not None:
            handles.append(started)
        elif p.grad.numel() * p.grad.element_size() < SMALL_BUCKET_BYTES and p.grad.is_contiguous():
            small.append(p.grad)
        else:
            handles.append((dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True), p.grad))
    bucket = None
    if small:
        bucket = torch.cat([g.reshape(-1) for g in small])
        handles.append((dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True), bucket))
    for h, g in handles:
        h.wait()
        if scale != 1.0:
            g.mul_(scale)
    if bucket is not None:
        torch._foreach_copy_(small, [c.view_as(g) for c, g in zip(bucket.split([g.numel() for g in small]), small)])


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
