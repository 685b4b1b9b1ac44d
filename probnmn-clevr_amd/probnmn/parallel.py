"""Data parallelism: one process per GPU, full replica, gradient all-reduce over RCCL/xGMI.

The reference wraps models in ``nn.DataParallel`` (single process, parameter broadcast and
gradient reduce through device 0 every step -- reference: probnmn/trainers/_trainer.py:94-100).
Here every rank owns a replica and a shard of the batch; after backward the gradient *arenas*
(one contiguous buffer per model, ``probnmn.runtime.arena``) are summed with a single
``all_reduce`` each -- for the NMN trunk that is one 52 MB collective instead of ~110 small ones (the
205 MB gradient of the fully connected layer is a loose tensor, reduced early from its gradient hook:
``EarlyReducer``), which is what a point-to-point xGMI fabric wants (per-link bound; few, large
messages) -- then scaled by 1/world so that equal-sized shards reproduce the single-process mean loss
gradient.
The element-wise clamp happens AFTER the reduce, as in the reference, where the clamp sees the
whole-batch gradient (joint_training_trainer.py:181-188).
"""
import weakref
from typing import Iterable, Sequence

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


_HOOK_OWNER = weakref.WeakValueDictionary()  # id(parameter) -> the EarlyReducer whose hook is live on it


class EarlyReducer:
    """Starts the all-reduce of chosen (large, loose) parameters the moment autograd has finished
    their gradient, so that the collective overlaps the rest of backward.  For the NMN that is
    ``classifier.4.weight``: 205 MB of the 257 MB gradient payload, final right after the
    classifier's backward -- before the whole module-program / stem backward runs.

    One live hook per parameter: a reducer that registers on a parameter another reducer already hooks
    (a second trainer over the same NMN -- the reference's phase pipeline, bench.py) takes the parameter
    over, so a gradient is never all-reduced twice."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self._pending = {}
        self._hooks = {}
        self._ready = set()
        self._next = 0  # collectives are ISSUED in registration order, whatever order autograd finishes in
        self.params = list(params)
        for p in self.params:
            old = _HOOK_OWNER.get(id(p))
            if old is not None and old is not self:
                old._drop(p)
            self._hooks[id(p)] = p.register_post_accumulate_grad_hook(self._fire)
            _HOOK_OWNER[id(p)] = self

    def _drop(self, p: torch.nn.Parameter) -> None:
        h = self._hooks.pop(id(p), None)
        if h is not None:
            h.remove()
        self._pending.pop(id(p), None)
        self.params = [q for q in self.params if q is not p]
        self.reset()

    def arm(self) -> None:
        """The trainer that owns this reducer is about to run backward: its hooks may start collectives.  A hook
        that fires while its reducer is NOT armed (another trainer over the same parameter stepping: its
        ``all_reduce_gradients`` reduces that gradient itself) does nothing -- otherwise the gradient would be
        summed twice and an orphan collective would be left in ``_pending``."""
        self.armed = True

    def _fire(self, p: torch.nn.Parameter) -> None:
        if world() == 1 or p.grad is None or not getattr(self, "armed", False):
            return
        # autograd may finish the hooked gradients in a different order on different ranks (their graphs
        # differ when a loss term has no rows in a shard): a collective is started only once every
        # parameter registered before it has been started, so the order is the same everywhere
        self._ready.add(id(p))
        while self._next < len(self.params) and id(self.params[self._next]) in self._ready:
            q = self.params[self._next]
            self._pending[id(q)] = (dist.all_reduce(q.grad, op=dist.ReduceOp.SUM, async_op=True), q.grad)
            self._next += 1

    def take(self, p: torch.nn.Parameter):
        return self._pending.pop(id(p), None)

    def reset(self) -> None:
        self._ready.clear()
        self._next = 0

    def remove(self) -> None:
        for p in list(self.params):
            self._drop(p)


SMALL_BUCKET_BYTES = 8 << 20


def all_reduce_gradients(arenas: Sequence, loose_params: Iterable[torch.nn.Parameter] = (), average: bool = True,
                         early: "EarlyReducer" = None) -> None:
    """Sum (and average) every gradient over the ranks.  The SEQUENCE of collectives is the same on every
    rank whatever happened locally: a parameter that received no gradient on this rank (its loss term had
    no rows in this shard) contributes zeros, and the parameters of ``early`` always come first, in
    registration order -- where their hook already fired during backward that collective IS the first,
    where it did not (no local rows) it is issued here, before anything else."""
    n = world()
    if n == 1:
        return
    scale = 1.0 / n if average else 1.0
    handles = []
    loose_params = list(loose_params)
    for p in loose_params:
        if p.grad is None:
            p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    first = set()
    if early is not None:
        for p in early.params:
            first.add(id(p))
            started = early.take(p)
            if started is None:
                if p.grad is None:
                    p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
                started = (dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True), p.grad)
            handles.append(started)
        early.reset()
        early.armed = False
    for a in arenas:
        handles.append((dist.all_reduce(a.grad, op=dist.ReduceOp.SUM, async_op=True), a.grad))
    small = []  # the seq2seq models have ~40 tensors of a few hundred KB: one bucket, one collective
    for p in loose_params:
        if id(p) in first:
            continue
        if p.grad.numel() * p.grad.element_size() < SMALL_BUCKET_BYTES and p.grad.is_contiguous():
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


def mean_weight(n_local: int, device):
    """n_local * world / n_global: the factor that turns a rank's local MEAN over n_local rows into its share
    of the global mean once gradients are averaged over ranks (shards need not be equal; subsets may be
    empty).  1.0 in a single process, otherwise a 0-dim device tensor -- the count is summed with a
    collective, reading it back would cost a host sync per loss term.  Every rank must call it the same
    number of times in the same order."""
    if world() == 1:
        return 1.0
    local = torch.full((), float(n_local), device=device)
    total = all_reduce_scalars(local.clone())
    return torch.where(total > 0, local * world() / total.clamp(min=1.0), torch.zeros_like(total))


def broadcast_parameters(arenas: Sequence, loose_params: Iterable[torch.nn.Parameter] = (), src: int = 0) -> None:
    if world() == 1:
        return
    for a in arenas:
        dist.broadcast(a.flat, src)
    for p in loose_params:
        dist.broadcast(p.data, src)
    # written through .data / the arena: no version counter moved, so tell the caches of derived parameters
    # (fragment-packed recurrent weights, bias sums) that the values changed
    from probnmn.optim import parameters_changed

    parameters_changed()
