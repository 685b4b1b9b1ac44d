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
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def serial_collectives() -> bool:
    """``PNMN_DP_SERIAL_COLLECTIVES=1``: no collective is started during backward -- no early reducer, no cut in the
    trunk's backward launch list; ``all_reduce_gradients`` issues all of them behind the joined streams, in the same
    fixed order, so a collective never shares the chip with a recurrent kernel.  The fallback ``probnmn.launch_guard``
    restarts a data-parallel job with after a hang (same gradients bit for bit: the sums do not depend on when they
    start).  Must be set identically on every rank."""
    import os

    return os.environ.get("PNMN_DP_SERIAL_COLLECTIVES", "0") == "1"


_DP_SAFE_DONE = False


def dp_safe() -> None:
    """Once per data-parallel process: keep RCCL off the CUs the multi-CU recurrent kernels are sized for.  Those
    kernels need ALL their workgroups resident at once and spin for each other; RCCL's workgroups stay resident
    for a whole collective and wait for their peers; the early all-reduces put the two on the chip together
    (a grid that needs every CU beside a kernel that itself waits is what stalled round 1).  With
    ``PNMN_DP_RESERVE_CUS`` CUs (default 32; 0 = off, the recurrent grids take the whole chip as in a single
    process) left out of those grids, both always fit.  The library then runs a batch whose row tiles no
    longer fit one launch as several launches (pnmn_cluster_reserve_cus)."""
    global _DP_SAFE_DONE
    if _DP_SAFE_DONE or world() == 1:
        return
    _DP_SAFE_DONE = True
    import os

    reserve = int(os.environ.get("PNMN_DP_RESERVE_CUS", "32"))
    if reserve > 0 and torch.cuda.is_available():
        from probnmn import _hip

        _hip.lib().pnmn_cluster_reserve_cus(reserve)


_HOOK_OWNER = weakref.WeakValueDictionary()  # id(parameter) -> the EarlyReducer whose hook is live on it


class EarlyReducer:
    """Starts all-reduces DURING backward, so that they overlap the rest of it.  Two kinds of slot, issued in
    one fixed order (all parameters in registration order, then all arena pieces in registration order):

    * chosen (large, loose) parameters, from their post-accumulate-grad hook.  For the NMN that is
      ``classifier.4.weight``: 205 MB of the 257 MB gradient payload, final right after the classifier's
      backward -- before the whole module-program / stem backward runs;
    * contiguous ranges of a gradient arena, announced by the engine (``piece_ready``) the moment the last
      kernel that writes them has been queued: the module convs' weight gradients (48 of the trunk arena's
      52 MB) leave while the stem's backward and the seq2seq backward still run.

    Autograd may finish things in a different order on different ranks (their graphs differ when a loss term
    has no rows in a shard), so a slot's collective is started only once every slot before it has been
    started; what never became ready on this rank (no local rows) is started by ``all_reduce_gradients``,
    in the same order.  The sequence of collectives is therefore the same on every rank.

    One live hook per parameter: a reducer that registers on a parameter another reducer already hooks
    (a second trainer over the same NMN -- the reference's phase pipeline, bench.py) takes the parameter
    over, so a gradient is never all-reduced twice."""

    def __init__(self, params: Iterable[torch.nn.Parameter], pieces: Sequence = ()):
        self._pending = {}
        self._hooks = {}
        self._ready = set()
        self._next = 0  # collectives are ISSUED in slot order, whatever order backward finishes in
        self.armed = False
        self.params = list(params)
        #: (arena, first float, one past the last float): registered once, same on every rank
        self.pieces = [(a, int(lo), int(hi)) for a, lo, hi in pieces]
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
        """The trainer that owns this reducer is about to run backward: its slots may start collectives.  A hook
        that fires while its reducer is NOT armed (another trainer over the same parameter stepping: its
        ``all_reduce_gradients`` reduces that gradient itself) does nothing -- otherwise the gradient would be
        summed twice and an orphan collective would be left in ``_pending``."""
        self.armed = True

    # ---- slots ---------------------------------------------------------------------------------
    def _slots(self):
        return [("p", p) for p in self.params] + [("a", k) for k in range(len(self.pieces))]

    def _key(self, slot):
        return id(slot[1]) if slot[0] == "p" else ("piece", slot[1])

    def _tensor(self, slot):
        if slot[0] == "p":
            p = slot[1]
            if p.grad is None:  # (its loss term had no rows in this shard: it contributes zeros)
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
            return p.grad
        a, lo, hi = self.pieces[slot[1]]
        for e in getattr(self, "engines", ()):  # (an engine that rebuilt its arena -- .to(), re-pointed parameters --
            # leaves this reducer holding the old one: reducing it would silently skip the live gradients)
            if e.arena is not None and e.arena is not a and any(a is q[0] for q in self.pieces):
                if not any(e.arena is q[0] for q in self.pieces):
                    raise RuntimeError("the NMN engine rebuilt its parameter arena after this reducer was created: "
                                       "build the trainer (early_reducer_for) after moving the model")
        return a.grad[lo:hi]

    def _start(self, slot) -> None:
        g = self._tensor(slot)
        self._pending[self._key(slot)] = (dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True), g)

    def _advance(self) -> None:
        slots = self._slots()
        while self._next < len(slots) and self._key(slots[self._next]) in self._ready:
            self._start(slots[self._next])
            self._next += 1

    def _fire(self, p: torch.nn.Parameter) -> None:
        if world() == 1 or p.grad is None or not self.armed:
            return
        self._ready.add(id(p))
        self._advance()

    def piece_ready(self, k: int) -> None:
        """Every kernel that writes arena piece ``k`` has been queued on the current stream (the collective is
        ordered behind them: the process group waits for the current stream at the call)."""
        if world() == 1 or not self.armed:
            return
        self._ready.add(("piece", k))
        self._advance()

    def finish(self):
        """End of backward: the handles of every slot, in slot order; what has not been started yet is started
        now.  Disarms the reducer."""
        handles = []
        for slot in self._slots():
            if self._key(slot) not in self._pending:
                self._start(slot)
            handles.append(self._pending.pop(self._key(slot)))
        self.reset()
        self.armed = False
        return handles

    def covers(self, arena) -> List:
        """The float ranges of ``arena`` this reducer's pieces cover, sorted."""
        return sorted((lo, hi) for a, lo, hi in self.pieces if a is arena)

    def take(self, p: torch.nn.Parameter):
        return self._pending.pop(id(p), None)

    def reset(self) -> None:
        self._ready.clear()
        self._next = 0

    def remove(self) -> None:
        for p in list(self.params):
            self._drop(p)
        self.pieces = []


def early_reducer_for(big_params, engines) -> "EarlyReducer":
    """What the trainers build: the large loose parameters from their gradient hooks plus -- in a data-parallel
    run -- the gradient pieces of every NMN engine's trunk arena, announced by the engine's backward.  A
    single process gets the hooks only (they return at once) and no cut in the engine's launch lists."""
    pieces = []
    dp_safe()
    if world() > 1 and serial_collectives():
        return None
    if world() > 1:
        for e in engines:
            pieces.extend(e.grad_pieces())
    if not big_params and not pieces:
        return None
    r = EarlyReducer(big_params, pieces)
    r.engines = list(engines) if pieces else []
    at = 0
    for e in r.engines:
        n = len(e.grad_pieces())
        e.on_grad_piece = (lambda k, base=at: r.piece_ready(base + k))
        e._grad_piece_owner = r
        at += n
    return r


SMALL_BUCKET_BYTES = 8 << 20

#: bench.py: when a list, every all_reduce_gradients appends (event before the first wait, event after the last
#: scale) recorded on the current stream -- the time the step's stream is held up by the collectives
TIMING = None


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
    dp_safe()
    scale = 1.0 / n if average else 1.0
    handles = []
    loose_params = list(loose_params)
    for p in loose_params:
        if p.grad is None:
            p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    first = set()
    covered = {}
    if early is not None:
        first = {id(p) for p in early.params}
        for a in arenas:
            covered[id(a)] = early.covers(a)
        handles.extend(early.finish())
    for a in arenas:
        # whatever of the arena no early piece covers (all of it when there are none)
        at = 0
        for lo, hi in covered.get(id(a), []) + [(a.grad.numel(), a.grad.numel())]:
            if lo > at:
                g = a.grad[at:lo]
                handles.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True), g))
            at = max(at, hi)
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
    timing = TIMING if (TIMING is not None and handles and handles[0][1].is_cuda) else None
    if timing is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    for h, g in handles:
        h.wait()
        if scale != 1.0:
            g.mul_(scale)
    if timing is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        timing.append((e0, e1))
    if bucket is not None:
        torch._foreach_copy_(small, [c.view_as(g) for c, g in zip(bucket.split([g.numel() for g in small]), small)])


_HOST_GROUP = None


def host_or(mask):
    """Element-wise OR of a small host-side bool array over the ranks (numpy in, numpy out) -- which parameters received a
    gradient on ANY rank: ``ClampAdam`` starts a parameter's Adam state at its first gradient, and the replicas must start it
    in the same iteration (a module one shard samples and the other does not would otherwise drift apart).  Travels over a
    gloo group of its own (created on first use: every rank must reach that call -- the optimiser step -- together): the
    result decides host-side step counts, and reading it back from the device would cost a stream synchronisation per step."""
    import numpy as np

    if world() == 1:
        return mask
    global _HOST_GROUP
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    t = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8).copy())
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_HOST_GROUP)
    return t.numpy().astype(bool)


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
    dp_safe()
    for a in arenas:
        dist.broadcast(a.flat, src)
    for p in loose_params:
        dist.broadcast(p.data, src)
    # written through .data / the arena: no version counter moved, so tell the caches of derived parameters
    # (fragment-packed recurrent weights, bias sums) that the values changed
    from probnmn.optim import parameters_changed

    parameters_changed()
