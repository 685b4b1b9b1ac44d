"""Fused gradient clamp + Adam over parameter arenas, as a ``torch.optim.Optimizer``.

The reference clamps every gradient element to [-5, 5] in a Python loop and then calls one
``torch.optim.Adam`` that spans all trainable models, driven by a ``ReduceLROnPlateau`` scheduler and
saved / restored by the checkpoint manager (reference: probnmn/trainers/module_training_trainer.py:94-96,
joint_training_trainer.py:182-188, _trainer.py:103-118,124-130,193).  On MI355X the whole update is one
streaming kernel (``pnmn_clamp_adam``): 28 bytes of HBM traffic per parameter, one launch for the
63 M-parameter trunk arena plus one item per loose tensor.  Arithmetic follows ``torch.optim.Adam``
(no amsgrad).

Optimizer surface: one parameter group holding every parameter in ``model.parameters()`` order (what
``optim.Adam(all_parameters, ...)`` builds), so ``param_groups[0]["lr"]`` is what an lr scheduler
changes and what the next fused step uses; ``state_dict()`` / ``load_state_dict()`` use
``torch.optim.Adam``'s layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter index, logical
parameter shapes), so optimizer state moves between this class and the reference's Adam in either
direction.  The moments of arena parameters are views into two arena-shaped buffers.

Per-parameter step counts, as ``torch.optim.Adam`` keeps them under the reference's torch 1.4.0 (requirements.txt:6):
a parameter's state starts at its FIRST gradient (a module no program has used yet is skipped), and from then on it is
updated in EVERY step -- torch 1.4.0's ``zero_grad`` zeroes gradients in place and never drops them, so Adam keeps
applying the momentum of a module the current batch does not use.  The engine reports which trunk parameters a
backward pass reached (``ParamArena.touched``); parameters that share a step count go out as contiguous arena ranges of
one launch (one range per arena once every module has been used: the first iteration at the reference's batch sizes).
Rounds 1-4 kept ONE counter for all parameters: a module first used at iteration k then took a first step of
0.74 lr .. 0.32 lr instead of lr (tests/test_trajectory_gpu.py found it).  Data parallel: a trunk parameter counts as
touched when ANY rank's batch reached it -- the union over the shards is what a single process on the whole batch would
see -- through one host-side OR per step (``parallel.host_or``, a few hundred bytes over gloo), so the replicas' Adam
states stay identical and equal the single-process ones (round 5 counted every parameter as touched on every rank).
"""
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from probnmn import _hip


_PARAMETER_EPOCH = 0
import os as _os
SIDE_BLOCKS_PER_ITEM = int(_os.environ.get("PNMN_ADAM_SIDE_BLOCKS", "512"))


def parameter_epoch() -> int:
    """Counts the fused optimiser steps of this process.  ``pnmn_clamp_adam`` writes the parameters through
    their pointers, which bumps no tensor version counter; caches of derived parameters
    (``probnmn.modules.seq2seq_base.DerivedParams``) key on this as well."""
    return _PARAMETER_EPOCH


def parameters_changed() -> None:
    """Tell the derived-parameter caches that parameter VALUES were changed without touching a version counter
    (a write through ``p.data`` or a raw device pointer).  ``ClampAdam.step`` calls it itself."""
    global _PARAMETER_EPOCH
    _PARAMETER_EPOCH += 1


class ClampAdam(torch.optim.Optimizer):
    def __init__(
        self,
        params: Iterable[nn.Parameter],
        arenas: Sequence = (),
        lr: float = 1e-4,
        betas=(0.9, 0.999),
        eps: float = 1e-8,
        weight_decay: float = 0.0,
        clamp: Optional[float] = 5.0,
    ):
        params = list(params)
        if any(isinstance(p, dict) for p in params):
            raise ValueError("ClampAdam takes one flat list of parameters (a single group, as the reference's Adam)")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, clamp=clamp)
        super().__init__(params, defaults)
        self.arenas = list(arenas)
        in_arena = set()
        for a in self.arenas:
            in_arena.update(id(a.param(n)) for n in a.names)
        known = {id(p) for p in params}
        for a in self.arenas:
            missing = [n for n in a.names if id(a.param(n)) not in known]
            if missing:
                raise ValueError("arena parameters missing from the parameter list: %s ..." % missing[:3])
        self.loose: List[nn.Parameter] = [p for p in params if id(p) not in in_arena and p.requires_grad]
        self.step_count = 0  # calls of step()
        self._check_full = True  # first step (and the first after load_state_dict): every parameter checked
        self._arena_state = [(torch.zeros_like(a.flat), torch.zeros_like(a.flat)) for a in self.arenas]
        self._loose_state = [(torch.zeros_like(p, memory_format=torch.contiguous_format),
                              torch.zeros_like(p, memory_format=torch.contiguous_format)) for p in self.loose]
        # Adam's per-parameter step counts (0: no gradient so far -- no state, as torch.optim.Adam)
        self._arena_steps = [np.zeros(len(a.names), np.int64) for a in self.arenas]
        self._loose_steps = np.zeros(len(self.loose), np.int64)
        self._loose_zero = {}  # zero gradients of started loose parameters a step gave none (rare)
        self._arena_bounds = [np.array([a.offsets[n] for n in a.names] + [a.total], np.int64) for a in self.arenas]
        self._loose_ptrs = [(m.data_ptr(), v.data_ptr()) for m, v in self._loose_state]
        self._loose_numel = [p.numel() for p in self.loose]
        self._bind_state()

    # ---- torch.optim surface -------------------------------------------------------------------
    def _bind_state(self) -> None:
        """self.state in torch.optim.Adam's layout, aliasing the buffers the kernel updates."""
        for a, (m, v), steps in zip(self.arenas, self._arena_state, self._arena_steps):
            for i, n in enumerate(a.names):
                p = a.param(n)
                self.state[p] = {"step": torch.tensor(float(steps[i])), "exp_avg": a.view_of(m, n),
                                 "exp_avg_sq": a.view_of(v, n)}
        for i, (p, (m, v)) in enumerate(zip(self.loose, self._loose_state)):
            self.state[p] = {"step": torch.tensor(float(self._loose_steps[i])), "exp_avg": m, "exp_avg_sq": v}

    def _sync_steps(self) -> None:
        for a, steps in zip(self.arenas, self._arena_steps):
            for i, n in enumerate(a.names):
                self.state[a.param(n)]["step"].fill_(float(steps[i]))
        for i, p in enumerate(self.loose):
            self.state[p]["step"].fill_(float(self._loose_steps[i]))

    def add_param_group(self, param_group) -> None:
        if getattr(self, "param_groups", None):
            raise ValueError("ClampAdam updates all parameters with one fused launch: a single parameter group")
        super().add_param_group(param_group)

    def state_dict(self):
        self._sync_steps()
        sd = super().state_dict()
        # (torch.optim.Adam holds no state for a parameter that never had a gradient)
        sd["state"] = {k: v for k, v in sd["state"].items() if float(v["step"]) > 0}
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)  # casts to each parameter's device / dtype, replaces self.state
        # torch REPLACES param_groups with the saved ones: a checkpoint written by the reference's torch.optim.Adam
        # has no 'clamp' entry (the reference clamps in a Python loop, _trainer.py:145-150) and carries options this
        # fused update does not implement -- restore the former, refuse the latter instead of ignoring them
        for g in self.param_groups:
            for flag in ("amsgrad", "maximize"):
                if g.get(flag):
                    raise ValueError("ClampAdam has no %s: the loaded optimizer state was written with %s=True" % (flag, flag))
            for k, v in self.defaults.items():
                g.setdefault(k, v)
        self._check_full = True  # (the next step re-verifies that every parameter still aliases its arena)
        loaded = dict(self.state)
        steps = [float(st["step"]) for st in loaded.values() if "step" in st]
        self.step_count = int(max(steps)) if steps else 0
        with torch.no_grad():
            for a, (m, v), asteps in zip(self.arenas, self._arena_state, self._arena_steps):
                m.zero_(), v.zero_()
                asteps[:] = 0
                for i, n in enumerate(a.names):
                    st = loaded.get(a.param(n))
                    if st and "step" in st:  # (no entry: the parameter had no gradient before the checkpoint)
                        a.view_of(m, n).copy_(st["exp_avg"])
                        a.view_of(v, n).copy_(st["exp_avg_sq"])
                        asteps[i] = int(float(st["step"]))
            self._loose_steps[:] = 0
            for i, (p, (m, v)) in enumerate(zip(self.loose, self._loose_state)):
                st = loaded.get(p)
                m.zero_(), v.zero_()
                if st and "step" in st:
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                    self._loose_steps[i] = int(float(st["step"]))
        self.state.clear()
        self._bind_state()

    def zero_grad(self, set_to_none: bool = True) -> None:
        # arena gradients are zeroed by the engine at the start of each backward
        for p in self.loose:
            p.grad = None

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    @torch.no_grad()
    def step(self, closure=None, side_stream=None, side_params=()) -> None:
        """``side_stream`` (with ``side_params``: ids of loose parameters): the update of every ARENA and of those loose
        parameters is launched on that stream instead of the current one -- a trainer whose NMN runs on its own stream lets
        the NMN's share of the update (the 51 M-parameter fully connected layer: three quarters of the traffic) run there,
        beside the next iteration's first kernels on the current stream, instead of at the end of this one.  The caller
        orders the stream behind the gradients and everything that reads those parameters behind the stream."""
        if closure is not None:
            raise ValueError("ClampAdam.step takes no closure")
        parameters_changed()
        from probnmn import parallel

        group = self.param_groups[0]
        self.step_count += 1
        dp = parallel.world() > 1
        # ONE launch: an item per contiguous run of arena parameters that share an Adam step count (one run per arena
        # once every module has been used -- at 128 questions per GPU a rarely sampled module can stay behind for good)
        # and an item per loose tensor; each item carries the bias corrections of its own count.
        ptrs, counts, steps_of = [], [], []
        side_ids = set(side_params) if side_stream is not None else set()
        for k, (a, (m, v), steps) in enumerate(zip(self.arenas, self._arena_state, self._arena_steps)):
            # every parameter on the first step, after load_state_dict and every 64th step; a rotating sample
            # otherwise (a re-pointed parameter would train on while the fused update writes the arena slice)
            if not a.intact(full=self._check_full or self.step_count % 64 == 0):
                raise _hip.HipLibraryError(
                    "a parameter no longer aliases the arena this optimizer was built on (model.to() / .data = "
                    "after the optimizer was constructed): build the optimizer after placing the model")
            if a.touched is None:
                steps += 1
            else:
                # data parallel: a parameter counts as touched when ANY rank's batch reached it (what one process on the
                # whole batch would see) -- one tiny host-side collective per step keeps the replicas' Adam states identical
                touched = parallel.host_or(a.touched) if dp else a.touched
                steps[(steps > 0) | touched] += 1
            if a.touched is not None:
                a.touched[:] = False
            base = np.array((a.flat.data_ptr(), a.grad.data_ptr(), m.data_ptr(), v.data_ptr()), np.uint64)
            # contiguous runs of parameters with one step count (alignment padding between two parameters rides along:
            # zero gradient, zero moments, stays zero)
            cuts = np.flatnonzero(steps[1:] != steps[:-1]) + 1
            first = np.concatenate(([0], cuts))
            bounds = self._arena_bounds[k]  # float offset of every parameter, then the arena's length
            lo, hi = bounds[first], bounds[np.concatenate((cuts, [len(steps)]))]
            live = steps[first] > 0
            if live.any():
                lo, hi = lo[live], hi[live]
                ptrs.append(base[None, :] + (4 * lo).astype(np.uint64)[:, None])
                counts.append(hi - lo)
                steps_of.append(steps[first][live])
        n_arena_items = sum(len(c) for c in counts)
        loose_rows, loose_side = [], []
        for i, (p, (m, v)) in enumerate(zip(self.loose, self._loose_state)):
            g = p.grad
            if g is None:
                if self._loose_steps[i] == 0:
                    continue  # (no gradient so far: no state, as torch.optim.Adam)
                # torch 1.4.0's zero_grad would have left a zero gradient behind: the momentum goes on
                g = self._loose_zero.get(i)
                if g is None:
                    g = self._loose_zero[i] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if not p.is_contiguous() or not g.is_contiguous():
                raise _hip.HipLibraryError("ClampAdam needs contiguous loose parameters and gradients")
            self._loose_steps[i] += 1
            loose_rows.append((p.data_ptr(), g.data_ptr(), self._loose_ptrs[i][0], self._loose_ptrs[i][1], self._loose_numel[i],
                               self._loose_steps[i]))
            on_side = id(p) in side_ids
            loose_side.append(on_side)
            if on_side:
                g.record_stream(side_stream)  # (read there after the caller has dropped the tensor)
        if loose_rows:
            lr_ = np.array(loose_rows, dtype=np.int64)
            ptrs.append(lr_[:, :4].astype(np.uint64))
            counts.append(lr_[:, 4])
            steps_of.append(lr_[:, 5])
        self._check_full = False
        if not ptrs:
            return
        ptrs, counts, steps_of = np.concatenate(ptrs), np.concatenate(counts), np.concatenate(steps_of).astype(np.float64)
        rec = np.zeros(len(counts), _hip.ADAM_ITEM)
        for c, name in enumerate(("param", "grad", "exp_avg", "exp_avg_sq")):
            rec[name] = ptrs[:, c]
        rec["n"] = counts
        # bias corrections in double, as torch.optim.Adam computes them on the host
        beta1, beta2 = group["betas"]
        rec["bc1"] = 1.0 - np.power(float(beta1), steps_of)
        rec["bc2_sqrt"] = np.sqrt(1.0 - np.power(float(beta2), steps_of))
        device = (self.arenas[0].flat if self.arenas else self.loose[0]).device
        clamp = float(group["clamp"]) if group["clamp"] is not None else 0.0

        def launch(items, blocks=2048):
            # (one launch, grid = (blocks, items).  Measured and not kept: the ~45 small tensors of the seq2seq models in a
            # launch of their own with 64 workgroups each -- their 90 000 mostly idle workgroups are cheaper than a second
            # launch: 0.357 -> 0.39 ms)
            if len(items):
                buf = _hip.to_device(items, device)
                _hip.check(
                    _hip.lib().pnmn_clamp_adam_blocks(buf.data_ptr(), len(items), float(group["lr"]), beta1, beta2, group["eps"],
                                                      group["weight_decay"], clamp, blocks, _hip.stream_ptr(device)),
                    "clamp_adam")

        if side_stream is None:
            launch(rec)
            return
        on_side = np.ones(len(rec), bool)
        on_side[n_arena_items:] = loose_side
        launch(rec[~on_side])
        with torch.cuda.stream(side_stream):  # (record upload and launch both on that stream)
            # two workgroups per CU: the next iteration's recurrent kernels (other stream) find wave slots on every CU
            launch(rec[on_side], SIDE_BLOCKS_PER_ITEM)
