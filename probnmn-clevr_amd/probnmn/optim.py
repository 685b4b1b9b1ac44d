"""Fused gradient clamp + Adam over parameter arenas.

The reference clamps every gradient element to [-5, 5] in a Python loop and then calls one
``torch.optim.Adam`` that spans all trainable models (reference:
probnmn/trainers/module_training_trainer.py:94-96, joint_training_trainer.py:182-188,
_trainer.py:103-108,193).  On MI355X the whole update is one streaming kernel
(``pnmn_clamp_adam``): 28 bytes of HBM traffic per parameter, one launch for the 63 M-parameter
trunk arena plus one item per loose tensor.  Arithmetic follows ``torch.optim.Adam`` (no amsgrad).
"""
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from probnmn import _hip


class ClampAdam:
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
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.clamp = clamp
        self.arenas = list(arenas)
        in_arena = set()
        for a in self.arenas:
            in_arena.update(id(a.param(n)) for n in a.names)
        self.loose: List[nn.Parameter] = [p for p in params if id(p) not in in_arena and p.requires_grad]
        self.step_count = 0
        self._arena_state = [(torch.zeros_like(a.flat), torch.zeros_like(a.flat)) for a in self.arenas]
        self._loose_state = [(torch.zeros_like(p, memory_format=torch.contiguous_format),
                              torch.zeros_like(p, memory_format=torch.contiguous_format)) for p in self.loose]

    def zero_grad(self) -> None:
        # arena gradients are zeroed by the engine at the start of each backward
        for p in self.loose:
            p.grad = None

    @torch.no_grad()
    def step(self) -> None:
        self.step_count += 1
        items = []
        for a, (m, v) in zip(self.arenas, self._arena_state):
            items.append((a.flat.data_ptr(), a.grad.data_ptr(), m.data_ptr(), v.data_ptr(), a.total))
        for p, (m, v) in zip(self.loose, self._loose_state):
            if p.grad is None:
                continue
            if not p.is_contiguous() or not p.grad.is_contiguous():
                raise _hip.HipLibraryError("ClampAdam needs contiguous loose parameters and gradients")
            items.append((p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()))
        if not items:
            return
        rec = np.zeros(len(items), _hip.ADAM_ITEM)
        for i, it in enumerate(items):
            rec[i]["param"], rec[i]["grad"], rec[i]["exp_avg"], rec[i]["exp_avg_sq"], rec[i]["n"] = it
        device = (self.arenas[0].flat if self.arenas else self.loose[0]).device
        buf = _hip.to_device(rec, device)
        clamp = float(self.clamp) if self.clamp is not None else 0.0
        _hip.check(
            _hip.lib().pnmn_clamp_adam(buf.data_ptr(), len(items), self.lr, self.betas[0], self.betas[1], self.eps,
                                       self.weight_decay, clamp, self.step_count, _hip.stream_ptr(device)),
            "clamp_adam")
