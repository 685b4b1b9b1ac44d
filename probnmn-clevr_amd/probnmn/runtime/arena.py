"""Flat parameter / gradient arenas.

All trunk parameters of a network live in ONE device buffer, each in the layout the kernels read
(conv weights as [Cout][KH*KW][Cin], i.e. torch ``channels_last``), and the gradients in a second
buffer with identical offsets.  The ``nn.Parameter`` objects stay what the reference's callers
expect -- same names, same logical shapes, usable by ``state_dict`` / ``load_state_dict`` /
any ``torch.optim`` -- their ``.data`` simply aliases the arena.  What this buys on MI355X:
kernels address weights by offset tables (one launch serves many modules), the optimizer is one
streaming pass over one buffer, and a data-parallel gradient all-reduce is one RCCL call on one
contiguous 257 MB range instead of ~110 small ones.
"""
from typing import Dict, Iterable, List, Tuple

import torch
from torch import nn


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class ParamArena:
    def __init__(self, named_params: Iterable[Tuple[str, nn.Parameter]], device: torch.device):
        self.device = device
        self.names: List[str] = []
        self.offsets: Dict[str, int] = {}
        self.sizes: Dict[str, int] = {}
        self._params: Dict[str, nn.Parameter] = {}
        total = 0
        entries = list(named_params)
        for name, p in entries:
            self.names.append(name)
            self.offsets[name] = total
            self.sizes[name] = p.numel()
            self._params[name] = p
            total += _align(p.numel())
        self.total = total
        # Which parameters received a gradient since the optimizer last looked (bool per name, in ``names`` order), or
        # None: not tracked -- every parameter counts as touched.  The engine ORs in the modules of a backward pass's
        # valid programs; ClampAdam reads and clears it: torch.optim.Adam starts a parameter's state at its FIRST
        # gradient and, under the reference's torch 1.4.0 zero_grad (gradients zeroed in place, never dropped), updates
        # it in every later step.
        self.touched = None
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self._views: Dict[str, torch.Tensor] = {}
        self._gviews: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for name, p in entries:
                v = self._view(self.flat, name, p)
                v.copy_(p.detach().to(device))
                p.data = v
                self._views[name] = v
                self._gviews[name] = self._view(self.grad, name, p)

    def _view(self, buf: torch.Tensor, name: str, p: torch.Tensor) -> torch.Tensor:
        off, n = self.offsets[name], p.numel()
        seg = buf[off : off + n]
        if p.dim() == 4:  # [Cout,Cin,KH,KW] logical, [Cout][KH][KW][Cin] physical
            co, ci, kh, kw = p.shape
            return seg.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def view_of(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        """The slice of an arena-shaped buffer (e.g. an optimizer moment) that mirrors parameter ``name``,
        in the parameter's logical shape."""
        return self._view(buf, name, self._params[name])

    def intact(self, full: bool = False) -> bool:
        """True while the parameters still alias the arena (``.to()`` / ``.data =`` breaks it).  Called several
        times per step, so a call checks the first and last parameter (a module-wide ``.to()`` moves them all) and
        a rotating handful of the others -- every parameter is looked at within ~35 calls; ``full=True`` checks
        all of them at once (218 ``data_ptr()`` pairs, 0.3 ms)."""
        names = self.names
        if full or len(names) <= 8:
            check = names
        else:
            k = self.__dict__.get("_intact_cursor", 0)
            check = [names[0], names[-1]] + [names[(k + i) % len(names)] for i in range(6)]
            self._intact_cursor = (k + 6) % len(names)
        return all(self._params[n].data_ptr() == self._views[n].data_ptr() for n in check)

    def grad_view(self, name: str) -> torch.Tensor:
        return self._gviews[name]

    def attach_grads(self) -> None:
        """Point every parameter's ``.grad`` at its slice of the gradient arena."""
        for n in self.names:
            self._params[n].grad = self._gviews[n]

    def param(self, name: str) -> nn.Parameter:
        return self._params[name]
