"""Deterministic, torch-RNG-independent tensor generator for fixtures (test infrastructure).

Golden vectors store only *outputs*; inputs and weights are regenerated on either box from
numpy's Philox bit generator, whose stream is stable for a given numpy version (both boxes run
the same image).
"""
import math
from typing import Dict, Sequence

import numpy as np
import torch


def rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(int(seed)))


def normal(gen: np.random.Generator, shape: Sequence[int], std: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((gen.standard_normal(tuple(shape)) * std).astype(np.float32))


def uniform(gen: np.random.Generator, shape: Sequence[int], bound: float) -> torch.Tensor:
    return torch.from_numpy(gen.uniform(-bound, bound, tuple(shape)).astype(np.float32))


def fill_state_dict(shapes: Dict[str, Sequence[int]], seed: int) -> Dict[str, torch.Tensor]:
    """Weights ~ N(0, 2/fan_in) for >=2-D tensors (kaiming-normal scale, the reference's module
    init -- nmn_modules.py:77-79), biases ~ U(+-1/sqrt(fan_in of the matching weight))."""
    gen = rng(seed)
    out: Dict[str, torch.Tensor] = {}
    fan_in_of: Dict[str, int] = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            fan_in_of[name.rsplit(".", 1)[0]] = fan_in
            out[name] = normal(gen, shape, math.sqrt(2.0 / fan_in))
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if len(shape) < 2:
            fan_in = fan_in_of.get(name.rsplit(".", 1)[0], shape[0])
            out[name] = uniform(gen, shape, 1.0 / math.sqrt(fan_in))
    return out
