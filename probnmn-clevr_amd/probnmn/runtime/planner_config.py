"""What the engine hands to the library's trunk planner (``pnmn_trunk_planner_create``, csrc/host_trunk.hip): the
per-token weight offsets and the planner's few knobs.  ``runtime/schedule.py`` -- the numpy planner the library's is tested
against word for word -- takes the same two objects, and is imported by tests only."""
from dataclasses import dataclass

import numpy as np


@dataclass
class WeightTables:
    """Float offsets (into the parameter / gradient arenas, which mirror each other) per program
    token; -1 where the token has no such weight.  ``wt3`` indexes the transposed-weight arena."""

    w3: np.ndarray  # [V, 6]  projection, conv1..conv5 weights
    b3: np.ndarray  # [V, 6]  ... biases
    wt3: np.ndarray  # [V, 6]  transposed copies (dgrad operand)
    dotw: np.ndarray  # [V]  conv3 (attention) / conv6 (relate) / conv (same) weight
    dotb: np.ndarray  # [V]


@dataclass
class PlannerConfig:
    """Knobs of the trunk planner (fields of ``pnmn_trunk_config``).

    ``fuse_mask_bwd`` -- backward of ``feats * attn`` in front of a masked conv: 2 = d(attention) in the data-gradient's
    epilogue, d(feats) deferred to ONE gather at the end of the backward pass (default); 1 = both fused into the epilogue
    (fp32 atomics / read-modify-write of the 100 KB d(feats) map per masked conv: data gradients ran ~15 % behind the
    forward convs); 0 = a separate kernel per level.  ``sort_by_weight`` False: a launch's items in batch order (tests /
    HBM-traffic experiments)."""

    wgrad_chunk: int = 8      # items per weight-gradient job (upper bound; finer for small batches)
    wgrad_groups: int = 4
    fuse_mask_bwd: int = 2
    sole_writer_rmw: bool = True
    sort_by_weight: bool = True
