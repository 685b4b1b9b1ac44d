"""Batch scheduler: compiled programs -> level-ordered kernel work lists.

The reference runs one example at a time, one torch op at a time (reference:
probnmn/models/nmn.py:197-238) -- thousands of tiny launches per step.  Here every module call of
every example in the batch is expanded into *primitives* (conv3x3, projection, 1-channel head,
Same, And/Or), each primitive gets the dependency level at which its inputs are ready, and all
primitives of one (level, kind) become ONE grouped kernel launch whose work list says, per item,
which example's buffers and which token's weights to use.

Host cost matters (a GPU step is a few milliseconds), so the per-batch work is a fixed number of
numpy operations, independent of batch size and of how many different programs the batch holds:
a program *structure* (its calls' kinds and wiring, tokens ignored) is expanded once into a dense
integer table of primitives (:class:`Template`, cached in a bank); a batch gathers its examples'
tables with one fancy index, and every address / weight pointer of every primitive is computed
by whole-array arithmetic (per-kind base + per-example stride + in-block offset; weights through
per-token offset tables).  Records are assembled as ``uint64`` matrices that are bit-identical to
the C structs of include/probnmn_hip.h.

Value placement: each example owns a contiguous block of the activation arena laid out by its
template; the gradient arena mirrors it offset-for-offset.  The value a program returns is
placed directly in the classifier's input row (``FINAL``).
"""
import os
from dataclasses import dataclass
from typing import Dict, List, NamedTuple, Sequence, Tuple

import numpy as np

from probnmn.runtime.planner_config import PlannerConfig, WeightTables  # noqa: F401

from . import program_compiler as pc

HW_ALIGN = 64  # floats; keeps every slot 256-byte aligned
TOKEN_ROW = pc.TOKEN_ROW  # calls per program of the fixed-width token rows (longer programs: a per-example copy loop)

# operand location kinds
L_SLOT, L_FEAT, L_ONES, L_FINAL = 0, 1, 2, 3
# primitive kinds
K_CONV, K_PROJ, K_DOT, K_SAME, K_MINMAX = 0, 1, 2, 3, 4
KIND_NAMES = ("conv", "proj", "dot", "same", "minmax")

RELATE_DILATIONS = (1, 2, 4, 8, 1)  # reference nmn_modules.py:146-150

# columns of a template's primitive table
(C_KIND, C_LEVEL, C_CALL, C_WIDX, C_DIL, C_AK, C_AO, C_BK, C_BO, C_OK, C_OO, C_ACH, C_BCH, C_ISMAX,
 C_MASKED, C_SCRATCH, C_PA, C_PB) = range(18)
NCOLS = 18  # C_PA/C_PB: local index of the primitive producing input a/b (-1: stem output / ones)


def _align(n: int) -> int:
    return (n + HW_ALIGN - 1) // HW_ALIGN * HW_ALIGN


@dataclass
class Template:
    """One program structure: ``table`` is [n_prims, NCOLS]; offsets are floats relative to the
    example's arena block (values first, then one scratch map per masked conv)."""

    table: np.ndarray
    n_calls: int
    size: int  # arena floats per example (values + backward scratch)
    result_is_feat: bool
    depth: int


def structure_key(prog: pc.CompiledProgram) -> Tuple:
    """Kinds, wiring and channel counts of the calls (tokens ignored) + the result value."""
    if prog._skey is None:  # (programs out of the batch compiler carry it)
        t = prog.table().copy()
        t[:, 1] = 0
        prog._skey = t.tobytes()
    return (prog._skey, prog.result)


def build_template(prog: pc.CompiledProgram, hw: int, channels: int) -> Template:
    big, small = _align(hw * channels), _align(hw)
    calls = prog.calls
    # liveness: only calls that reach the result are executed (a dead call's output feeds nothing;
    # validity was already decided on the full program)
    needed = [False] * len(calls)
    stack = [prog.result]
    while stack:
        v = stack.pop()
        if v >= 2 and not needed[v - 2]:
            needed[v - 2] = True
            stack.extend((calls[v - 2].a, calls[v - 2].b))

    cursor = 0

    def alloc(n: int) -> int:
        nonlocal cursor
        off = cursor
        cursor += n
        return off

    loc: Dict[int, Tuple[int, int]] = {pc.FEAT: (L_FEAT, 0), pc.ONES: (L_ONES, 0)}
    lvl: Dict[int, int] = {pc.FEAT: 0, pc.ONES: 0}
    prod: Dict[int, int] = {pc.FEAT: -1, pc.ONES: -1}  # value id -> local index of its producing primitive
    rows: List[List[int]] = []

    def prim(kind, level, call, widx=0, dil=1, a=(L_ONES, 0), b=(L_ONES, 0), out=(L_SLOT, 0), a_ch=0,
             b_ch=0, is_max=0, masked=0, pa=-1, pb=-1):
        rows.append([kind, level, call, widx, dil, a[0], a[1], b[0], b[1], out[0], out[1], a_ch, b_ch,
                     is_max, masked, -1, pa, pb])
        return len(rows) - 1

    for ci, c in enumerate(calls):
        if not needed[ci]:
            continue
        vid = ci + 2
        is_result = vid == prog.result
        if c.out_channels == channels:
            out = (L_FINAL, 0) if is_result else (L_SLOT, alloc(big))
        else:
            out = (L_SLOT, alloc(small))
        if c.kind in (pc.AND, pc.OR):
            level = max(lvl[c.a], lvl[c.b]) + 1
            last = prim(K_MINMAX, level, ci, a=loc[c.a], b=loc[c.b], out=out, a_ch=c.a_channels, b_ch=c.b_channels,
                        is_max=int(c.kind == pc.OR), pa=prod[c.a], pb=prod[c.b])
        elif c.kind == pc.SAME:
            level = lvl[c.a] + 1
            last = prim(K_SAME, level, ci, a=(L_FEAT, 0), b=loc[c.a], out=out, pb=prod[c.a])
        elif c.kind == pc.CMP:
            level = max(lvl[c.a], lvl[c.b]) + 1
            t0 = (L_SLOT, alloc(big))
            t1 = (L_SLOT, alloc(big))
            j0 = prim(K_PROJ, level, ci, widx=0, a=loc[c.a], b=loc[c.b], out=t0, pa=prod[c.a], pb=prod[c.b])
            j1 = prim(K_CONV, level + 1, ci, widx=1, a=t0, out=t1, pa=j0)
            last = prim(K_CONV, level + 2, ci, widx=2, a=t1, out=out, pa=j1)
            level += 2
        else:  # ATT / QUERY / REL
            nconv = 5 if c.kind == pc.REL else 2
            dils = RELATE_DILATIONS if c.kind == pc.REL else (1, 1)
            level = lvl[c.a]
            src = (L_FEAT, 0)
            last = -1
            for k in range(nconv):
                level += 1
                last_is_out = (k == nconv - 1) and c.kind == pc.QUERY
                dst = out if last_is_out else (L_SLOT, alloc(big))
                if k == 0:  # input is FEAT * attention (L_ONES -> no multiply)
                    last = prim(K_CONV, level, ci, widx=1, dil=dils[0], a=src, b=loc[c.a], out=dst, masked=1,
                                pb=prod[c.a])
                else:
                    last = prim(K_CONV, level, ci, widx=k + 1, dil=dils[k], a=src, out=dst, pa=last)
                src = dst
            if c.kind != pc.QUERY:
                level += 1
                last = prim(K_DOT, level, ci, a=src, out=out, pa=last)
        loc[vid] = out
        lvl[vid] = level
        prod[vid] = last

    for r in rows:  # backward scratch: gradient wrt (FEAT * attention) of each masked conv
        if r[C_MASKED]:
            r[C_SCRATCH] = alloc(big)
    table = np.asarray(rows, dtype=np.int64).reshape(-1, NCOLS)
    depth = int(table[:, C_LEVEL].max()) if len(rows) else 0
    return Template(table, len(calls), cursor, prog.result < 2, depth)


# ------------------------------------------------------------------------------------------------
@dataclass
class Buffers:
    """Device base addresses (bytes) for one step."""

    params: int
    grads: int
    wt: int
    act: int
    gact: int
    feat: int  # stem output  [B][HW][C]
    gfeat: int
    final: int  # classifier input [B][HW][C]
    gfinal: int
    ones: int  # [HW] of 1.0


class Launch(NamedTuple):
    kind: str
    level: int
    begin: int
    end: int


@dataclass
class StepPlan:
    records: Dict[str, np.ndarray]  # kind -> record array (sorted by level / by weight)
    forward: List[Launch]
    backward: List[List[Launch]]  # phases per level (reverse level order)
    wgrad_jobs: Dict[str, np.ndarray]
    arena_floats: int
    feat_result_examples: np.ndarray  # examples whose program returns FEAT itself
    n_prims: int
    # module-conv weight-gradient jobs grouped by forward level range, deepest first:
    # (lowest level of the group, first job, one past the last job) -- a group may be launched as soon
    # as the backward pass has finished the phase of its lowest level
    wgrad_groups: List[Tuple[int, int, int]] = None


def _cut(levels: np.ndarray) -> List[Tuple[int, int, int]]:
    """(level, begin, end) runs of a sorted level array."""
    if levels.size == 0:
        return []
    cuts = np.flatnonzero(np.diff(levels)) + 1
    bounds = np.concatenate(([0], cuts, [levels.size]))
    return [(int(levels[b]), int(b), int(e)) for b, e in zip(bounds[:-1], bounds[1:])]


class BatchScheduler:
    def __init__(self, hw: int, channels: int, tables: WeightTables, record_dtypes: Dict[str, np.dtype],
                 wgrad_chunk: int = 8, wgrad_groups: int = 4):
        self.hw = hw
        self.channels = channels
        self.tables = tables
        self.dt = record_dtypes
        self.wgrad_chunk = wgrad_chunk
        self.wgrad_groups = wgrad_groups
        # backward of `feats * attn` in front of a masked conv: 2 = d(attention) in the data-gradient's epilogue, d(feats)
        # deferred to ONE gather at the end of the backward pass (default); 1 = both fused into the epilogue (fp32
        # atomics / read-modify-write of the 100 KB d(feats) map per masked conv: data gradients ran ~15 % behind
        # the forward convs); 0 = a separate kernel per level
        self.fuse_mask_bwd = 2
        self.sole_writer_rmw = True
        self.sort_by_weight = True  # (False: a launch's items in batch order -- tests / HBM-traffic experiments)
        self._tables64 = tuple(np.ascontiguousarray(a, dtype=np.int64)
                               for a in (tables.w3, tables.b3, tables.wt3, tables.dotw, tables.dotb))
        self._tables64_ptrs = tuple(a.ctypes.data for a in self._tables64)
        self._ids: Dict[Tuple, int] = {}
        self._templates: List[Template] = []
        self._bank = None  # (tables [T, Pmax, NCOLS], nprims [T], sizes [T])

    # ---- template bank -------------------------------------------------------------------------
    def template_id(self, prog: pc.CompiledProgram) -> int:
        tid = prog._template_id
        if tid is not None and prog._template_owner is self:
            return tid
        key = structure_key(prog)
        tid = self._ids.get(key)
        if tid is None:
            tid = len(self._templates)
            self._templates.append(build_template(prog, self.hw, self.channels))
            self._ids[key] = tid
            self._bank = None
        prog._template_id = tid
        prog._template_owner = self
        if prog._tokens is None:  # (programs out of the batch compiler carry both)
            prog._tokens = prog.table()[:, 1].astype(np.int64)
            if prog._tokens.size <= TOKEN_ROW:  # fixed-width copy: a batch's token matrix is then one np.array() call
                row = np.zeros(TOKEN_ROW, np.int64)
                row[: prog._tokens.size] = prog._tokens
                prog._tokens_row = row
        return tid

    def template(self, prog: pc.CompiledProgram) -> Template:
        return self._templates[self.template_id(prog)]

    def _get_bank(self):
        if self._bank is None:
            pmax = max([t.table.shape[0] for t in self._templates] + [1])
            tables = np.zeros((len(self._templates), pmax, NCOLS), np.int64)
            for i, t in enumerate(self._templates):
                tables[i, : t.table.shape[0]] = t.table
            nprims = np.asarray([t.table.shape[0] for t in self._templates], np.int64)
            sizes = np.asarray([t.size for t in self._templates], np.int64)
            isfeat = np.asarray([t.result_is_feat for t in self._templates], bool)
            self._bank = (tables, nprims, sizes, isfeat)
        return self._bank

    def arena_floats(self, programs: Sequence[pc.CompiledProgram]) -> int:
        """Activation-arena size (floats) the batch needs; the gradient arena mirrors it."""
        return self._prepare(programs)[5]

    def _prepare(self, programs: Sequence[pc.CompiledProgram]):
        """Per-batch index arrays shared by ``arena_floats`` and ``plan`` (the engine sizes its buffers with the
        first and plans with the second): valid examples, their template ids, arena block bases."""
        hit = self.__dict__.get("_prepared")
        if hit is not None and hit[0] is programs:
            return hit
        ex_valid, ids = [], []
        for e, p in enumerate(programs):
            if p.valid:
                ex_valid.append(e)
                ids.append(p._template_id if p._template_owner is self else self.template_id(p))
        nv = len(ex_valid)
        tids = np.asarray(ids, dtype=np.int64)
        sizes = self._get_bank()[2]
        blk = sizes[tids] if nv else np.zeros(0, np.int64)
        base = np.cumsum(blk) - blk  # first float of each example's arena block
        self._prepared = (programs, ex_valid, tids, np.asarray(ex_valid, dtype=np.int64), base, int(blk.sum()))
        return self._prepared

    # --------------------------------------------------------------------------------------------
    def plan_numpy(self, programs: Sequence[pc.CompiledProgram], buf: Buffers) -> StepPlan:
        """The planner as whole-array numpy arithmetic: the specification ``plan`` (the library's
        ``pnmn_plan_batch``) is checked against, record for record (tests/test_schedule.py).  Not used by the
        engine: ~250 numpy calls cost 0.75 ms per 65 programs, on the critical path of a small-batch step."""
        hw, C = self.hw, self.channels
        map_bytes = hw * C * 4
        tb = self.tables
        u64 = np.uint64

        ex_valid = [e for e, p in enumerate(programs) if p.valid]
        tids = np.asarray([self.template_id(programs[e]) for e in ex_valid], dtype=np.int64)
        tables, nprims, sizes, isfeat = self._get_bank()
        E = np.asarray(ex_valid, dtype=np.int64)
        nv = E.size
        empty = {
            "conv": np.zeros(0, self.dt["conv"]), "proj": np.zeros(0, self.dt["conv"]),
            "dgrad": np.zeros(0, self.dt["conv"]), "pdgrad": np.zeros(0, self.dt["conv"]),
            "dot": np.zeros(0, self.dt["dot"]), "same": np.zeros(0, self.dt["same"]),
            "minmax": np.zeros(0, self.dt["minmax"]), "maskbwd": np.zeros(0, self.dt["maskbwd"]),
            "wg3": np.zeros(0, self.dt["wgrad_item"]), "wgp": np.zeros(0, self.dt["wgrad_item"]),
        }
        empty_jobs = {"wg3": np.zeros(0, self.dt["wgrad_job"]), "wgp": np.zeros(0, self.dt["wgrad_job"])}
        if nv == 0:
            return StepPlan(empty, [], [], empty_jobs, 0, np.zeros(0, np.int64), 0, [])

        # per-example arena block
        blk = sizes[tids]
        base = np.cumsum(blk) - blk  # floats
        arena = int(blk.sum())
        feat_result = E[isfeat[tids]]

        # tokens of every call, padded
        cmax = max(1, max(programs[e]._tokens.size for e in ex_valid))
        tokens = np.zeros((nv, cmax), np.int64)
        for i, e in enumerate(ex_valid):
            t = programs[e]._tokens
            tokens[i, : t.size] = t

        # gather every example's primitive table and drop the padding
        P = tables[tids]  # [nv, Pmax, NCOLS]
        keep = np.arange(P.shape[1])[None, :] < nprims[tids][:, None]
        rows = P[keep]  # [N, NCOLS]
        xi = np.broadcast_to(np.arange(nv)[:, None], keep.shape)[keep]  # index into the valid list
        N = rows.shape[0]
        chunk = min(self.wgrad_chunk, max(2, N // 256))  # (items per weight-gradient job: finer for small batches)
        ex = E[xi]
        tok = tokens[xi, rows[:, C_CALL]]
        blockbase = base[xi]

        kind_base = np.asarray([buf.act, buf.feat, 0, buf.final], np.int64)
        kind_gbase = np.asarray([buf.gact, buf.gfeat, 0, buf.gfinal], np.int64)
        kind_stride = np.asarray([0, map_bytes, 0, map_bytes], np.int64)

        def addr(kcol, ocol, grad=False, ones=0):
            k = rows[:, kcol]
            a = (kind_gbase if grad else kind_base)[k] + kind_stride[k] * ex
            a = a + (k == L_SLOT) * ((blockbase + rows[:, ocol]) * 4)
            if ones:
                a = np.where(k == L_ONES, ones, a)
            return a

        a_f, a_g = addr(C_AK, C_AO), addr(C_AK, C_AO, grad=True)
        b_f, b_g = addr(C_BK, C_BO), addr(C_BK, C_BO, grad=True)
        o_f, o_g = addr(C_OK, C_OO), addr(C_OK, C_OO, grad=True)
        level = rows[:, C_LEVEL]
        kind = rows[:, C_KIND]
        widx = rows[:, C_WIDX]
        dil = rows[:, C_DIL]

        records: Dict[str, np.ndarray] = {}
        launches: Dict[str, List[Tuple[int, int, int]]] = {}

        def finish(name, mat, lv, dtype_key, wcol=None):
            """sort by level (and, inside a level, by weight: the conv kernels deal contiguous ranges of a
            launch's items to the XCDs, so each XCD's L2 holds the one or two weights its items use instead of
            every weight of the level -- 3x less HBM read traffic, scripts/pmc_conv.sh), view as records, cut
            launches"""
            if wcol is None or not self.sort_by_weight:
                idx = np.argsort(lv, kind="stable")
            else:
                idx = np.lexsort((mat[:, wcol], lv))
            records[name] = np.ascontiguousarray(mat[idx]).view(self.dt[dtype_key]).reshape(-1)
            launches[name] = _cut(lv[idx])
            return idx

        # ---- 3x3 convs ---------------------------------------------------------------------------
        m = kind == K_CONV
        n = int(m.sum())
        if n:
            t_, w_, lv = tok[m], widx[m], level[m]
            masked = rows[m, C_MASKED] == 1
            mask_ptr = np.where(masked, b_f[m], 0)  # 0 for the all-ones attention too
            w_off, b_off, wt_off = tb.w3[t_, w_], tb.b3[t_, w_], tb.wt3[t_, w_]
            scratch = buf.gact + (blockbase[m] + rows[m, C_SCRATCH]) * 4
            fw = np.zeros((n, 12), u64)
            fw[:, 0], fw[:, 2] = a_f[m], mask_ptr
            fw[:, 4], fw[:, 5], fw[:, 6] = buf.params + w_off * 4, buf.params + b_off * 4, o_f[m]
            fw[:, 7] = dil[m]  # dilation in the low 32 bits, flags = 0
            dg = np.zeros((n, 12), u64)
            dg[:, 0], dg[:, 3], dg[:, 4] = o_g[m], o_f[m], buf.wt + wt_off * 4
            if self.fuse_mask_bwd == 2:
                # deferred d(feats): dx goes to the conv's scratch map, d(attention) stays in the epilogue, and
                # pnmn_feat_grad_gather sums the scratch maps per example at the end of the backward pass
                att = masked & (mask_ptr != 0)
                dg[:, 6] = np.where(masked, scratch, a_g[m])
                dg[:, 7] = dil[m] + np.where(att, 16 << 32, 0)
                dg[:, 8] = np.where(att, a_f[m], 0)
                dg[:, 9] = np.where(att, mask_ptr, 0)
                dg[:, 11] = np.where(att, b_g[m], 0)
            elif self.fuse_mask_bwd:
                # masked convs: the data-gradient kernel adds straight into dFEAT / d(attention)
                dg[:, 6] = np.where(masked, 0, a_g[m])
                # flags: fused mask backward; + "sole writer" when no other masked conv of the same level
                # (= the same launch) adds into this example's dFEAT map, which lets the kernel use a plain
                # read-modify-write instead of 25 000 atomics per item
                _, inv, cnt = np.unique(lv.astype(np.int64) * (1 << 48) + (a_g[m] >> 4).astype(np.int64) * masked,
                                        return_inverse=True, return_counts=True)
                sole = masked & (cnt[inv] == 1) & self.sole_writer_rmw
                dg[:, 7] = dil[m] + np.where(masked, 4 << 32, 0) + np.where(sole, 8 << 32, 0)
                dg[:, 8] = np.where(masked, a_f[m], 0)
                dg[:, 9] = mask_ptr
                dg[:, 10] = np.where(masked, a_g[m], 0)
                dg[:, 11] = np.where(masked & (mask_ptr != 0), b_g[m], 0)
            else:
                dg[:, 6] = np.where(masked, scratch, a_g[m])
                dg[:, 7] = dil[m]
            wg = np.zeros((n, 6), u64)
            wg[:, 0], wg[:, 2], wg[:, 3], wg[:, 4], wg[:, 5] = a_f[m], mask_ptr, o_g[m], o_f[m], dil[m]
            idx = finish("conv", fw, lv, "conv", wcol=4)
            finish("dgrad", dg, lv, "conv", wcol=4)
            # mask backward for the masked convs (same level order as the dgrads)
            mm = masked[idx]
            if mm.any() and self.fuse_mask_bwd == 2:  # the gather's items, sorted by the d(feats) map they add into
                src = idx[mm]
                src = src[np.argsort(a_g[m][src], kind="stable")]
                mb = np.zeros((src.size, 5), u64)
                mb[:, 0], mb[:, 1], mb[:, 2], mb[:, 3] = scratch[src], a_f[m][src], mask_ptr[src], a_g[m][src]
                records["maskbwd"] = mb.view(self.dt["maskbwd"]).reshape(-1)
            if mm.any() and not self.fuse_mask_bwd:
                src = idx[mm]
                mb = np.zeros((src.size, 5), u64)
                mb[:, 0], mb[:, 1], mb[:, 2] = scratch[src], a_f[m][src], mask_ptr[src]
                mb[:, 3] = a_g[m][src]
                mb[:, 4] = np.where(mask_ptr[src] != 0, b_g[m][src], 0)
                records["maskbwd"] = mb.view(self.dt["maskbwd"]).reshape(-1)
                launches["maskbwd"] = _cut(lv[src])
            wkey = t_ * 8 + w_
            depth3 = int(lv.max())
            grp = (depth3 - lv) * self.wgrad_groups // max(depth3, 1)  # 0 = deepest levels
            records["wg3"], jobs3, jgrp = self._wgrad_jobs(wg, grp * 4096 + wkey, buf.grads + w_off * 4,
                                                           buf.grads + b_off * 4, group=grp, chunk=chunk)
            wgroups = []
            for gid, jb, je in _cut(jgrp):
                wgroups.append((int(lv[grp == gid].min()), jb, je))
        else:
            jobs3 = empty_jobs["wg3"]
            wgroups = []

        # ---- projections (ComparisonModule) ------------------------------------------------------
        m = kind == K_PROJ
        n = int(m.sum())
        if n:
            t_, lv = tok[m], level[m]
            w_off, b_off, wt_off = tb.w3[t_, 0], tb.b3[t_, 0], tb.wt3[t_, 0]
            fw = np.zeros((n, 12), u64)
            fw[:, 0], fw[:, 1] = a_f[m], b_f[m]
            fw[:, 4], fw[:, 5], fw[:, 6] = buf.params + w_off * 4, buf.params + b_off * 4, o_f[m]
            fw[:, 7] = 1
            finish("proj", fw, lv, "conv", wcol=4)
            # two dgrads (one per operand), accumulate flag set (non-atomic read-modify-write), ONE launch per level -- unless
            # ANY map is both some item's first and some item's second operand in the batch: then the halves stay apart
            # (host_plan.hip).  Why the halves themselves never alias: the interpreter is a two-register machine (nmn.py:197-238);
            # `saved_output` may feed several binary modules of one program, but each of them also takes the running `output`,
            # which depends on the previous one -- so two uses of one map inside a program lie on different levels, and
            # different examples own disjoint arena blocks (tests/test_trunk_planner.py checks the invariant on sampled programs).
            pd = np.zeros((2 * n, 12), u64)
            pd[:, 0], pd[:, 3] = np.tile(o_g[m], 2), np.tile(o_f[m], 2)
            pd[:n, 4], pd[n:, 4] = buf.wt + wt_off * 4, buf.wt + (wt_off + C * C) * 4
            pd[:n, 6], pd[n:, 6] = a_g[m], b_g[m]
            pd[:, 7] = 1 | (1 << 32)
            odd = 1 if np.intersect1d(a_g[m], b_g[m]).size else 0
            finish("pdgrad", pd, np.concatenate((lv * 2, lv * 2 + odd)), "conv", wcol=4)
            wg = np.zeros((n, 6), u64)
            wg[:, 0], wg[:, 1], wg[:, 3], wg[:, 4] = a_f[m], b_f[m], o_g[m], o_f[m]
            records["wgp"], jobsp, _ = self._wgrad_jobs(wg, t_, buf.grads + w_off * 4, buf.grads + b_off * 4, chunk=chunk)
        else:
            jobsp = empty_jobs["wgp"]

        # ---- one-channel heads -----------------------------------------------------------------
        m = kind == K_DOT
        n = int(m.sum())
        if n:
            t_ = tok[m]
            r = np.zeros((n, 8), u64)
            r[:, 0], r[:, 1], r[:, 2] = a_f[m], buf.params + tb.dotw[t_] * 4, buf.params + tb.dotb[t_] * 4
            r[:, 3], r[:, 4], r[:, 5] = o_f[m], o_g[m], a_g[m]
            r[:, 6], r[:, 7] = buf.grads + tb.dotw[t_] * 4, buf.grads + tb.dotb[t_] * 4
            finish("dot", r, level[m], "dot")

        m = kind == K_SAME
        n = int(m.sum())
        if n:
            t_ = tok[m]
            bk = rows[m, C_BK]
            r = np.zeros((n, 10), u64)
            r[:, 0], r[:, 1] = a_f[m], np.where(bk == L_ONES, buf.ones, b_f[m])
            r[:, 2], r[:, 3] = buf.params + tb.dotw[t_] * 4, buf.params + tb.dotb[t_] * 4
            r[:, 4], r[:, 5], r[:, 6], r[:, 7] = o_f[m], o_g[m], a_g[m], b_g[m]  # dattn = 0 for all-ones
            r[:, 8], r[:, 9] = buf.grads + tb.dotw[t_] * 4, buf.grads + tb.dotb[t_] * 4
            finish("same", r, level[m], "same")

        m = kind == K_MINMAX
        n = int(m.sum())
        if n:
            ak, bk = rows[m, C_AK], rows[m, C_BK]
            r = np.zeros((n, 8), u64)
            r[:, 0] = np.where(ak == L_ONES, buf.ones, a_f[m])
            r[:, 1] = np.where(bk == L_ONES, buf.ones, b_f[m])
            r[:, 2], r[:, 3], r[:, 4], r[:, 5] = o_f[m], o_g[m], a_g[m], b_g[m]
            r[:, 6] = rows[m, C_ACH] | (rows[m, C_BCH] << 32)
            r[:, 7] = rows[m, C_ISMAX]
            finish("minmax", r, level[m], "minmax")

        for k, v in empty.items():
            records.setdefault(k, v)

        fwd, bwd = self._order(launches, int(level.max()) if N else 0)
        return StepPlan(records, fwd, bwd, {"wg3": jobs3, "wgp": jobsp}, arena, feat_result, N, wgroups)

    @staticmethod
    def _order(launches: Dict[str, List[Tuple[int, int, int]]], depth: int):
        """Forward launch list (by level; within a level And/Or, Same, heads, projections, convs) and the backward
        phases (levels in reverse) from the per-kind (level, begin, end) cuts."""
        at: Dict[int, Dict[str, Tuple[int, int]]] = {}
        for k in ("conv", "proj", "dot", "same", "minmax", "dgrad", "maskbwd"):
            for lv, b, e in launches.get(k, []):
                at.setdefault(lv, {})[k] = (b, e)
        pd_at: Dict[int, List[Tuple[int, int]]] = {}
        for lv, b, e in launches.get("pdgrad", []):
            pd_at.setdefault(lv // 2, []).append((b, e))

        fwd: List[Launch] = []
        bwd: List[List[Launch]] = []
        for lv in range(1, depth + 1):
            here = at.get(lv, {})
            for k in ("minmax", "same", "dot", "proj", "conv"):
                if k in here:
                    fwd.append(Launch(k, lv, *here[k]))
        for lv in range(depth, 0, -1):
            here = at.get(lv, {})
            phase: List[Launch] = []
            for k in ("minmax", "same", "dot"):  # the forward records carry the backward fields
                if k in here:
                    phase.append(Launch(k + "_bwd", lv, *here[k]))
            for b, e in pd_at.get(lv, []):
                phase.append(Launch("pdgrad", lv, b, e))
            for k in ("dgrad", "maskbwd"):
                if k in here:
                    phase.append(Launch(k, lv, *here[k]))
            if phase:
                bwd.append(phase)
        return fwd, bwd

    # --------------------------------------------------------------------------------------------
    _RECORD_KINDS = (("conv", "conv"), ("dgrad", "conv"), ("wg3", "wgrad_item"), ("jobs3", "wgrad_job"), ("proj", "conv"),
                     ("pdgrad", "conv"), ("wgp", "wgrad_item"), ("jobsp", "wgrad_job"), ("dot", "dot"), ("same", "same"),
                     ("minmax", "minmax"), ("maskbwd", "maskbwd"))
    _CUT_KINDS = ("conv", "proj", "dot", "same", "minmax", "dgrad", "maskbwd", "pdgrad", "wgroup")

    def plan(self, programs: Sequence[pc.CompiledProgram], buf: Buffers) -> StepPlan:
        """Work lists of one batch: ONE call of the library's host routine ``pnmn_plan_batch`` (csrc/host_plan.hip)
        over the template bank; see ``plan_numpy`` for the arithmetic."""
        from probnmn import _hip

        _, ex_valid, tids, E, base, arena = self._prepare(programs)
        self._prepared = None  # (one batch, one plan: do not keep the batch's programs alive)
        nv = len(ex_valid)
        dt = self.dt
        records = {}
        if nv == 0:
            records = {name: np.zeros(0, dt[key]) for name, key in self._RECORD_KINDS if not name.startswith("jobs")}
            return StepPlan(records, [], [], {"wg3": np.zeros(0, dt["wgrad_job"]), "wgp": np.zeros(0, dt["wgrad_job"])},
                            0, np.zeros(0, np.int64), 0, [])
        tables, nprims, sizes, isfeat = self._get_bank()
        feat_result = E[isfeat[tids]]
        rows = [programs[e]._tokens_row for e in ex_valid]
        if all(r is not None for r in rows):
            cmax = TOKEN_ROW
            tokens = np.array(rows)
        else:
            cmax = max(1, max(programs[e]._tokens.size for e in ex_valid))
            tokens = np.zeros((nv, cmax), np.int64)
            for i, e in enumerate(ex_valid):
                t = programs[e]._tokens
                tokens[i, : t.size] = t
        n_total = int(nprims[tids].sum())
        # (items per weight-gradient job: finer for small batches, as the library's trunk planner does)
        chunk = min(self.wgrad_chunk, max(2, n_total // 256))
        words = np.empty(n_total * 48 + 64, np.uint64)  # (a projection: 12 + 24 + 6 + 3 words; a masked conv: 38)
        meta = np.zeros(40, np.int64)
        cuts = self.__dict__.get("_cuts")
        if cuts is None:  # (64 KB of scratch the library fills; copied out below)
            cuts = self._cuts = np.empty((4096, 4), np.int32)
        rec = np.zeros(1, _hip.PLAN_IN)
        rec[0] = (tables.ctypes.data, nprims.ctypes.data, tids.ctypes.data, E.ctypes.data, base.ctypes.data,
                  tokens.ctypes.data) + self._tables64_ptrs + (
            buf.params, buf.grads, buf.wt, buf.act, buf.gact, buf.feat, buf.gfeat, buf.final, buf.gfinal, buf.ones,
            tables.shape[0], tables.shape[1], nv, cmax, self.hw, self.channels, chunk, self.wgrad_groups,
            int(self.fuse_mask_bwd), int(self.sole_writer_rmw),
            int(self.sort_by_weight), 0)
        _hip.check(_hip.lib().pnmn_plan_batch(rec.ctypes.data, words.ctypes.data, words.size, meta.ctypes.data,
                                              cuts.ctypes.data, cuts.shape[0]), "plan_batch")
        jobs = {}
        for k, (name, key) in enumerate(self._RECORD_KINDS):
            off, rows, cols = (int(v) for v in meta[2 + 3 * k: 5 + 3 * k])
            if rows == 0:
                arr = np.zeros(0, dt[key])
            else:
                arr = words[off: off + rows * cols].view(dt[key])
            if name.startswith("jobs"):
                jobs["wg3" if name == "jobs3" else "wgp"] = arr
            else:
                records[name] = arr
        launches: Dict[str, List[Tuple[int, int, int]]] = {}
        wgroups = []
        for kind, lv, b, e in cuts[: int(meta[38])].tolist():
            if kind == 8:
                wgroups.append((lv, b, e))
            else:
                launches.setdefault(self._CUT_KINDS[kind], []).append((lv, b, e))
        fwd, bwd = self._order(launches, int(meta[1]))
        return StepPlan(records, fwd, bwd, jobs, arena, feat_result, int(meta[0]), wgroups)

    def _wgrad_jobs(self, items: np.ndarray, wkey: np.ndarray, dw: np.ndarray, db: np.ndarray, group=None, chunk=None):
        """Sort weight-gradient items by (group,) weight and cut each run into jobs of at most
        ``wgrad_chunk`` items (one workgroup column per job).  Returns (items, jobs, group id per job)."""
        idx = np.argsort(wkey, kind="stable")
        items, wkey, dw, db = items[idx], wkey[idx], dw[idx], db[idx]
        group = np.zeros(wkey.size, np.int64) if group is None else group[idx]
        n = wkey.size
        newgrp = np.empty(n, bool)
        newgrp[0] = True
        np.not_equal(wkey[1:], wkey[:-1], out=newgrp[1:])
        gstart = np.flatnonzero(newgrp)
        gid = np.cumsum(newgrp) - 1
        pos = np.arange(n) - gstart[gid]
        chunk = chunk or self.wgrad_chunk
        jstart = np.flatnonzero(pos % chunk == 0)
        gend = np.concatenate((gstart[1:], [n]))
        jend = np.minimum(jstart + chunk, gend[gid[jstart]])
        jobs = np.zeros((jstart.size, 3), np.uint64)
        jobs[:, 0], jobs[:, 1] = dw[jstart], db[jstart]
        jobs[:, 2] = jstart.astype(np.uint64) | (jend.astype(np.uint64) << np.uint64(32))
        rec = np.ascontiguousarray(items).view(self.dt["wgrad_item"]).reshape(-1)
        return rec, jobs.view(self.dt["wgrad_job"]).reshape(-1), group[jstart]
