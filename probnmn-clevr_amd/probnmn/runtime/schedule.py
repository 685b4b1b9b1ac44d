"""Batch scheduler: compiled programs -> level-ordered kernel work lists.

The reference runs one example at a time, one torch op at a time (reference:
probnmn/models/nmn.py:197-238) -- thousands of tiny launches per step.  Here every module call of
every example in the batch is expanded into *primitives* (conv3x3, projection, 1-channel head,
Same, And/Or), each primitive gets the dependency level at which its inputs are ready, and all
primitives of one (level, kind) become ONE grouped kernel launch whose work list says, per item,
which example's buffers and which token's weights to use.

Host cost matters (the GPU step is milliseconds), so nothing here loops over examples:
programs are grouped by *structure* (their calls' kinds and wiring, tokens ignored); a structure's
primitive list is built once and cached (:class:`Template`), and the per-batch records are
produced with numpy broadcasting over all examples that share the structure -- weights come from
per-token offset tables, buffers from per-example base addresses.

Value placement: each example owns a contiguous block of the activation arena laid out by its
template; the gradient arena mirrors it offset-for-offset, so ``grad(x) = x - act_base + grad_base``.
The value a program returns is placed directly in the classifier's input row (``FINAL``).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import program_compiler as pc

HW_ALIGN = 64  # floats; keeps every slot 256-byte aligned

# operand location kinds
L_SLOT, L_FEAT, L_ONES, L_FINAL = 0, 1, 2, 3

RELATE_DILATIONS = (1, 2, 4, 8, 1)  # reference nmn_modules.py:146-150


def _align(n: int) -> int:
    return (n + HW_ALIGN - 1) // HW_ALIGN * HW_ALIGN


@dataclass
class _Prim:
    kind: str  # "conv" | "proj" | "dot" | "same" | "minmax"
    level: int
    call: int  # index of the module call (selects the token -> weights)
    widx: int = 0  # which weight of the module (0 = projection, 1..5 = conv1..conv5)
    dil: int = 1
    a: Tuple[int, int] = (L_ONES, 0)  # main input (conv/dot: feature map; same/minmax: a)
    b: Tuple[int, int] = (L_ONES, 0)  # second input (proj: in2; conv: mask; same: attn; minmax: b)
    out: Tuple[int, int] = (L_SLOT, 0)
    a_ch: int = 0
    b_ch: int = 0
    is_max: int = 0
    masked: bool = False  # conv whose input is FEAT * attention (needs mask backward)


@dataclass
class Template:
    """Primitive list of one program structure; offsets are floats relative to the example's
    arena block."""

    n_calls: int
    size: int  # arena floats per example
    result: Tuple[int, int]  # location of the returned value (L_FINAL, 0) or (L_FEAT, 0)
    prims: List[_Prim]
    depth: int
    n_masked: int = 0  # convs whose input is FEAT * attention (each needs one scratch map in backward)


def structure_key(prog: pc.CompiledProgram) -> Tuple:
    return tuple((c.kind, c.a, c.b) for c in prog.calls) + (prog.result,)


def build_template(prog: pc.CompiledProgram, hw: int, channels: int) -> Template:
    big, small = _align(hw * channels), _align(hw)
    calls = prog.calls
    # liveness: only calls that reach the result are executed (results are unaffected: a dead
    # call's output feeds nothing; validity was already decided on the full program)
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
    prims: List[_Prim] = []

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
            prims.append(
                _Prim("minmax", level, ci, a=loc[c.a], b=loc[c.b], out=out, a_ch=c.a_channels,
                      b_ch=c.b_channels, is_max=int(c.kind == pc.OR))
            )
        elif c.kind == pc.SAME:
            level = lvl[c.a] + 1
            prims.append(_Prim("same", level, ci, a=(L_FEAT, 0), b=loc[c.a], out=out))
        elif c.kind == pc.CMP:
            level = max(lvl[c.a], lvl[c.b]) + 1
            t0 = (L_SLOT, alloc(big))
            t1 = (L_SLOT, alloc(big))
            prims.append(_Prim("proj", level, ci, widx=0, a=loc[c.a], b=loc[c.b], out=t0))
            prims.append(_Prim("conv", level + 1, ci, widx=1, a=t0, out=t1))
            prims.append(_Prim("conv", level + 2, ci, widx=2, a=t1, out=out))
            level += 2
        else:  # ATT / QUERY / REL
            nconv = 5 if c.kind == pc.REL else 2
            dils = RELATE_DILATIONS if c.kind == pc.REL else (1, 1)
            level = lvl[c.a]
            src = (L_FEAT, 0)
            for k in range(nconv):
                level += 1
                last_is_out = (k == nconv - 1) and c.kind == pc.QUERY
                dst = out if last_is_out else (L_SLOT, alloc(big))
                p = _Prim("conv", level, ci, widx=k + 1, dil=dils[k], a=src, out=dst)
                if k == 0:
                    p.b = loc[c.a]  # attention mask (L_ONES -> no multiply)
                    p.masked = True
                prims.append(p)
                src = dst
            if c.kind != pc.QUERY:
                level += 1
                prims.append(_Prim("dot", level, ci, a=src, out=out))
        loc[vid] = out
        lvl[vid] = level

    result = loc[prog.result] if prog.result >= 2 else (L_FEAT, 0)
    depth = max([p.level for p in prims], default=0)
    return Template(len(calls), cursor, result, prims, depth, sum(1 for p in prims if p.masked))


# ------------------------------------------------------------------------------------------------
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


@dataclass
class Launch:
    kind: str
    level: int
    begin: int
    end: int


@dataclass
class StepPlan:
    records: Dict[str, np.ndarray]  # kind -> record array (sorted by level)
    forward: List[Launch]
    backward: List[List[Launch]]  # phases per level (reverse level order)
    wgrad_jobs: Dict[str, np.ndarray]
    arena_floats: int
    feat_result_examples: np.ndarray  # examples whose program returns FEAT itself
    n_prims: int


class BatchScheduler:
    def __init__(self, hw: int, channels: int, tables: WeightTables, record_dtypes: Dict[str, np.dtype],
                 wgrad_chunk: int = 8):
        self.hw = hw
        self.channels = channels
        self.tables = tables
        self.dt = record_dtypes
        self.wgrad_chunk = wgrad_chunk
        self._templates: Dict[Tuple, Template] = {}

    def arena_floats(self, programs: Sequence[pc.CompiledProgram]) -> int:
        """Activation-arena size (floats) the batch needs; the gradient arena mirrors it."""
        big = _align(self.hw * self.channels)
        total = 0
        for prog in programs:
            if prog.valid:
                t = self.template(prog)
                total += t.size + t.n_masked * big
        return total

    def template(self, prog: pc.CompiledProgram) -> Template:
        key = structure_key(prog)
        t = self._templates.get(key)
        if t is None:
            t = build_template(prog, self.hw, self.channels)
            self._templates[key] = t
        return t

    # --------------------------------------------------------------------------------------------
    def plan(self, programs: Sequence[pc.CompiledProgram], buf: Buffers) -> StepPlan:
        hw, C = self.hw, self.channels
        map_bytes = hw * C * 4
        tb = self.tables

        # group valid examples by structure
        groups: Dict[Tuple, List[int]] = {}
        for e, prog in enumerate(programs):
            if prog.valid:
                groups.setdefault(structure_key(prog), []).append(e)

        parts: Dict[str, List[np.ndarray]] = {k: [] for k in
                                              ("conv", "proj", "dot", "same", "minmax", "dgrad", "pdgrad",
                                               "maskbwd", "wg3", "wgp")}
        levels: Dict[str, List[np.ndarray]] = {k: [] for k in parts}
        wkeys: Dict[str, List[np.ndarray]] = {"wg3": [], "wgp": []}
        feat_result: List[int] = []
        cursor = 0
        n_prims = 0

        for key, ex in groups.items():
            prog0 = programs[ex[0]]
            t = self.template(prog0)
            E = np.asarray(ex, dtype=np.int64)
            n = len(ex)
            tokens = np.asarray([[c.token for c in programs[e].calls] for e in ex], dtype=np.int64).reshape(n, -1)
            base = cursor + np.arange(n, dtype=np.int64) * t.size  # floats
            cursor += n * t.size
            if t.result[0] == L_FEAT:
                feat_result.extend(ex)
            n_prims += n * len(t.prims)

            def addr(loc, grad=False):
                kind, off = loc
                if kind == L_SLOT:
                    return (buf.gact if grad else buf.act) + (base + off) * 4
                if kind == L_FEAT:
                    return (buf.gfeat if grad else buf.feat) + E * map_bytes
                if kind == L_FINAL:
                    return (buf.gfinal if grad else buf.final) + E * map_bytes
                return np.zeros(n, dtype=np.int64)  # L_ONES

            def addr_or_ones(loc):
                if loc[0] == L_ONES:
                    return np.full(n, buf.ones, dtype=np.int64)
                return addr(loc)

            for p in t.prims:
                tok = tokens[:, p.call]
                lv = np.full(n, p.level, dtype=np.int32)
                if p.kind in ("conv", "proj"):
                    name = "conv" if p.kind == "conv" else "proj"
                    r = np.zeros(n, self.dt["conv"])
                    r["in"] = addr(p.a)
                    if p.kind == "proj":
                        r["in2"] = addr(p.b)
                    elif p.masked and p.b[0] != L_ONES:
                        r["mask"] = addr(p.b)
                    r["weight"] = buf.params + tb.w3[tok, p.widx] * 4
                    r["bias"] = buf.params + tb.b3[tok, p.widx] * 4
                    r["out"] = addr(p.out)
                    r["dilation"] = p.dil
                    parts[name].append(r)
                    levels[name].append(lv)
                    # ---- backward: dgrad ----
                    if p.kind == "conv":
                        d = np.zeros(n, self.dt["conv"])
                        d["in"] = addr(p.out, grad=True)
                        d["gate"] = addr(p.out)
                        d["weight"] = buf.wt + tb.wt3[tok, p.widx] * 4
                        d["dilation"] = p.dil
                        if p.masked:
                            # gradient wrt (FEAT * attn): private scratch map, assigned below
                            d["out"] = 0
                        else:
                            d["out"] = addr(p.a, grad=True)
                        parts["dgrad"].append(d)
                        levels["dgrad"].append(lv)
                        w = np.zeros(n, self.dt["wgrad_item"])
                        w["x"] = addr(p.a)
                        if p.masked and p.b[0] != L_ONES:
                            w["xmask"] = addr(p.b)
                        w["dy"] = addr(p.out, grad=True)
                        w["gate"] = addr(p.out)
                        w["dilation"] = p.dil
                        parts["wg3"].append(w)
                        levels["wg3"].append(lv)
                        wkeys["wg3"].append(tok * 8 + p.widx)
                    else:
                        for half, operand in ((0, p.a), (1, p.b)):
                            d = np.zeros(n, self.dt["conv"])
                            d["in"] = addr(p.out, grad=True)
                            d["gate"] = addr(p.out)
                            d["weight"] = buf.wt + (tb.wt3[tok, 0] + half * C * C) * 4
                            d["out"] = addr(operand, grad=True)
                            d["flags"] = 1  # accumulate into the operand's gradient
                            parts["pdgrad"].append(d)
                            levels["pdgrad"].append(lv * 2 + half)  # the two halves never share a launch
                        w = np.zeros(n, self.dt["wgrad_item"])
                        w["x"] = addr(p.a)
                        w["x2"] = addr(p.b)
                        w["dy"] = addr(p.out, grad=True)
                        w["gate"] = addr(p.out)
                        parts["wgp"].append(w)
                        levels["wgp"].append(lv)
                        wkeys["wgp"].append(tok)
                elif p.kind == "dot":
                    r = np.zeros(n, self.dt["dot"])
                    r["in"] = addr(p.a)
                    r["w"] = buf.params + tb.dotw[tok] * 4
                    r["b"] = buf.params + tb.dotb[tok] * 4
                    r["out"] = addr(p.out)
                    r["dout"] = addr(p.out, grad=True)
                    r["din"] = addr(p.a, grad=True)
                    r["dw"] = buf.grads + tb.dotw[tok] * 4
                    r["db"] = buf.grads + tb.dotb[tok] * 4
                    parts["dot"].append(r)
                    levels["dot"].append(lv)
                elif p.kind == "same":
                    r = np.zeros(n, self.dt["same"])
                    r["feats"] = addr(p.a)
                    r["attn"] = addr_or_ones(p.b)
                    r["w"] = buf.params + tb.dotw[tok] * 4
                    r["b"] = buf.params + tb.dotb[tok] * 4
                    r["out"] = addr(p.out)
                    r["dout"] = addr(p.out, grad=True)
                    r["dfeats"] = addr(p.a, grad=True)
                    r["dattn"] = addr(p.b, grad=True)  # 0 for the all-ones attention
                    r["dw"] = buf.grads + tb.dotw[tok] * 4
                    r["db"] = buf.grads + tb.dotb[tok] * 4
                    parts["same"].append(r)
                    levels["same"].append(lv)
                else:  # minmax
                    r = np.zeros(n, self.dt["minmax"])
                    r["a"] = addr_or_ones(p.a)
                    r["b"] = addr_or_ones(p.b)
                    r["out"] = addr(p.out)
                    r["dout"] = addr(p.out, grad=True)
                    r["da"] = addr(p.a, grad=True)
                    r["db"] = addr(p.b, grad=True)
                    r["a_channels"] = p.a_ch
                    r["b_channels"] = p.b_ch
                    r["is_max"] = p.is_max
                    parts["minmax"].append(r)
                    levels["minmax"].append(lv)

        # masked convs: private dx scratch (one map per masked conv, after the example blocks)
        records: Dict[str, np.ndarray] = {}
        order: Dict[str, np.ndarray] = {}
        for k in parts:
            if parts[k]:
                rec = np.concatenate(parts[k])
                lv = np.concatenate(levels[k])
            else:
                proto = {"conv": "conv", "proj": "conv", "dgrad": "conv", "pdgrad": "conv", "dot": "dot",
                         "same": "same", "minmax": "minmax", "maskbwd": "maskbwd", "wg3": "wgrad_item",
                         "wgp": "wgrad_item"}[k]
                rec = np.zeros(0, self.dt[proto])
                lv = np.zeros(0, np.int32)
            records[k] = rec
            order[k] = lv

        # scratch maps + mask-backward records for masked convs
        dg, dgl = records["dgrad"], order["dgrad"]
        masked = np.nonzero(dg["out"] == 0)[0]
        big = _align(hw * C)
        if masked.size:
            scratch = cursor + np.arange(masked.size, dtype=np.int64) * big
            cursor += masked.size * big
            dg["out"][masked] = buf.gact + scratch * 4
            # the matching forward conv records are in the same order as the dgrad records
            fw = records["conv"][masked]
            mb = np.zeros(masked.size, self.dt["maskbwd"])
            mb["dx"] = dg["out"][masked]
            mb["feats"] = fw["in"]
            mb["attn"] = fw["mask"]
            mb["dfeats"] = fw["in"] - buf.feat + buf.gfeat
            has_attn = fw["mask"] != 0
            mb["dattn"][has_attn] = fw["mask"][has_attn] - buf.act + buf.gact
            records["maskbwd"] = mb
            order["maskbwd"] = dgl[masked]
        # sort every kind by level and cut launches
        launches: Dict[str, List[Launch]] = {}
        for k, rec in records.items():
            lv = order[k]
            if k in ("wg3", "wgp"):
                continue
            idx = np.argsort(lv, kind="stable")
            records[k] = rec[idx]
            lv = lv[idx]
            cuts = np.flatnonzero(np.diff(lv)) + 1 if lv.size else np.zeros(0, np.int64)
            bounds = np.concatenate(([0], cuts, [lv.size])).astype(np.int64)
            launches[k] = [Launch(k, int(lv[b]), int(b), int(e)) for b, e in zip(bounds[:-1], bounds[1:]) if e > b]

        depth = max([l.level for ls in launches.values() for l in ls if l.kind != "pdgrad"], default=0)
        fwd: List[Launch] = []
        by_level: Dict[int, Dict[str, Launch]] = {}
        for k in ("conv", "proj", "dot", "same", "minmax"):
            for l in launches.get(k, []):
                by_level.setdefault(l.level, {})[k] = l
        for level in range(1, depth + 1):
            for k in ("minmax", "same", "dot", "proj", "conv"):
                if k in by_level.get(level, {}):
                    fwd.append(by_level[level][k])

        bwd_by_level: Dict[int, Dict[str, List[Launch]]] = {}
        for k in ("dgrad", "maskbwd"):
            for l in launches.get(k, []):
                bwd_by_level.setdefault(l.level, {}).setdefault(k, []).append(l)
        for l in launches.get("pdgrad", []):
            bwd_by_level.setdefault(l.level // 2, {}).setdefault("pdgrad", []).append(l)
        bwd: List[List[Launch]] = []
        for level in range(depth, 0, -1):
            phase: List[Launch] = []
            lv_f = by_level.get(level, {})
            lv_b = bwd_by_level.get(level, {})
            for k in ("minmax", "same", "dot"):  # reuse the forward records (they carry bwd fields)
                if k in lv_f:
                    phase.append(Launch(k + "_bwd", level, lv_f[k].begin, lv_f[k].end))
            phase.extend(lv_b.get("pdgrad", []))
            phase.extend(lv_b.get("dgrad", []))
            phase.extend(lv_b.get("maskbwd", []))
            bwd.append(phase)

        # weight-gradient jobs: items sorted by weight, cut into chunks
        jobs: Dict[str, np.ndarray] = {}
        for k, col in (("wg3", None), ("wgp", None)):
            rec = records[k]
            if rec.size == 0:
                jobs[k] = np.zeros(0, self.dt["wgrad_job"])
                continue
            wk = np.concatenate(wkeys[k])
            idx = np.argsort(wk, kind="stable")
            records[k] = rec[idx]
            wk = wk[idx]
            starts = np.concatenate(([0], np.flatnonzero(np.diff(wk)) + 1))
            ends = np.concatenate((starts[1:], [wk.size]))
            jb, je, jw = [], [], []
            for s, e in zip(starts, ends):
                cs = np.arange(s, e, self.wgrad_chunk)
                jb.append(cs)
                je.append(np.minimum(cs + self.wgrad_chunk, e))
                jw.append(np.full(cs.size, wk[s]))
            jb, je, jw = np.concatenate(jb), np.concatenate(je), np.concatenate(jw)
            j = np.zeros(jb.size, self.dt["wgrad_job"])
            if k == "wg3":
                tok, widx = jw // 8, jw % 8
            else:
                tok, widx = jw, np.zeros_like(jw)
            j["dw"] = buf.grads + tb.w3[tok, widx] * 4
            j["dbias"] = buf.grads + tb.b3[tok, widx] * 4
            j["item_begin"] = jb
            j["item_end"] = je
            jobs[k] = j

        return StepPlan(records, fwd, bwd, jobs, cursor, np.asarray(feat_result, dtype=np.int64), n_prims)
