"""Static compiler for prefix CLEVR programs.

The reference interprets each program token-by-token inside a ``try/except`` and calls any
exception "invalid program" (reference: probnmn/models/nmn.py:197-238).  On the GPU that would
also swallow genuine kernel failures, so here validity is decided *before* anything is
launched, by tracking only the channel count of the two registers the interpreter has
(``output`` and ``saved_output``); the rules are SURVEY.md App. C, which were checked
against the reference interpreter case by case (tests/golden/nmn_validity.json).

A compiled program is a short list of :class:`ModuleCall` s in execution order (the
reference walks the token sequence right-to-left) over *values*:

    value 0 = FEAT   the example's stem output, ``module_channels`` channels
    value 1 = ONES   the all-ones single-channel attention that ``scene`` produces
    value k>=2       output of call ``k-2``
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

# token kinds
SKIP, SCENE, AND, OR, CMP, ATT, QUERY, REL, SAME = range(9)
KIND_NAMES = ["skip", "scene", "and", "or", "comparison", "attention", "query", "relate", "same"]

FEAT = 0
ONES = 1

_PLACEHOLDERS = {"@@PADDING@@", "@@UNKNOWN@@", "@start@", "@end@", "unique"}


def classify_token(token: str) -> int:
    """Token -> module kind, in the same test order as the reference's constructor
    (reference: probnmn/models/nmn.py:87-111) -- the order matters ("equal" before "query")."""
    if token in _PLACEHOLDERS:
        return SKIP
    if token == "scene":
        return SCENE
    if token == "intersect":
        return AND
    if token == "union":
        return OR
    if "equal" in token or token in ("less_than", "greater_than"):
        return CMP
    if "query" in token or token in ("exist", "count"):
        return QUERY
    if "relate" in token:
        return REL
    if "same" in token:
        return SAME
    return ATT


@dataclass(frozen=True)
class ModuleCall:
    kind: int  # AND .. SAME
    token: int  # program-vocabulary index (selects the weights)
    a: int  # first operand value id  (AND/OR/CMP: ``output``; others: the attention)
    b: int  # second operand value id (AND/OR/CMP: ``saved_output``; others: FEAT)
    a_channels: int
    b_channels: int
    out_channels: int


@dataclass(frozen=True)
class CompiledProgram:
    valid: bool
    calls: Tuple[ModuleCall, ...]  # empty when invalid
    result: int  # value id of the final output (FEAT for an empty program)


class ProgramCompiler:
    def __init__(self, index_to_token: Dict[int, str], module_channels: int = 128):
        self.module_channels = module_channels
        size = max(index_to_token) + 1
        self.kinds: List[int] = [SKIP] * size
        for idx, tok in index_to_token.items():
            self.kinds[idx] = classify_token(tok)
        self._cache: Dict[Tuple[int, ...], CompiledProgram] = {}
        self._bytes_cache: Dict[bytes, CompiledProgram] = {}

    def compile(self, tokens: Sequence[int]) -> CompiledProgram:
        key = tuple(int(t) for t in tokens)
        hit = self._cache.get(key)
        if hit is None:
            hit = self._compile(key)
            self._cache[key] = hit
        return hit

    def compile_batch(self, programs) -> List[CompiledProgram]:
        """``programs``: (B, T) integer array (numpy), already on the host."""
        import numpy as np

        arr = np.ascontiguousarray(programs, dtype=np.int64)
        out = []
        cache = self._bytes_cache
        for row in arr:
            key = row.tobytes()
            hit = cache.get(key)
            if hit is None:
                hit = self.compile(row.tolist())
                cache[key] = hit
            out.append(hit)
        return out

    # ---------------------------------------------------------------------------------
    def _compile(self, tokens: Tuple[int, ...]) -> CompiledProgram:
        D = self.module_channels
        invalid = CompiledProgram(False, (), FEAT)
        out, out_c = FEAT, D
        saved: Optional[int] = None
        saved_c = 0
        calls: List[ModuleCall] = []
        nkinds = len(self.kinds)
        for tok in reversed(tokens):
            if tok < 0 or tok >= nkinds:
                return invalid  # the reference's vocabulary lookup would raise KeyError
            kind = self.kinds[tok]
            if kind == SKIP:
                continue
            if kind == SCENE:
                saved, saved_c = out, out_c
                out, out_c = ONES, 1
                continue
            if kind in (AND, OR):
                if saved is None:
                    return invalid
                oc = max(out_c, saved_c)  # torch.min/max broadcast 1 <-> D channels
                calls.append(ModuleCall(kind, tok, out, saved, out_c, saved_c, oc))
            elif kind == CMP:
                if saved is None or out_c != D or saved_c != D:
                    return invalid
                calls.append(ModuleCall(kind, tok, out, saved, D, D, D))
                oc = D
            else:  # ATT / QUERY / REL / SAME take (FEAT, attention)
                if out_c != 1:
                    return invalid
                oc = D if kind == QUERY else 1
                calls.append(ModuleCall(kind, tok, out, FEAT, 1, D, oc))
            out, out_c = len(calls) + 1, oc
        if out_c != D:
            return invalid
        return CompiledProgram(True, tuple(calls), out)
