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


TOKEN_ROW = 48  # calls per program of the fixed-width token rows the scheduler stacks (longer programs: a copy loop)


class CompiledProgram:
    """valid / calls / result of one program.  ``calls`` (a tuple of :class:`ModuleCall`) is built on
    first use when the program came out of the batch compiler as an int32 table ``[n_calls, 7]``
    (kind, token, a, b, a_channels, b_channels, out_channels) -- the scheduler only needs the table."""

    __slots__ = ("valid", "result", "_raw", "_calls", "_template_id", "_template_owner", "_tokens", "_tokens_row", "_skey")

    def __init__(self, valid: bool, calls: Tuple[ModuleCall, ...] = (), result: int = FEAT, raw=None):
        self.valid = valid
        self.result = result  # value id of the final output (FEAT for an empty program)
        self._raw = raw
        self._calls = None if raw is not None else tuple(calls)  # empty when invalid
        self._template_id = None
        self._template_owner = None
        self._tokens = None
        self._tokens_row = None
        self._skey = None  # bytes of the call table with the token column zeroed (the scheduler's template key)

    @property
    def calls(self) -> Tuple[ModuleCall, ...]:
        if self._calls is None:
            self._calls = tuple(ModuleCall(*row) for row in self._raw.tolist())
        return self._calls

    def table(self):
        """int32 [n_calls, 7], the calls as rows."""
        if self._raw is None:
            import numpy as np

            self._raw = np.asarray([[c.kind, c.token, c.a, c.b, c.a_channels, c.b_channels, c.out_channels]
                                    for c in self._calls], dtype=np.int32).reshape(-1, 7)
        return self._raw

    def __eq__(self, other):
        return (isinstance(other, CompiledProgram) and self.valid == other.valid and self.result == other.result
                and self.calls == other.calls)

    def __repr__(self):
        return "CompiledProgram(valid=%r, calls=%r, result=%r)" % (self.valid, self.calls, self.result)


class ProgramCompiler:
    def __init__(self, index_to_token: Dict[int, str], module_channels: int = 128):
        self.module_channels = module_channels
        size = max(index_to_token) + 1
        self.kinds: List[int] = [SKIP] * size
        for idx, tok in index_to_token.items():
            self.kinds[idx] = classify_token(tok)
        self._cache: Dict[Tuple[int, ...], CompiledProgram] = {}
        self._bytes_cache: Dict[bytes, CompiledProgram] = {}
        self._kinds_array = None
        self._invalid = CompiledProgram(False, (), FEAT)

    def compile(self, tokens: Sequence[int]) -> CompiledProgram:
        key = tuple(int(t) for t in tokens)
        hit = self._cache.get(key)
        if hit is None:
            hit = self._compile(key)
            self._cache[key] = hit
        return hit

    def compile_batch(self, programs) -> List[CompiledProgram]:
        """``programs``: (B, T) integer array (numpy), already on the host.  Programs seen before
        come from a cache keyed by their bytes; the others are compiled together by the library's
        host routine (``pnmn_compile_programs``, the same rules as ``_compile``)."""
        import numpy as np

        from probnmn import _hip

        arr = np.ascontiguousarray(programs, dtype=np.int64)
        if arr.ndim != 2:
            raise ValueError("programs must be (batch, length), got shape %s" % (arr.shape,))
        cache = self._bytes_cache
        if len(cache) > 500000:  # bounded: sampled programs keep arriving for the whole training run
            cache.clear()
        keys = [row.tobytes() for row in arr]
        out = [cache.get(k) for k in keys]
        miss = [i for i, hit in enumerate(out) if hit is None]
        if miss:
            sub = np.ascontiguousarray(arr[miss])
            n, length = sub.shape
            if self._kinds_array is None:
                self._kinds_array = np.asarray(self.kinds, dtype=np.int32)
            valid = np.empty(n, np.uint8)
            n_calls = np.empty(n, np.int32)
            calls = np.empty((n, max(length, 1), 7), np.int32)
            result = np.empty(n, np.int32)
            _hip.check(_hip.lib().pnmn_compile_programs(
                sub.ctypes.data, n, length, self._kinds_array.ctypes.data, self._kinds_array.size, self.module_channels,
                valid.ctypes.data, n_calls.ctypes.data, calls.ctypes.data, result.ctypes.data), "compile_programs")
            valid_l, n_l, res_l = valid.tolist(), n_calls.tolist(), result.tolist()
            # what the scheduler wants of a new program, for the whole batch at once: the structure key (call table
            # without the tokens) and the tokens as a zero-padded int64 row -- per program these were a fancy-indexed
            # copy, an astype and a zeros() each, 4 us x ~35 new programs per step on the way to the first launch
            width = max(length, TOKEN_ROW)
            used = np.arange(max(length, 1))[None, :] < n_calls[:, None]
            struct = np.where(used[:, :, None], calls, 0)
            rows = np.zeros((n, width), np.int64)
            rows[:, : struct.shape[1]] = struct[:, :, 1]
            struct[:, :, 1] = 0
            for j, i in enumerate(miss):
                key = keys[i]
                hit = cache.get(key)  # (the same new program may occur several times in the batch)
                if hit is None:
                    if valid_l[j]:
                        nj = n_l[j]
                        hit = CompiledProgram(True, result=res_l[j], raw=calls[j, :nj].copy())
                        hit._skey = struct[j, :nj].tobytes()
                        hit._tokens = rows[j, :nj]
                        if nj <= TOKEN_ROW:
                            hit._tokens_row = rows[j, :TOKEN_ROW]
                    else:
                        hit = self._invalid
                    cache[key] = hit
                out[i] = hit
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
