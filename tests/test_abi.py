"""The C-ABI library loads without a GPU and exports exactly what include/probnmn_hip.h declares;
the numpy record layouts used by the host code match the C structs byte for byte."""
import ctypes
import os
import re
import subprocess
import tempfile

from probnmn import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "probnmn_hip.h")


def declared_functions():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"^(?:int|int64_t)\s+(pnmn_\w+)\s*\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 18
    assert sorted(_hip.SIGNATURES) == names  # binding table and header agree
    handle = _hip.lib()
    for n in names:
        assert hasattr(handle, n), n
    assert handle.pnmn_abi_version() == _hip.ABI_VERSION == 12


def test_launch_trace_without_launches_is_empty():
    """pnmn_launch_trace_begin / _end around nothing: zero entries, no device needed; a second _end is harmless."""
    handle = _hip.lib()
    import numpy as np
    out, n = np.zeros(4, _hip.LAUNCH_TIMING), np.full(1, -1, np.int32)
    assert handle.pnmn_launch_trace_begin() == 0
    assert handle.pnmn_launch_trace_end(out.ctypes.data, 4, n.ctypes.data) == 0 and n[0] == 0
    assert handle.pnmn_launch_trace_end(out.ctypes.data, 4, n.ctypes.data) == 0 and n[0] == 0
    assert handle.pnmn_launch_trace_end(0, 0, 0) == _hip.EINVAL


def test_record_layouts_match_c_structs():
    """Compile a tiny C program against the header and compare sizeof/offsetof with numpy."""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "probnmn_hip.h"', "int main(){"]
    for cname, (dtype, size) in _hip.ITEM_SIZES.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for field in dtype.names:
            cfield = {"in": "in"}.get(field, field)
            lines.append('printf(" %s=%%zu", offsetof(%s, %s));' % (field, cname, cfield))
        lines.append('printf("\\n");')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write("\n".join(lines))
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().strip().splitlines()
    seen = {}
    for line in out:
        parts = line.split()
        seen[parts[0]] = (int(parts[1]), {p.split("=")[0]: int(p.split("=")[1]) for p in parts[2:]})
    for cname, (dtype, size) in _hip.ITEM_SIZES.items():
        csize, offsets = seen[cname]
        assert csize == size == dtype.itemsize, cname
        for field in dtype.names:
            assert dtype.fields[field][1] == offsets[field], (cname, field)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libprobnmn_hip.so")
    try:
        _hip.lib()
    except _hip.HipLibraryError as e:
        assert "no" in str(e) and "fallback" in str(e)
    else:
        raise AssertionError("expected HipLibraryError")
