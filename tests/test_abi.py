"""CPU: the C-ABI library loads and exports every symbol include/snag_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "snag_hip.h")).read()
    return sorted(set(re.findall(r"SNAG_API\s+[\w\s\*]+?\b(snag_\w+)\s*\(", text)))


def test_header_declares_something():
    names = _declared()
    assert "snag_score_margin_fwd" in names and "snag_last_error" in names


def test_library_exports_every_declared_symbol():
    from snag_amd import _lib
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/snag_hip.h but not exported"
    assert lib.snag_version() == 1
    # every declared entry point has a ctypes signature on the Python side
    bound = set(_lib.SIGNATURES) | set(_lib._SPECIAL)
    assert set(_declared()) == bound, sorted(set(_declared()) ^ bound)


def test_argument_refusal_needs_no_gpu():
    """NULL pointers are refused before any HIP call, with a message."""
    from snag_amd import _lib
    lib = _lib.load()
    rc = lib.snag_score_margin_fwd(None, 128, 10, None, 2, 128, None, None, None, None, 4, 8, 4, 1.0, None, None, None, None)
    assert rc == -1
    assert "null" in _lib.last_error().lower()


def test_ops_refuse_cpu_tensors():
    import torch
    from snag_amd import ops, _lib
    e = torch.zeros(4, 8)
    i = torch.zeros(2, dtype=torch.int64)
    with pytest.raises(_lib.SnagHipError):
        ops.score_margin_fwd(e, e, i, i, i, None, 1.0)
