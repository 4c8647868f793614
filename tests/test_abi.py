"""The C-ABI library loads (no GPU needed) and exports every symbol include/spx.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "spx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from spx import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libspx.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "python binding and header disagree"


def test_abi_version_and_strerror():
    from spx import _lib
    lib = _lib.load()
    assert lib.spx_abi_version() == _lib.SPX_ABI_VERSION
    assert lib.spx_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6):
        assert len(lib.spx_strerror(code)) > 4
    assert b"unknown" in lib.spx_strerror(-99)


def test_argument_validation_without_gpu():
    """Host-side argument checks run before any launch, so they can be exercised on a CPU-only box."""
    from spx import _lib
    lib = _lib.load()
    i3 = _lib.i3
    assert lib.spx_subm_rulebook(None, 10, None, 1, i3([1, 1, 1]), i3([3, 3, 3]), i3([1, 1, 1]), None, 10, None, 0, None,
                                 None, 0, None) == -1
    assert lib.spx_pack_weight(None, 16, 27, 16, 0, None, None) == -1
    assert lib.spx_conv_out_cap(1000, 2, i3([21, 800, 704]), i3([3, 3, 3]), i3([2, 2, 2])) == 8000
    assert lib.spx_conv_out_cap(1000, 1, i3([2, 3, 4]), i3([3, 3, 3]), i3([1, 1, 1])) == 24
    assert lib.spx_subm_rulebook_ws_bytes(16000) >= 32768 * 12
    assert lib.spx_conv_wgrad_ws_bytes(64, 64, 27, 100000) >= 27 * 64 * 64 * 4


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the operator layer raises instead of computing on the host."""
    import pytest
    import torch
    from spx import _lib, ops
    with pytest.raises(_lib.SpxError):
        ops.subm_rulebook(torch.zeros((4, 4), dtype=torch.int32), 1, [4, 4, 4], (3, 3, 3))
