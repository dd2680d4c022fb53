"""The C-ABI library loads and exports every symbol include/mpengine.h declares; argument errors map to Python
exceptions.  No compute is launched (runs without a GPU)."""
import os
import re

import pytest

from gcnn_keras_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mpengine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", text)))


def test_library_built_and_loads():
    assert os.path.exists(_ffi.LIB_PATH), "run __graft_entry__.build() first"
    assert _ffi.lib().mp_version() >= 100


def test_every_declared_symbol_is_exported():
    header = _header_symbols()
    assert len(header) >= 30
    assert header == _ffi.declared_symbols(), "ctypes table and header disagree"
    lib = _ffi.lib()
    for name in header:
        assert hasattr(lib, name), name


def test_argument_errors_map_to_python_exceptions():
    lib = _ffi.lib()
    rc = lib.mp_dense_f32(None, 4, 0, None, None, 4, 0, 0.0, None, None)
    assert rc == _ffi.MP_EINVAL
    with pytest.raises(ValueError):
        _ffi.check(rc)
    assert b"mp_dense_f32" in lib.mp_last_error()
    rc = lib.mp_segment_reduce_csr_f32(9, None, 0, 1, None, None, 0, None, 0, None, None)
    assert rc == _ffi.MP_EINVAL
    # zero-sized problems are accepted without touching the device
    assert lib.mp_gather_rows_f32(None, 0, 4, None, 0, 1, _ffi.int32_array([0]), None, None) == _ffi.MP_OK
    assert lib.mp_shift_index_i64(None, 0, 2, None, None, 0, 1, None, None) == _ffi.MP_OK


def test_no_cpu_fallback():
    import torch
    from gcnn_keras_amd.layers.modules import dense_values
    with pytest.raises(_ffi.EngineError):
        dense_values(torch.zeros(2, 3), torch.zeros(3, 4), None)
