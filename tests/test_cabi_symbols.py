"""CPU: the C-ABI shared library loads and exports every symbol include/mvsgi.h declares;
argument validation rejects bad calls before anything touches a GPU."""
import ctypes
import os
import re

import pytest

from mvs_gi_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mvsgi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvsgi_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared_symbols()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mvsgi.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes SIGNATURES out of sync with the header"


def test_abi_version(lib):
    assert lib.mvsgi_abi_version() == _lib.ABI_VERSION


def test_argument_validation_sets_last_error(lib):
    # null pointers / bad dims are rejected on the host, nothing is enqueued
    rc = lib.mvsgi_conv3d_f32(None, None, None, None, None, None, None, 1, 16, 4, 4, 4, 16, 1, 0.01, 0, None)
    assert rc != 0 and b"null pointer" in lib.mvsgi_last_error()
    rc = lib.mvsgi_softargmin_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None,
                                  1, 8, 4, 4, 3, None)
    assert rc != 0 and b"scale" in lib.mvsgi_last_error()
    rc = lib.mvsgi_conv3d_pack_weights_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), 24, 16, None)
    assert rc != 0 and b"multiples of 16" in lib.mvsgi_last_error()
    rc = lib.mvsgi_sweep_std_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 0, ctypes.c_void_p(16),
                                 ctypes.c_void_p(16), 1, 9, 16, 8, 8, 8, 8, 4, 4, 4, None)
    assert rc != 0 and b"num_cams" in lib.mvsgi_last_error()
    assert lib.mvsgi_conv3d_packed_weight_floats(32, 16) == 27 * 32 * 16


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MvsgiLibraryMissing):
        _lib.load()
