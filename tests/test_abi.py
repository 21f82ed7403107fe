"""The C-ABI library: it loads, exports every symbol include/glvoc.h declares, the ctypes
prototypes cover the same set, and argument validation that needs no GPU behaves."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "glvoc.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glv_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from tts_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("glv_plan_create", "glv_plan_destroy", "glv_plan_set_inverse_basis", "glv_mel_invert",
                 "glv_griffinlim", "glv_peak_normalize", "glv_stft", "glv_istft", "glv_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"libglvoc.so does not export {name}"


def test_ctypes_prototypes_cover_the_header():
    from tts_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_version_and_validation_without_gpu(lib):
    from tts_amd import _lib
    assert lib.glv_version() >= 100
    handle = ctypes.c_void_p()
    # unsupported transform sizes are rejected before any device call
    assert lib.glv_plan_create(ctypes.byref(handle), 1536, 256, 1536, 80, 16, 1, 0) == -2
    assert b"n_fft" in lib.glv_last_error()
    assert lib.glv_plan_create(ctypes.byref(handle), 1024, 200, 1024, 80, 16, 1, 0) == -2
    assert lib.glv_plan_create(ctypes.byref(handle), 1024, 256, 4096, 80, 16, 1, 0) == -1
    assert lib.glv_plan_create(ctypes.byref(handle), 1024, 256, 1024, 80, 1, 1, 0) == -1
    assert lib.glv_plan_create(None, 1024, 256, 1024, 80, 16, 1, 0) == -1
    assert handle.value is None
    with pytest.raises(_lib.GlvError):
        _lib.check(-1)
    # null plans are refused, not dereferenced
    assert lib.glv_plan_set_option(None, b"timing", 1) == -1
    assert lib.glv_griffinlim(None, None, None, 0, None, 1, 16, 1, 0.99, None, 0, None, None) == -1
    assert lib.glv_plan_workspace_bytes(None) == 0
    lib.glv_plan_destroy(None)
