"""CPU: the C-ABI library loads, exports what include/rvm_mi355x.h declares, and agrees with the
Python spec on the state_dict contract.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from robustvideomatting_amd import native, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "rvm_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rvm_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    L = native.lib()
    syms = _header_symbols()
    assert sorted(native.SYMBOLS) == syms
    for s in syms:
        assert getattr(L, s) is not None


def test_version_and_error_string():
    L = native.lib()
    assert b"gfx950" in L.rvm_version()
    assert isinstance(L.rvm_last_error(), bytes)


def test_create_rejects_bad_arguments():
    L = native.lib()
    h = ctypes.c_void_p()
    assert L.rvm_create(7, 0, 0, 0, ctypes.byref(h)) != 0
    assert b"variant" in L.rvm_last_error()
    assert L.rvm_create(0, 0, 9, 0, ctypes.byref(h)) != 0
    assert b"dtype" in L.rvm_last_error()
    assert not h.value


def test_null_context_calls_fail_cleanly():
    L = native.lib()
    assert L.rvm_finalize(None) != 0
    assert L.rvm_plan(None, 1, 8, 8, 1.0) != 0
    assert L.rvm_param_count(None) == 0
    L.rvm_destroy(None)


def _gpu_present():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_gpu_present(), reason="CPU-only check of the no-device error path")
def test_create_without_device_is_an_error_not_a_crash():
    L = native.lib()
    h = ctypes.c_void_p()
    rc = L.rvm_create(0, 0, 1, 0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert len(L.rvm_last_error()) > 0
