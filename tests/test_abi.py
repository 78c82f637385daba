"""The C-ABI library loads without a GPU and exports every symbol include/vo_hip.h declares."""
import ctypes as C
import os
import re

import pytest

from visual_odometry_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", src)))


def test_header_and_loader_agree():
    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"libvo_hip.so does not export {name}"


def test_strerror_and_version():
    lib = _lib.load()
    assert lib.vo_version() >= 100
    assert lib.vo_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.vo_strerror(-1)


def test_no_device_is_a_loud_error():
    """Without a HIP device the product refuses to work (there is no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.vo_ctx_create(0, C.byref(h)) == -1
    with pytest.raises(_lib.VoError):
        _lib.Context(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package includes, imports or loads it
    (comments may cite its normative definitions)."""
    pkg = os.path.join(ROOT, "visual_odometry_amd")
    bad = re.compile(r'#include\s+"[^"]*oracle|^\s*(import|from)\s+oracle|libvo_oracle|dlopen[^\n]*oracle', re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), f"{f} references the test oracle"
