"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol."""
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "unimp_hip.h")).read()
    return sorted(set(re.findall(r"\b(unimp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    from unimp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), s
    assert set(_lib.declared_symbols()) == set(syms)
    assert L.unimp_abi_version() == 1


def test_no_cpu_fallback():
    import torch
    from unimp_amd import ops, _lib
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(_lib.UnimpHipError):
        ops.gemm(a, a)


def test_product_does_not_import_oracle():
    import subprocess, sys
    bad = subprocess.run(["grep", "-rlE", r"^\s*(from|import)\s+oracle", os.path.join(ROOT, "unimp_amd")],
                         capture_output=True, text=True).stdout.split()
    assert bad == [], bad


def test_ctypes_mirrors_have_the_c_struct_sizes():
    """the ctypes Structures of the binding (and of INTEGRATION.md's stub) against sizeof() inside the library."""
    import ctypes as C
    from unimp_amd import _lib
    from unimp_amd.data import _ImageDesc
    lib = C.CDLL(_lib._LIB_PATH) if hasattr(_lib, "_LIB_PATH") else _lib.lib()
    lib.unimp_struct_size.restype = C.c_int
    assert lib.unimp_struct_size(0) == C.sizeof(_lib.GemmDesc)
    assert lib.unimp_struct_size(1) == C.sizeof(_lib.AttnDesc)
    assert lib.unimp_struct_size(2) == C.sizeof(_ImageDesc)
    assert lib.unimp_struct_size(9) == -1
