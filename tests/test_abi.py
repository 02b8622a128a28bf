"""CPU-only: the C-ABI libraries load and export every symbol include/*.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from approximatenn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_backend_exports(prec):
    _lib.build()
    lib = _lib.load(prec)
    assert lib.annhip_precision().decode() == prec
    for sym in _lib.EXPORTED:
        assert hasattr(lib, sym), sym
    disp = C.CDLL(os.path.join(_lib.CSRC, "libann_dispatch_%s.so" % prec))
    for sym in _lib.DISPATCH_EXPORTED:
        assert hasattr(disp, sym), sym


def test_headers_and_bindings_agree():
    """Every function declared in include/{algg,gpu_comp,ann_hip}.h is in the binding list and vice versa."""
    declared = set()
    for h in ("algg.h", "gpu_comp.h", "ann_hip.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(\w+)\s*\([^;{]*\)\s*;", src))
    declared -= {"defined"}
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)


def test_save_t_layout_matches_reference_abi():
    # ann.h:8-12: int, 4 size_t, 3 pointers, 2 pointers -> 80 bytes on LP64
    assert C.sizeof(_lib.SaveT) == 80
    assert _lib.SaveT.n.offset == 8 and _lib.SaveT.which_par.offset == 40 and _lib.SaveT.bases.offset == 72
