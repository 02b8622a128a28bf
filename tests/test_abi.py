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


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_synth_randnorm_is_the_reference_drivers_stream(prec):
    """annhip_synth_randnorm (host-only, threads for the libm part) against the oracle's serial generator -- which is
    pinned to the reference's rand_norm/genRand by the golden point sets (tests/test_oracle_golden.py): same values bit for bit, same number of
    random() draws, the pending second value of an odd-length call carried into the next call."""
    import numpy as np

    import approximatenn_amd as A
    from oracle import oracle_py as O
    orc = O.CpuBackend(prec, "oracle")
    for counts in ([7, 1, 2, 5], [100001, 3, 20000], [4096 * 2 * 3 + 1, 10]):
        O.srandom(77)
        orc.rand_norm_reset()
        want = [orc.gen_rand(c) for c in counts]
        after_want = O.libc_random()
        O.srandom(77)
        got = [A.synth_randnorm(c, prec, reset=(i == 0)) for i, c in enumerate(counts)]
        after_got = O.libc_random()
        for a, b in zip(want, got):
            assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))
        assert after_want == after_got
