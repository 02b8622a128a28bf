"""GPU: edge cases of the domain and miniature versions of BASELINE.json's other configurations."""
import ctypes as C

import numpy as np
import pytest

import approximatenn_amd as A
from oracle import oracle_py as O
from tests.util import assert_save_equal, bits_equal, load_golden

pytestmark = pytest.mark.gpu


def _data(prec, n, d, Q, seed):
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(seed)
    orc.rand_norm_reset()
    return orc, orc.gen_rand(n * d).reshape(n, d), orc.gen_rand(Q * d).reshape(Q, d)


def _both(prec, pts, y, k, T, seed=3, **rot):
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(seed)
    o = orc.precomp(pts, k, T, **rot)
    O.srandom(seed)
    names = dict(rb="rots_before", rlb="rot_len_before", ra="rots_after", rla="rot_len_after")
    g = A.precomp(pts, k, T, **{names[a]: v for a, v in rot.items()})
    return orc, o, g


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_cfg5_miniature_large_k(prec):
    """BASELINE configs[4] in small: d=256, k=100 (K1=101 selection buffers, L2=10100 -> 8193-entry network)."""
    orc, pts, y = _data(prec, 1500, 256, 24, 55)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both(prec, pts, y, 100, 2)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        assert_save_equal(save.to_dict(), o_save)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("d", [16, 512, 1024, 48, 130])
def test_row_lengths_fast_and_generic(d):
    """smallest / largest register-tiled d, and two generic (non power of two) ones; float."""
    orc, pts, y = _data("f32", 700, d, 30, 70 + d)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both("f32", pts, y, 5, 3)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load("f32").annhip_cache_clear()
        save.free()


def test_k_larger_than_sorted_prefix_forces_exact_path():
    """P(L1) < k: the top-k includes entries the network never sorts; only the literal path is right."""
    orc = O.CpuBackend("f32", "oracle")
    found = None
    for seed in range(200):
        O.srandom(seed)
        orc.rand_norm_reset()
        pts = orc.gen_rand(30 * 16).reshape(30, 16)
        O.srandom(seed)
        _, _, sv = orc.precomp(pts, 20, 1, ra=0)   # d_short is 1 here: a post-Walsh rotation needs 2 coordinates
        L1 = (sv["d_short"] + 1) * int(sv["par_maxes"][0])
        if (1 << int(np.floor(np.log2(L1)))) < 20:
            found = (seed, pts, sv)
            break
    assert found, "no seed with P(L1) < k"
    seed, pts, o_save = found
    y = orc.gen_rand(6 * 16).reshape(6, 16)
    O.srandom(seed)
    o_ids, o_d, _ = orc.precomp(pts, 20, 1, ra=0)
    O.srandom(seed)
    ids, dd, save = A.precomp(pts, 20, 1, rots_after=0)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load("f32").annhip_cache_clear()
        save.free()


def test_empty_and_single_query_batches():
    g = load_golden("pow2_d32_f32")
    save = A.Save.from_dict("f32", g["save"])
    pts = np.ascontiguousarray(g["points"])
    ids, dd = A.query(save, pts, np.zeros((0, pts.shape[1]), dtype=np.float32))
    assert ids.shape == (0, 10) and dd.shape == (0, 10)
    orc = O.CpuBackend("f32", "oracle")
    one = g["y"][:1]
    want, got = orc.query(g["save"], pts, one), A.query(save, pts, one)    # Q=1: no scramble (Q2)
    assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    A._lib.load("f32").annhip_cache_clear()


def test_non_default_rotation_parameters():
    orc, pts, y = _data("f64", 900, 64, 40, 91)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both("f64", pts, y, 7, 4, rb=3, rlb=8, ra=2, rla=3)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        assert_save_equal(save.to_dict(), o_save)
    finally:
        A._lib.load("f64").annhip_cache_clear()
        save.free()


def test_lifecycle_register_cleanup_and_reinit():
    lib = A._lib.load("f32")
    calls = []
    cb = C.CFUNCTYPE(None)(lambda: calls.append(1))
    lib.gpu_init()
    lib.register_cleanup(cb)          # initialised: runs at gpu_cleanup (gpu_comp.c:93-101)
    assert calls == []
    lib.gpu_cleanup()
    assert calls == [1]
    lib.register_cleanup(cb)          # not initialised: runs immediately
    assert calls == [1, 1]
    g = load_golden("tiny_appendixA_f32")
    save = A.Save.from_dict("f32", g["save"])
    ids, dd = A.query(save, np.ascontiguousarray(g["points"]), g["y"])   # re-initialises by itself
    assert np.array_equal(ids, g["query_ids"])
    lib.annhip_cache_clear()


def test_residency_cache_detects_changed_content():
    """Same host addresses, different content => the cached index must not be reused."""
    g = load_golden("pow2_d64_f32")
    save = A.Save.from_dict("f32", g["save"])
    pts = np.ascontiguousarray(g["points"]).copy()
    ids, _ = A.query(save, pts, g["y"])
    assert np.array_equal(ids, g["query_ids"])
    orc = O.CpuBackend("f32", "oracle")
    pts[:] = pts[::-1].copy()         # in place: same pointer, reversed rows
    want = orc.query(g["save"], pts, g["y"])
    got = A.query(save, pts, g["y"])
    assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    A._lib.load("f32").annhip_cache_clear()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_network_in_hbm_and_chunked_host_driven_exact_path(prec, monkeypatch):
    """Paths that only very large shapes reach (full-size cfg5 rows do not fit LDS; cfg4's Q=100k exceeds the exact
    path's workspace budget): force them with the size hooks on a ties-heavy dataset, in both path modes."""
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(77)
    orc.rand_norm_reset()
    half = orc.gen_rand(600 * 32).reshape(600, 32)
    pts = np.ascontiguousarray(np.concatenate([half, half]))     # duplicated points: most queries get rejected
    y = orc.gen_rand(70 * 32).reshape(70, 32)
    O.srandom(9)
    o_ids, o_d, o_save = orc.precomp(pts, 6, 4)
    want = orc.query(o_save, pts, y)
    monkeypatch.setenv("ANN_HIP_LDS_ROW_MAX", "64")              # every exact-path row sorts in HBM
    monkeypatch.setenv("ANN_HIP_EXACT_BYTES", "20000")           # a few rows per chunk, host-driven
    for exact in (False, True):
        if exact:
            monkeypatch.setenv("ANN_HIP_EXACT", "1")
        O.srandom(9)
        ids, dd, save = A.precomp(pts, 6, 4)
        try:
            assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
            got = A.query(save, pts, y)
            assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
        finally:
            A._lib.load(prec).annhip_cache_clear()
            save.free()
