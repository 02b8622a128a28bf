"""The query hash (codes_kernel / codes_lpq_kernel; /root/reference/alg.c:462-492, compute.cl:160-167,223-231,268-275)
on inputs where the SIGN OF ZERO decides hash bits: the reference adds `+ 0` in every node of the product tree, which
turns -0 into +0, and then reads the raw sign bit.  A query equal to the column means projects to sums of +-0 only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import approximatenn_amd as A  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.util import bits_equal  # noqa: E402


@pytest.mark.parametrize("lpq", ["1", "0"])
@pytest.mark.parametrize("prec,d", [("f32", 32), ("f32", 64), ("f32", 128), ("f64", 16), ("f64", 64), ("f32", 256), ("f32", 80)])
def test_hash_bits_when_products_are_signed_zeros(prec, d, lpq, monkeypatch):
    monkeypatch.setenv("ANN_HIP_CODES_LPQ", lpq)  # a lane per query (short power-of-two rows) / lanes per row
    A._lib.reload_env()
    ft = np.float32 if prec == "f32" else np.float64
    n, Q, k, T = 4096, 300, 5, 4
    rng = np.random.default_rng(d)
    pts = rng.integers(-3, 4, size=(n, d)).astype(ft)
    pts[:, ::5] = 0  # whole columns of zeros: centred value +0, product +-0 with the sign of the projection entry
    means = (pts.astype(np.float64).sum(axis=0) / n).astype(ft)  # exact: small integers, n a power of two
    y = pts[rng.integers(0, n, size=Q)].copy()
    y[:60] = means                      # every product is +-0
    y[60:120, : d // 2] = means[: d // 2]  # half of them
    pts, y = np.ascontiguousarray(pts), np.ascontiguousarray(y)
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(5)
    o_ids, o_d, o_save = orc.precomp(pts, k, T)
    O.srandom(5)
    ids, dd, save = A.precomp(pts, k, T)
    try:
        assert bits_equal(np.asarray(save.to_dict()["row_means"]).ravel(), means)
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        monkeypatch.delenv("ANN_HIP_CODES_LPQ")
        A._lib.reload_env()
        A._lib.load(prec).annhip_cache_clear()
        save.free()
