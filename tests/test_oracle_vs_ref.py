"""Oracle vs the reference's own CPU path compiled into oracle/_ref (authoring container only)."""
import numpy as np
import pytest

from oracle import oracle_py as O
from tests.util import assert_save_equal, bits_equal

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference here)")

CASES = [  # n, d, k, T, rb, rlb, ra, rla, Q
    (64, 16, 2, 2, 6, 1, 1, 1, 7),
    (301, 24, 5, 3, 0, 1, 0, 1, 9),
    (640, 48, 4, 5, 3, 4, 2, 2, 33),
    (1200, 64, 10, 6, 6, 1, 1, 1, 40),
    (257, 17, 16, 1, 1, 1, 1, 1, 3),
]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("case", CASES)
def test_full_path_bitwise(prec, case):
    n, d, k, T, rb, rlb, ra, rla, Q = case
    ref, orc = O.CpuBackend(prec, "ref"), O.CpuBackend(prec, "oracle")
    O.srandom(4242 + n)
    orc.rand_norm_reset()
    pts = orc.gen_rand(n * d).reshape(n, d)
    O.srandom(77)
    i1, d1, s1 = ref.precomp(pts, k, T, rb, rlb, ra, rla)
    O.srandom(77)
    i2, d2, s2 = orc.precomp(pts, k, T, rb, rlb, ra, rla)
    assert np.array_equal(i1, i2) and bits_equal(d1, d2)
    assert_save_equal(s1, s2)
    orc.rand_norm_reset()
    y = orc.gen_rand(Q * d).reshape(Q, d)
    for kwargs in (dict(y=y), dict(y=min(Q, n), alias=True), dict(y=pts[: min(Q, n)].copy())):
        a = ref.query(s1, pts, **kwargs)
        b = orc.query(s2, pts, **kwargs)
        assert np.array_equal(a[0], b[0]) and bits_equal(a[1], b[1])


def test_rand_perm_stream():
    ref, orc = O.CpuBackend("f32", "ref"), O.CpuBackend("f32", "oracle")
    for d_pre, d_post in ((2, 80), (80, 128), (7, 128), (16, 16)):
        O.srandom(5)
        a = ref.rand_perm(d_pre, d_post)
        nxt_a = O.libc_random()
        O.srandom(5)
        b = orc.rand_perm(d_pre, d_post)
        assert np.array_equal(a, b) and nxt_a == O.libc_random()
