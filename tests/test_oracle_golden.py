"""The oracle (oracle/ann_oracle.c) against the golden vectors recorded from the compiled reference.

CPU-only.  This is what pins the oracle on the GPU box, where /root/reference does not exist.
"""
import numpy as np
import pytest

from oracle import oracle_py as O
from tests.util import assert_save_equal, bits_equal, golden_cases, load_golden, GOLDEN


def test_libc_stream_known_answers():
    # glibc TYPE_3 additive-feedback generator.
    O.srandom(7)
    assert [O.libc_random() for _ in range(4)] == [1045618677, 1863967299, 1272579899, 461085871]
    O.srandom(12345)
    assert [O.libc_random() for _ in range(4)] == [383100999, 858300821, 357768173, 455528251]


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_reproduces_reference_run(name):
    g = load_golden(name)
    c = g["cfg"]
    orc = O.CpuBackend(g["prec"], "oracle")
    # same libc stream as the generator: seed -> points -> precomp draws -> queries
    O.srandom(c["seed"])
    orc.rand_norm_reset()
    pts = orc.gen_rand(c["n"] * c["d"] + (c["n"] * c["d"]) % 2)[: c["n"] * c["d"]].reshape(c["n"], c["d"])
    assert bits_equal(pts, g["points"]), "randNorm restatement"
    ids, dists, save = orc.precomp(pts, c["k"], c["tries"], c["rb"], c["rlb"], c["ra"], c["rla"])
    assert np.array_equal(ids, g["precomp_ids"])
    assert bits_equal(dists, g["precomp_dists"])
    assert_save_equal(save, g["save"])
    y = orc.gen_rand(c["Q"] * c["d"] + (c["Q"] * c["d"]) % 2)[: c["Q"] * c["d"]].reshape(c["Q"], c["d"])
    assert bits_equal(y, g["y"])
    q_ids, q_d = orc.query(g["save"], pts, y)
    assert np.array_equal(q_ids, g["query_ids"]) and bits_equal(q_d, g["query_dists"])
    qa = min(c["Q"], c["n"])
    a_ids, a_d = orc.query(g["save"], pts, qa, alias=True)
    assert np.array_equal(a_ids, g["alias_ids"]) and bits_equal(a_d, g["alias_dists"])
    c_ids, c_d = orc.query(g["save"], pts, pts[:qa].copy())
    assert np.array_equal(c_ids, g["copy_ids"]) and bits_equal(c_d, g["copy_dists"])


def test_appendix_a_known_answers():
    g = load_golden("tiny_appendixA_f32")
    assert int(g["d_short"]) == 5 and [int(v) for v in g["par_maxes"]] == [5, 6]
    assert [int(v) for v in g["precomp_ids"][0]] == [34, 35]
    assert [int(v) for v in g["precomp_ids"][3]] == [7, 41]
    np.testing.assert_allclose(g["precomp_dists"][0], [17.3973, 18.8040], rtol=1e-5)
    assert [int(v) for v in g["which_par_0"][0]] == [7, 64, 64, 64, 64]
    assert [int(v) for v in g["which_par_0"][17]] == [50, 43, 35, 32, 8]
    b0 = g["bases"][0].astype(np.float64)
    np.testing.assert_allclose(b0 @ b0.T, np.eye(5), atol=1e-5)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_sort_network_probes(prec):
    z = np.load(GOLDEN + "/sortnet_probes.npz")
    orc = O.CpuBackend(prec, "oracle")
    Ls = sorted({int(k.split("_")[1][1:]) for k in z.files if k.startswith(prec)})
    assert 10 in Ls and 110 in Ls
    for L in Ls:
        p = "%s_L%d_" % (prec, L)
        s_ids, s_keys = orc.sort_net(z[p + "in_ids"], z[p + "in_keys"])
        assert np.array_equal(s_ids, z[p + "sort_ids"]) and bits_equal(s_keys, z[p + "sort_keys"]), L
        u_ids, u_keys = orc.topk_stage(z[p + "in_ids"], z[p + "in_keys"])
        assert np.array_equal(u_ids, z[p + "uniq_ids"]) and bits_equal(u_keys, z[p + "uniq_keys"]), L


def test_sort_prefix_quirk_q1():
    # descending keys: only the first 2^floor(log2 L) entries get sorted (L >= 16)
    orc = O.CpuBackend("f32", "oracle")
    for L, P in ((16, 16), (64, 64), (65, 64), (100, 64), (110, 64), (160, 128)):
        ids, keys = orc.sort_net(np.arange(L), np.arange(L, 0, -1))
        assert np.all(np.diff(keys[:P]) > 0)
        assert np.array_equal(keys[P:], np.arange(L - P, 0, -1).astype(np.float32))
    ids, keys = orc.sort_net(np.arange(10), np.arange(10, 0, -1))
    assert np.all(np.diff(keys[:8]) > 0) and list(keys[8:]) == [1.0, 2.0]


def test_tree_sum_order_q4():
    orc = O.CpuBackend("f32", "oracle")
    v = np.array([1e8, 1.0, -1e8, 1.0, 3.0], dtype=np.float32)
    # s=5: m0=m0+(m2+m4)=1e8+(-1e8+3)= ~0 (rounded), m1=m1+m3=2 ; s=2: m0+m1
    m0 = np.float32(v[0] + np.float32(v[2] + v[4]))
    m1 = np.float32(v[1] + v[3])
    assert orc.tree_sum(v) == np.float32(m0 + m1)


def test_query_scramble_q2():
    # copies of dataset points only find themselves when T == 1 or Q == 1
    g = load_golden("one_try_one_query_f32")
    assert int(g["copy_ids"][0, 0]) == 0 and g["copy_dists"][0, 0] == 0.0
    g = load_golden("pow2_d32_f32")
    self_found = int(np.sum(g["copy_ids"][:, 0] == np.arange(len(g["copy_ids"]))))
    assert 0 < self_found < len(g["copy_ids"])
    # aliased y == points excludes self entirely (Q3)
    assert not np.any(g["alias_ids"] == np.arange(len(g["alias_ids"]))[:, None])


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_sampled_precomp_checks_agree_with_the_full_oracle(prec):
    """oracle_precomp_tables_sample / oracle_precomp_graph_rows (what tests/test_gpu_configs.py checks a GPU-built index of
    BASELINE size with) against the full oracle_precomp -- itself pinned to the reference: same means, bases, draws from
    random(), bucket membership of the sampled points, graph rows and distances."""
    orc = O.CpuBackend(prec, "oracle")
    for (n, d, k, T) in [(3000, 64, 10, 10), (1500, 80, 7, 4), (700, 33, 3, 5)]:
        O.srandom(5)
        orc.rand_norm_reset()
        pts = orc.gen_rand(n * d).reshape(n, d)
        O.srandom(77)
        ids, dd, save = orc.precomp(pts, k, T)
        after = O.libc_random()
        rows = np.array([0, 1, n // 2, n - 1, 17, n // 3], dtype=np.uint64)
        O.srandom(77)
        ds, means, bases, codes = orc.precomp_tables_sample(pts, k, T, rows)
        assert O.libc_random() == after
        assert ds == save["d_short"] and bits_equal(means, save["row_means"]) and bits_equal(bases, save["bases"])
        for i, x in enumerate(rows):
            for t in range(T):
                assert x in save["which_par"][t][int(codes[i, t])]
        gi, gd = orc.precomp_graph_rows(save, pts, rows)
        sel = rows.astype(np.int64)
        assert np.array_equal(gi, ids[sel]) and bits_equal(gd, dd[sel])
