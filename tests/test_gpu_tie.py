"""The tie path (approximatenn_amd/csrc/ann_tie.h): rows whose k+1 best candidates hold ONE run of equal distances
between different ids are answered from their class bits instead of the reference's network (sort_and_uniq,
/root/reference/alg.c:224-230; compute.cl:188-217).  Checked here three ways: free-standing rows against the literal
network kernel (bit-equal, thousands of rows, every row length class), against tools/tie_model.py (same decision which
rows qualify), and end to end against the oracle on data sets with a few duplicated points."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import approximatenn_amd as A  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.util import bits_equal  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _lg(x):
    return int(x).bit_length() - 1


def _need_len(L, k):
    if L < 16:
        return L
    P = 1 << _lg(L)
    return min(L, max(P, k) + 1)


def _rows(rng, L, k, nq, n, ft, groups):
    """nq rows of reference length L: ids drawn from a small pool (copies!), padding id n; `groups` lists of pool members
    forced to one distance."""
    ln, P = _need_len(L, k), 1 << _lg(L)
    ids = np.full((nq, ln), n, np.uint32)
    dist = np.full((nq, ln), np.inf, ft)
    cand_d = np.full((nq, k + 1), np.inf, ft)
    cand_i = np.full((nq, k + 1), 0xFFFFFFFF, np.uint32)
    for r in range(nq):
        pool = rng.choice(n, size=int(rng.choice([k + 8, 2 * k + 9, 60, 300])), replace=False)
        dv = rng.integers(1, 1 << 20, size=pool.size).astype(ft) / ft(64)
        for g in range(groups[r % len(groups)]):
            mem = rng.choice(pool.size, size=int(rng.choice([2, 2, 2, 3, 5])), replace=False)
            dv[mem] = dv[mem[0]]
            if g == 0 and rng.random() < 0.7:  # make the run matter: among the smallest
                dv[mem] = dv.min() + ft(rng.integers(0, 3))
        fill = rng.random(ln) < rng.choice([0.15, 0.5, 0.9])
        pick = rng.integers(0, pool.size, size=ln)
        ids[r, fill] = pool[pick[fill]]
        dist[r, fill] = dv[pick[fill]]
        keys = sorted(set((float(dist[r, j]), int(ids[r, j])) for j in range(min(P, ln)) if np.isfinite(dist[r, j])))[:k + 1]
        for t, (dd, ii) in enumerate(keys):
            cand_d[r, t], cand_i[r, t] = dd, ii
    return ids, dist, cand_d, cand_i


def _sort_rows(lib, L, k, ids, dist, cand_d, cand_i, use_tie, derive=0):
    nq = ids.shape[0]
    ti, td = torch.from_numpy(ids.copy()).cuda(), torch.from_numpy(dist.copy()).cuda()
    oi = torch.zeros((nq, k), dtype=torch.int32, device="cuda")
    od = torch.zeros((nq, k), dtype=td.dtype, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    cd, ci = torch.from_numpy(cand_d).cuda(), torch.from_numpy(cand_i.view(np.int32)).cuda()
    torch.cuda.synchronize()
    lib.annhip_test_sort_rows(L, k, nq, ti.data_ptr(), td.data_ptr(), cd.data_ptr() if use_tie else None,
                              ci.data_ptr() if use_tie else None, oi.data_ptr(), od.data_ptr(), cnt.data_ptr(), derive)
    torch.cuda.synchronize()
    return oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), int(cnt.item())


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("L,k", [(16, 3), (40, 5), (64, 10), (100, 10), (256, 1), (700, 17), (1024, 10), (4096, 10),
                                 (6006, 10), (6006, 63), (9000, 10), (20000, 12), (40000, 10)])
def test_tie_rows_match_the_network(prec, L, k):
    """Free-standing rows: tie path on (candidate lists given) vs. the literal network kernel, bit for bit; the tie
    path must actually answer a good share of the single-run rows.  L = 40000: P = 32768 is beyond the tie path's
    16384 positions -- everything falls through to the network."""
    lib = A._lib.load(prec)
    ft = np.float32 if prec == "f32" else np.float64
    rng = np.random.default_rng(L * 131 + k)
    nq = 96 if L <= 9000 else 24
    ids, dist, cand_d, cand_i = _rows(rng, L, k, nq, 1 << 20, ft, groups=[1, 1, 1, 2, 0])
    li, ld, _ = _sort_rows(lib, L, k, ids, dist, cand_d, cand_i, False)
    gi, gd, resolved = _sort_rows(lib, L, k, ids, dist, cand_d, cand_i, True)
    assert np.array_equal(gi, li) and bits_equal(gd, ld)
    # ... and with the candidate lists derived from the rows by the kernel itself (sharded hosts, staged API)
    di, dd_, resolved_d = _sort_rows(lib, L, k, ids, dist, cand_d, cand_i, False, derive=1)
    assert np.array_equal(di, li) and bits_equal(dd_, ld)
    if L <= 16384 + 16383:
        assert resolved >= nq // 16, resolved
        # (the derived list's keys live in LDS: 16 B each in double -- 8 192 of them plus the tie path's own state
        # exceed a CU's LDS, as do 16 384 in float: those rows take the network)
        too_big = L >= (8192 if prec == "f64" else 16384)
        assert resolved_d == (0 if too_big else resolved), (resolved_d, resolved)
    else:
        assert resolved == 0 and resolved_d == 0


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_tie_rows_same_decision_as_the_model(prec):
    """tools/tie_model.py is the algorithm in Python (itself checked against a literal network there): the kernel
    answers exactly the rows the model answers, with the same result."""
    import tie_model as M
    lib = A._lib.load(prec)
    ft = np.float32 if prec == "f32" else np.float64
    rng = np.random.default_rng(7)
    for L, k in [(16, 2), (24, 5), (64, 10), (100, 3), (200, 10), (256, 20)]:
        ids, dist, cand_d, cand_i = _rows(rng, L, k, 64, 1000, ft, groups=[1, 1, 2, 0, 3])
        gi, gd, resolved = _sort_rows(lib, L, k, ids, dist, cand_d, cand_i, True)
        want = 0
        for r in range(ids.shape[0]):
            key = [float(v) for v in dist[r]]
            rid = [int(v) for v in ids[r]]
            m = M.tie_path(L, len(rid), key, rid, k, 1000)
            want += m is not None
            lit = M.literal(L, len(rid), key, rid, k)
            assert [int(v) for v in gi[r]] == lit[0] and [float(v) for v in gd[r]] == lit[1], (L, k, r)
            if m is not None:
                assert (m[0], m[1]) == lit
        assert resolved == want, (L, k, resolved, want)


@pytest.mark.parametrize("fuse", ["0", "1"])
@pytest.mark.parametrize("prec,d,k,T", [("f32", 64, 10, 6), ("f64", 32, 10, 4), ("f32", 128, 5, 8), ("f32", 80, 17, 5)])
def test_queries_with_a_few_duplicated_points(prec, d, k, T, fuse, monkeypatch):
    """End to end: 2 % of the points exist twice (equal distances between different ids wherever one of them is a
    candidate).  Precomp and query against the oracle with the tie path on; the statistics show it answered rows.
    fuse: stage 2 as its own kernel (what large batches take) / in the tail of the stage-1 workgroup (small batches)."""
    n, Q = 6000, 600
    monkeypatch.setenv("ANN_HIP_FUSE", fuse)
    A._lib.reload_env()
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(4242 + d)
    orc.rand_norm_reset()
    pts = orc.gen_rand(n * d).reshape(n, d)
    y = orc.gen_rand(Q * d).reshape(Q, d)
    rng = np.random.default_rng(d)
    src = rng.choice(n, size=n // 50, replace=False)
    dst = rng.choice(np.setdiff1d(np.arange(n), src), size=src.size, replace=False)
    pts[dst] = pts[src]
    y[:100] = pts[src[:100]] + (0.01 * orc.gen_rand(100 * d).reshape(100, d)).astype(pts.dtype)
    pts, y = np.ascontiguousarray(pts), np.ascontiguousarray(y)
    O.srandom(11)
    o_ids, o_d, o_save = orc.precomp(pts, k, T)
    O.srandom(11)
    ids, dd, save = A.precomp(pts, k, T)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        ix = A.Index.from_save(save, torch.from_numpy(pts).cuda())
        ix.stats(reset=True)
        want = orc.query(o_save, pts, y)
        g_ids, g_d, _ = ix.query(torch.from_numpy(y).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(g_ids.cpu().numpy().astype(np.uint64), want[0]) and bits_equal(g_d.cpu().numpy(), want[1])
        st = ix.stats()
        assert st["exact_queries"] > 0 and st["tie_queries"] > 0, st
        assert st["tie_queries"] <= st["exact_queries"]
        want = orc.query(o_save, pts, 500, alias=True)
        g_ids, g_d, _ = ix.query(torch.from_numpy(pts[:500]).cuda(), alias=True)
        torch.cuda.synchronize()
        assert np.array_equal(g_ids.cpu().numpy().astype(np.uint64), want[0]) and bits_equal(g_d.cpu().numpy(), want[1])
        ix.close()
    finally:
        monkeypatch.delenv("ANN_HIP_FUSE")
        A._lib.reload_env()
        A._lib.load(prec).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("prec,world", [("f32", 2), ("f64", 3)])
def test_sharded_exact_step_uses_the_tie_path(prec, world):
    """Rows sharded over thread ranks of one GPU: the flagged queries' rows are reduced across the ranks and EVERY rank
    answers them -- from a candidate list it derives from the reduced row (only the owner ever held the merged one).
    Same answers as the oracle, and the ranks' statistics show rows answered by the tie path."""
    from tests import test_gpu_sharded as TS
    d, k, T, n, Q = 64, 10, 6, 6000, 400
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(99)
    orc.rand_norm_reset()
    pts = orc.gen_rand(n * d).reshape(n, d)
    y = orc.gen_rand(Q * d).reshape(Q, d)
    rng = np.random.default_rng(3)
    src = rng.choice(n, size=n // 50, replace=False)
    dst = rng.choice(np.setdiff1d(np.arange(n), src), size=src.size, replace=False)
    pts[dst] = pts[src]
    y[:100] = pts[src[:100]] + (0.01 * orc.gen_rand(100 * d).reshape(100, d)).astype(pts.dtype)
    pts, y = np.ascontiguousarray(pts), np.ascontiguousarray(y)
    O.srandom(11)
    _, _, o_save = orc.precomp(pts, k, T)
    want = orc.query(o_save, pts, y)
    TS.LAST_TIE_QUERIES.clear()
    for s_ids, s_d, s_ex in TS._run_sharded(prec, o_save, pts, y, world):
        assert np.array_equal(s_ids, want[0]) and bits_equal(s_d, want[1]) and s_ex > 0
    assert len(TS.LAST_TIE_QUERIES) == world and all(v > 0 for v in TS.LAST_TIE_QUERIES.values()), TS.LAST_TIE_QUERIES
