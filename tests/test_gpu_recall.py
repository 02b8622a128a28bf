"""GPU: exact-rank recall scoring (annhip_recall_ranks) against a numpy brute force, and the
reference's quality level on its default shape (P(correct) ~0.60 precomp / ~0.52 query)."""
import numpy as np
import pytest
import torch

import approximatenn_amd as A
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _brute_ranks(points, y, guess, self_exclude):
    p, q = points.astype(np.float64), y.astype(np.float64)
    d2 = ((q[:, None, :] - p[None, :, :]) ** 2).sum(-1)          # integer-valued inputs: exact in every precision
    ranks = np.zeros(guess.shape, dtype=np.int64)
    for i in range(len(y)):
        row = d2[i].copy()
        if self_exclude:
            row[i] = np.inf
        for j, g in enumerate(guess[i]):
            ranks[i, j] = int(np.sum(row < d2[i, g]))
    return ranks


@pytest.mark.parametrize("d,dtype", [(32, np.float32), (128, np.float64), (40, np.float32)])
def test_ranks_match_brute_force_on_integer_grid(d, dtype):
    rng = np.random.default_rng(3)
    n, Q, k = 1500, 37, 7
    pts = rng.integers(-6, 7, size=(n, d)).astype(dtype)
    y = rng.integers(-6, 7, size=(Q, d)).astype(dtype)
    guess = rng.integers(0, n, size=(Q, k)).astype(np.int64)      # arbitrary, unsorted guesses
    got = A.recall_ranks(torch.from_numpy(pts).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(guess).cuda())
    assert np.array_equal(got.cpu().numpy(), _brute_ranks(pts, y, guess, False))
    got = A.recall_ranks(torch.from_numpy(pts).cuda(), torch.from_numpy(pts[:Q].copy()).cuda(), torch.from_numpy(guess).cuda(),
                         self_exclude=True)
    assert np.array_equal(got.cpu().numpy(), _brute_ranks(pts, pts[:Q], guess, True))


def test_reference_default_shape_quality():
    """n=1000 d=80 k=10 tries=10 (the reference drivers' defaults): recall is modest by construction (Q1/Q2)."""
    orc = O.CpuBackend("f32", "oracle")
    O.srandom(20)
    orc.rand_norm_reset()
    n, d, k, Q = 1000, 80, 10, 200
    pts = orc.gen_rand(n * d).reshape(n, d)
    y = orc.gen_rand(Q * d).reshape(Q, d)
    ids, _, save = A.precomp(pts, k)
    try:
        tp = torch.from_numpy(pts).cuda()
        r = A.recall_ranks(tp, tp, torch.from_numpy(ids.astype(np.int64)).cuda(), self_exclude=True)
        s_pre = A.recall_summary(r, k)
        q_ids, _ = A.query(save, pts, y)
        r = A.recall_ranks(tp, torch.from_numpy(y).cuda(), torch.from_numpy(q_ids.astype(np.int64)).cuda())
        s_q = A.recall_summary(r, k)
        # The survey measured 0.598 / 0.522 with the reference's own scorer; that scorer double-counts one term of
        # its distance tree whenever a level is odd (`z = step % 1`, test_correctness.c:218-223; d=80 has one), which
        # perturbs its ground truth.  With exact ranks the same index scores ~0.75 / ~0.65.
        print("precomp", s_pre, "query", s_q)
        assert 0.60 < s_pre["prob_correct"] < 0.90, s_pre
        assert 0.50 < s_q["prob_correct"] < 0.85, s_q
        assert s_pre["prob_correct"] > s_q["prob_correct"]      # the query path loses recall to the Q2 scramble
        assert s_pre["avg_index_score"] >= 0 and s_q["avg_index_score"] >= 0
    finally:
        A._lib.load("f32").annhip_cache_clear()
        save.free()
