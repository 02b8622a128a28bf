"""The tie path's algorithm (tools/tie_model.py, the Python twin of approximatenn_amd/csrc/ann_tie.h) against a literal
model of the reference's sort_and_uniq (/root/reference/alg.c:224-230, compute.cl:188-217) -- CPU only."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import tie_model as M  # noqa: E402


def test_tie_path_equals_the_literal_network_on_random_rows():
    tot, hit = M.self_check(rows=3000, seed=7)
    assert tot == 3000 and hit > 500  # a good share of the rows has exactly one run of ties


def test_rows_that_do_not_qualify_are_refused():
    inf = float("inf")
    # two runs of ties among the best keys
    ids = [1, 2, 3, 4, 5, 6, 7, 8] + [9] * 8
    key = [1.0, 1.0, 2.0, 2.0, 3.0, 4.0, 5.0, 6.0] + [inf] * 8
    assert M.tie_path(16, 16, key, ids, 3, 9) is None
    # no tie at all
    key = [float(i + 1) for i in range(8)] + [inf] * 8
    assert M.tie_path(16, 16, key, ids, 3, 9) is None
    # one run: answered, and equal to the network
    key = [1.0, 2.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0] + [inf] * 8
    got = M.tie_path(16, 16, key, ids, 3, 9)
    assert got is not None and (got[0], got[1]) == M.literal(16, 16, key, ids, 3)
