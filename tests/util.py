"""Shared helpers for the test-suite (golden loading, bitwise comparisons)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases(prec=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*_f32.npz")) + glob.glob(os.path.join(GOLDEN, "*_f64.npz"))):
        name = os.path.basename(p)[:-4]
        if prec is None or name.endswith(prec):
            out.append(name)
    return out


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    n, d, k, T, rb, rlb, ra, rla, Q = [int(v) for v in g["params"]]
    g["cfg"] = dict(n=n, d=d, k=k, tries=T, rb=rb, rlb=rlb, ra=ra, rla=rla, Q=Q, seed=int(g["seed"]))
    g["prec"] = name[-3:]
    g["save"] = dict(tries=T, n=n, k=k, d_short=int(g["d_short"]), d_long=d, par_maxes=g["par_maxes"],
                     graph=g["graph"], which_par=[g["which_par_%d" % t] for t in range(T)],
                     row_means=g["row_means"], bases=g["bases"])
    return g


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def assert_save_equal(a, b):
    """cdiff_save of /root/reference/compare_results.c:152-171, tightened to bit equality."""
    for f in ("tries", "n", "k", "d_short", "d_long"):
        assert int(a[f]) == int(b[f]), f
    assert np.array_equal(np.asarray(a["par_maxes"], dtype=np.uint64), np.asarray(b["par_maxes"], dtype=np.uint64))
    assert np.array_equal(a["graph"], b["graph"]), "graph"
    for t, (w1, w2) in enumerate(zip(a["which_par"], b["which_par"])):
        assert np.array_equal(w1, w2), "which_par[%d]" % t
    assert bits_equal(a["row_means"], b["row_means"]), "row_means"
    assert bits_equal(a["bases"], b["bases"]), "bases"
