#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: precomp + query (+ aliased query, + sharded query on thread ranks) of the HIP
backend against the oracle on random shapes and seeds, every result bit for bit.

    python tools/fuzz_parity.py [--cases 60] [--seed 1] [--prec f32,f64] [--sharded]

Shapes are drawn to hit the different code paths: power-of-two d (register layout + bucket-centric precomp), d with a
static 3- or 5-lane layout (24, 40, 48, 80, 96, 160), other multiples of the 16-byte chunk, arbitrary d (LDS tree);
k from 1 to 40 (and occasionally beyond the sorted prefix); tries 1..12; duplicated points (ties) now and then.
Every mismatch is printed with its case (shape, seed, rotations) so that it can be replayed; exit status 1 if there was one.
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import approximatenn_amd as A
from oracle import oracle_py as O


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def draw_big_case(rng):
    """Larger shapes (d_short 11..14, thousands of candidates per row, batches of thousands): the sizes at which the
    production paths (bucket-centric precomp, multi-wave stage 1, fused stage 2) run with realistic occupancy."""
    d = rng.choice([64, 128, 80, 32, 256])
    k = rng.choice([5, 10, 10, 20, 50])
    n = rng.choice([20000, 30000, 50000])
    return dict(kind="big", n=n, d=d, k=k, T=rng.choice([4, 7, 10]), Q=rng.choice([1000, 3000]), dup=rng.random() < 0.1,
                seed=rng.randrange(1, 1 << 30), rb=6, rlb=1, ra=1, rla=1)


def draw_bigk_case(rng):
    """k >= 32: stage-2 rows beyond the fused kernel's LDS row (stage2_select_kernel + its literal fallback)."""
    c = draw_case(rng)
    c["k"] = rng.choice([32, 33, 40, 50, 64, 100])
    c["n"] = rng.choice([1200, 2500, 4000])
    c["T"] = rng.choice([1, 2, 3, 5])
    c["Q"] = rng.choice([1, 30, 100])
    c["dup"] = rng.random() < 0.3
    c["kind"] += "+bigk"
    return c


def draw_case(rng):
    kind = rng.choice(["pow2", "pow2", "static_oc", "static_oc", "chunks", "any"])
    if kind == "pow2":
        d = rng.choice([16, 32, 64, 128, 256])
    elif kind == "static_oc":
        d = rng.choice([24, 40, 48, 80, 96, 160])
    elif kind == "chunks":
        d = rng.choice([20, 28, 36, 44, 52, 60, 72, 100, 112, 144])
    else:
        d = rng.choice([17, 19, 23, 33, 50, 65, 77, 130, 150, 250])
    k = rng.choice([1, 2, 3, 5, 8, 10, 10, 10, 16, 17, 25, 40])
    n = rng.choice([300, 700, 1000, 2500, 6000])
    if n <= 4 * k:
        n = 8 * k + 100
    T = rng.choice([1, 2, 3, 5, 7, 10, 12])
    Q = rng.choice([1, 7, 64, 200, 500])
    return dict(kind=kind, n=n, d=d, k=k, T=T, Q=Q, dup=rng.random() < 0.15, seed=rng.randrange(1, 1 << 30),
                rb=rng.choice([6, 6, 2, 0]), rlb=rng.choice([1, 1, 2]), ra=rng.choice([1, 1, 0, 2]), rla=1)


def run_case(c, prec, sharded):
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(c["seed"])
    orc.rand_norm_reset()
    n, d, k, T, Q = c["n"], c["d"], c["k"], c["T"], c["Q"]
    pts = orc.gen_rand(n * d).reshape(n, d)
    if not np.isfinite(pts).all():
        # random() == 0 makes Box-Muller emit inf/NaN (log 0); the column mean is then NaN, every centred coordinate too,
        # and the hash is the sign bit of NaNs -- which no two arithmetic units agree on.  Found by this sweep
        # (seed 617739607: one bucket of 5 999 points); not a comparison the reference itself could pass.
        return None
    if c["dup"]:
        pts[n // 2:] = pts[: n - n // 2]           # every point twice: ties between different ids
    pts = np.ascontiguousarray(pts)
    y = np.ascontiguousarray(orc.gen_rand(Q * d).reshape(Q, d))
    if 2 * c["rlb"] > d:
        c["rlb"] = 1
    rot = dict(rb=c["rb"], rlb=c["rlb"], ra=c["ra"], rla=c["rla"])
    O.srandom(c["seed"] ^ 0x5A5A)
    o_ids, o_d, o_save = orc.precomp(pts, k, T, **rot)
    O.srandom(c["seed"] ^ 0x5A5A)
    ids, dd, save = A.precomp(pts, k, T, rots_before=c["rb"], rot_len_before=c["rlb"], rots_after=c["ra"], rot_len_after=c["rla"])
    try:
        if not (np.array_equal(ids, o_ids) and bits_equal(dd, o_d)):
            return "precomp ids/dists"
        sd = save.to_dict()
        for f in ("par_maxes", "graph"):
            if not np.array_equal(np.asarray(sd[f], dtype=np.uint64), np.asarray(o_save[f], dtype=np.uint64)):
                return "save." + f
        for t, (a, b) in enumerate(zip(sd["which_par"], o_save["which_par"])):
            if not np.array_equal(a, b):
                return "save.which_par[%d]" % t
        if not (bits_equal(sd["row_means"], o_save["row_means"]) and bits_equal(sd["bases"], o_save["bases"])):
            return "save.row_means/bases (bits)"
        want = orc.query(o_save, pts, y)
        got = A.query(save, pts, y)
        if not (np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])):
            return "query"
        qa = min(Q, n)
        want = orc.query(o_save, pts, qa, alias=True)
        got = A.query(save, pts, pts[:qa])
        if not (np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])):
            return "aliased query"
        if sharded:
            from tests.test_gpu_sharded import _run_sharded
            want = orc.query(o_save, pts, y)
            for world in (2, 5):
                for s_ids, s_d, _ in _run_sharded(prec, o_save, pts, y, world):
                    if not (np.array_equal(s_ids, want[0]) and bits_equal(s_d, want[1])):
                        return "sharded query, world %d" % world
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--prec", default="f32,f64")
    ap.add_argument("--sharded", action="store_true")
    ap.add_argument("--big", action="store_true", help="larger shapes (slow: the oracle is the bottleneck)")
    ap.add_argument("--bigk", action="store_true", help="k >= 32 (long stage-2 rows)")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bad = 0
    for i in range(a.cases):
        c = draw_big_case(rng) if a.big else draw_bigk_case(rng) if a.bigk else draw_case(rng)
        for prec in a.prec.split(","):
            err = run_case(dict(c), prec, a.sharded)
            if err:
                bad += 1
                print("MISMATCH (%s) %s %r" % (err, prec, c), flush=True)
        if (i + 1) % (2 if a.big else 5 if a.bigk else 10) == 0:
            print("%d cases done, %d mismatches" % (i + 1, bad), flush=True)
    print("fuzz: %d cases x %s, %d mismatches" % (a.cases, a.prec, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
