"""Generate the golden vectors in this directory FROM THE REFERENCE ITSELF.

Runs only in the authoring container: it loads oracle/_ref/libref_{f32,f64}.so, i.e. the reference's
own CPU path (precomp_cpu/query_cpu, /root/reference/algc.c + alg.c + compute.cl) compiled by
oracle/Makefile, and records inputs + outputs at fixed srandom() seeds.  The .npz files hold DATA only
(inputs, every save_t field, returned ids and squared distances); no reference source travels.

    python tests/golden/make_golden.py

Stream order mirrors time_results.c:92-105: srandom(seed) -> points -> precomp's own draws -> queries.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

# name: (seed, n, d, k, tries, rots_before, rot_len_before, rots_after, rot_len_after, Q, precisions)
CASES = {
    "tiny_appendixA": (7, 64, 16, 2, 2, 6, 1, 1, 1, 8, ("f32", "f64")),
    "defaults_d80": (11, 1000, 80, 10, 10, 6, 1, 1, 1, 50, ("f32", "f64")),
    "odd_everything": (12, 777, 33, 3, 4, 2, 3, 2, 2, 31, ("f32", "f64")),
    "one_try_one_query": (13, 500, 16, 1, 1, 1, 1, 1, 1, 1, ("f32", "f64")),
    "k17_d100": (14, 400, 100, 17, 2, 3, 2, 1, 2, 13, ("f32", "f64")),
    "pow2_d32": (15, 2000, 32, 10, 10, 6, 1, 1, 1, 100, ("f32", "f64")),
    "pow2_d64": (16, 1500, 64, 10, 10, 6, 1, 1, 1, 64, ("f32",)),
    "pow2_d128": (17, 1500, 128, 10, 10, 6, 1, 1, 1, 64, ("f32", "f64")),
    "few_candidates": (18, 40, 16, 4, 1, 1, 1, 1, 1, 5, ("f32", "f64")),
}


def main():
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle ref` in the authoring container"
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (seed, n, d, k, T, rb, rlb, ra, rla, Q, precs) in CASES.items():
        for prec in precs:
            ref = O.CpuBackend(prec, "ref")
            O.srandom(seed)
            pts = ref.gen_rand(n * d + (n * d) % 2)[: n * d].reshape(n, d)  # even count: Box-Muller cache empty
            ids, dists, save = ref.precomp(pts, k, T, rb, rlb, ra, rla)
            y = ref.gen_rand(Q * d + (Q * d) % 2)[: Q * d].reshape(Q, d)
            q_ids, q_dists = ref.query(save, pts, y)
            qa = min(Q, n)
            a_ids, a_dists = ref.query(save, pts, qa, alias=True)           # y == points pointer (Q3)
            c_ids, c_dists = ref.query(save, pts, pts[:qa].copy())           # same values, no aliasing
            blob = dict(seed=seed, params=np.array([n, d, k, T, rb, rlb, ra, rla, Q], dtype=np.int64),
                        points=pts, y=y, precomp_ids=ids, precomp_dists=dists,
                        d_short=save["d_short"], par_maxes=save["par_maxes"], graph=save["graph"],
                        row_means=save["row_means"], bases=save["bases"],
                        query_ids=q_ids, query_dists=q_dists, alias_ids=a_ids, alias_dists=a_dists,
                        copy_ids=c_ids, copy_dists=c_dists)
            for t, w in enumerate(save["which_par"]):
                blob["which_par_%d" % t] = w
            path = os.path.join(out_dir, "%s_%s.npz" % (name, prec))
            np.savez_compressed(path, **blob)
            print("%-28s %s ds=%d pm=%s  %.0f KB" % (name, prec, save["d_short"],
                  [int(v) for v in save["par_maxes"]][:5], os.path.getsize(path) / 1024))
    # sort-network probes (do_sort_cpu / sort_and_uniq_cpu, alg.c:137-144,224-230)
    rng = np.random.default_rng(2024)
    probes = {}
    for prec in ("f32", "f64"):
        ref = O.CpuBackend(prec, "ref")
        for L in (1, 2, 3, 5, 6, 8, 10, 12, 15, 16, 17, 20, 31, 64, 65, 100, 110, 160, 600):
            ids = rng.integers(0, max(2, L // 3), L).astype(np.uint64)
            keys = rng.integers(0, 6, L).astype(ref.ft)
            keys[rng.random(L) < 0.25] = np.inf
            s_ids, s_keys = ref.sort_net(ids, keys)
            u_ids, u_keys = ref.topk_stage(ids, keys)
            for nm, v in (("in_ids", ids), ("in_keys", keys), ("sort_ids", s_ids), ("sort_keys", s_keys),
                          ("uniq_ids", u_ids), ("uniq_keys", u_keys)):
                probes["%s_L%d_%s" % (prec, L, nm)] = v
    np.savez_compressed(os.path.join(out_dir, "sortnet_probes.npz"), **probes)
    # libc stream known answers
    O.srandom(7)
    a = [O.libc_random() for _ in range(4)]
    O.srandom(12345)
    b = [O.libc_random() for _ in range(4)]
    print("random():", a, b)


if __name__ == "__main__":
    main()
