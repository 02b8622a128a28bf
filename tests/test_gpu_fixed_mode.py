"""GPU: the opt-in "fixed" query mode (annhip_index_set_fixed).  Parity unpinned BY DESIGN -- the reference
has no such mode -- so it is checked against what it promises: a brute force (numpy, float64) over exactly the candidate
sets the index defines (own hash codes -> own bucket + Hamming-1 buckets of every try; then the graph neighbours of the
stage-1 result), and against the parity mode on recall."""
import numpy as np
import pytest
import torch

import approximatenn_amd as A
from approximatenn_amd.sharded import HipEngine
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _build(prec, n, d, k, T, seed):
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(seed)
    orc.rand_norm_reset()
    pts = np.ascontiguousarray(orc.gen_rand(n * d).reshape(n, d))
    O.srandom(seed + 1)
    tp = torch.from_numpy(pts).cuda()
    ix = A.Index.precomp(tp, k, T)
    return orc, pts, tp, ix


def _brute(save, pts, y, codes, k, alias_ids=None):
    """k smallest distinct (distance, id) among the candidates of the fixed mode, both stages; float64 arithmetic."""
    n, T, ds = len(pts), save["tries"], save["d_short"]
    graph = np.asarray(save["graph"]).reshape(n, k)
    out_i, out_d = [], []
    for x in range(len(y)):
        cand = []
        for t in range(T):
            tab = np.asarray(save["which_par"][t]).reshape(1 << ds, -1)
            c = int(codes[x, t])
            for yy in range(ds + 1):
                row = tab[c ^ ((1 << (yy - 1)) if yy else 0)]
                cand.append(row[row < n])
        cand = np.unique(np.concatenate(cand)).astype(np.int64)
        if alias_ids is not None:
            cand = cand[cand != alias_ids[x]]

        def best(ids):
            dd = ((pts[ids].astype(np.float64) - y[x].astype(np.float64)) ** 2).sum(1)
            o = np.lexsort((ids, dd))[:k]
            return ids[o], dd[o]
        top, _ = best(cand)
        c2 = np.unique(np.concatenate([top, graph[top].reshape(-1)])).astype(np.int64)
        c2 = c2[c2 < n]
        if alias_ids is not None:
            c2 = c2[c2 != alias_ids[x]]
        i2, d2 = best(c2)
        out_i.append(i2), out_d.append(d2)
    return out_i, out_d


@pytest.mark.parametrize("prec,n,d,k,T", [("f64", 3000, 32, 5, 4), ("f32", 5000, 64, 10, 6), ("f64", 2500, 80, 8, 3),
                                         ("f64", 2000, 16, 33, 2)])
def test_fixed_mode_is_the_exact_top_k_of_its_candidate_sets(prec, n, d, k, T):
    orc, pts, tp, ix = _build(prec, n, d, k, T, 4100 + d)
    try:
        y = np.ascontiguousarray(orc.gen_rand(60 * d).reshape(60, d))
        ty = torch.from_numpy(y).cuda()
        save = ix.export()
        sd = save.to_dict()
        eng = HipEngine(ix)
        ix.set_fixed(True)                            # (the parity mode hashes only the queries whose codes it reads)
        codes = _codes_of(eng, ty, T)                 # code[q*T+t]: the array as the hash kernel writes it
        ids, dd, _ = ix.query(ty)
        ids, dd = ids.cpu().numpy(), dd.cpu().numpy()
        want_i, want_d = _brute(sd, pts, y, codes, k)
        tol = 1e-9 if prec == "f64" else 2e-5
        for x in range(60):
            m = len(want_i[x])
            assert np.allclose(dd[x, :m], want_d[x], rtol=tol, atol=0), (x, dd[x], want_d[x])
            assert np.all(np.isinf(dd[x, m:])) and np.all(ids[x, m:] == n)
            same = ids[x, :m] == want_i[x]
            if not same.all():   # a different id only where two candidates are (nearly) equally far
                bad = np.flatnonzero(~same)
                gd = ((pts[ids[x, bad]].astype(np.float64) - y[x]) ** 2).sum(1)
                assert np.allclose(gd, want_d[x][bad], rtol=tol * 10, atol=0)
            assert len(set(ids[x, :m].tolist())) == m
        # aliased batch: the query's own row is not a candidate
        ida, dda, _ = ix.query(tp[:50].contiguous(), alias=True)
        wi, wd = _brute(sd, pts, pts[:50], _codes_of(eng, tp[:50].contiguous(), T), k, alias_ids=np.arange(50))
        for x in range(50):
            m = len(wi[x])
            assert np.allclose(dda.cpu().numpy()[x, :m], wd[x], rtol=tol, atol=0) and x not in ida.cpu().numpy()[x].tolist()
        # and the switch goes back: parity mode again, bit for bit the reference's answer
        ix.set_fixed(False)
        ids0, dd0, _ = ix.query(ty)
        want = orc.query(sd, pts, y)
        assert np.array_equal(ids0.cpu().numpy().astype(np.uint64), want[0])
        assert np.array_equal(dd0.cpu().numpy().view(np.uint8), want[1].view(np.uint8))
        save.free()
    finally:
        ix.close()


def _codes_of(eng, ty, T):
    codes = torch.empty((ty.shape[0], T), dtype=torch.int32, device="cuda")
    with eng.use(None):
        eng.sh_codes(ty, 0, ty.shape[0], codes)
    torch.cuda.synchronize()
    return codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF


def test_fixed_mode_finds_what_the_parity_mode_cannot():
    """Queries = indexed points + a little noise: the true nearest neighbour is the point itself and it sits in the
    query's own bucket in most tries.  The reference's scrambled code read (Q2) looks into other queries' buckets."""
    orc, pts, tp, ix = _build("f32", 20000, 32, 10, 10, 777)
    try:
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        src = torch.arange(0, 2000, device="cuda")
        ty = (tp[src] + 0.01 * torch.randn((2000, 32), device="cuda", generator=g)).contiguous()
        ids_p, _, _ = ix.query(ty)
        ix.set_fixed(True)
        ids_f, _, _ = ix.query(ty)
        hit_p = float((ids_p[:, 0] == src).float().mean())
        hit_f = float((ids_f[:, 0] == src).float().mean())
        assert hit_f > 0.95 and hit_f > hit_p + 0.3, (hit_p, hit_f)
    finally:
        ix.close()
