import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import approximatenn_amd as A
from approximatenn_amd.sharded import HipEngine, ShardedQuery
from oracle import oracle_py as O
prec = sys.argv[1]
orc = O.CpuBackend(prec, "oracle")
O.srandom(321); orc.rand_norm_reset()
half = orc.gen_rand(700*32).reshape(700,32); pts = np.ascontiguousarray(np.concatenate([half,half])); y = orc.gen_rand(90*32).reshape(90,32)
O.srandom(17); o_ids,o_d,o_save = orc.precomp(pts,6,4)
want = orc.query(o_save, pts, y)
save = A.Save.from_dict(prec, o_save)
full = HipEngine(A.Index.from_save(save, torch.from_numpy(pts).cuda(), 0, 1400))
sh = [HipEngine(A.Index.from_save(save, torch.from_numpy(np.ascontiguousarray(pts[lo:hi])).cuda(), lo, hi)) for lo,hi in ((0,700),(700,1400))]
yt = torch.from_numpy(y).cuda()
codes = full.codes(yt)
for e in sh: assert torch.equal(e.codes(yt), codes)
allq = torch.arange(90, dtype=torch.int32, device="cuda")
fi, fd = full.stage1_rows(yt, False, codes, allq)
rows = [e.stage1_rows(yt, False, codes, allq) for e in sh]
torch.cuda.synchronize()
print("ids equal", [bool(torch.equal(r[0], fi)) for r in rows])
dd = torch.minimum(rows[0][1], rows[1][1])
print("min-reduced dists equal full:", bool(torch.equal(dd.view(torch.int64) if prec=="f64" else dd.view(torch.int32), fd.view(torch.int64) if prec=="f64" else fd.view(torch.int32))))
bad = (dd != fd) & ~(torch.isinf(dd) & torch.isinf(fd))
print("bad entries", int(bad.sum()))
if bad.any():
    q, j = [int(v[0]) for v in torch.nonzero(bad)[:1].T]
    print("first bad q,j", q, j, "id", int(fi[q,j]) & 0xFFFFFFFF, "full", float(fd[q,j]), "r0", float(rows[0][1][q,j]), "r1", float(rows[1][1][q,j]))
k=6
ti = torch.zeros((90,k), dtype=torch.int32, device="cuda"); td = torch.zeros((90,k), dtype=yt.dtype, device="cuda")
full.exact_select(1, fi.clone(), fd.clone(), allq, ti, td)
ti2 = torch.zeros_like(ti); td2 = torch.zeros_like(td)
sh[0].exact_select(1, rows[0][0].clone(), dd.clone(), allq, ti2, td2)
torch.cuda.synchronize()
print("stage1 top equal:", bool(torch.equal(ti, ti2)))
# whole sharded flow by hand
r2 = [e.stage2_rows(yt, False, ti2, td2) for e in sh]
f2 = full.stage2_rows(yt, False, ti, td)
dd2 = torch.minimum(r2[0][1], r2[1][1])
print("stage2 ids equal", [bool(torch.equal(r[0], f2[0])) for r in r2], "dists equal", bool(torch.equal(dd2, f2[1])))
oi = torch.zeros_like(ti); od = torch.zeros_like(td)
sh[1].exact_select(2, r2[1][0], dd2, None, oi, od)
torch.cuda.synchronize()
got = (oi.to(torch.int64) & 0xFFFFFFFF).cpu().numpy().astype(np.uint64)
print("final match", np.array_equal(got, want[0]))
