import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import oracle_py as O
from tests.cpu_engine import CpuShardEngine
from approximatenn_amd.sharded import ShardedQuery
prec = sys.argv[1]
orc = O.CpuBackend(prec, "oracle")
O.srandom(321); orc.rand_norm_reset()
half = orc.gen_rand(700*32).reshape(700,32); pts = np.ascontiguousarray(np.concatenate([half,half])); y = orc.gen_rand(90*32).reshape(90,32)
O.srandom(17); o_ids,o_d,o_save = orc.precomp(pts,6,4)
want = orc.query(o_save, pts, y)
# emulate 2 ranks in-process, lockstep (no dist): do the orchestration by hand
engs = [CpuShardEngine(o_save, pts, 0, 700, prec), CpuShardEngine(o_save, pts, 700, 1400, prec)]
yt = torch.from_numpy(y)
codes = engs[0].codes(yt)
loc = [e.stage1_local(yt, False, codes) for e in engs]
class FakeSQ(ShardedQuery):
    def __init__(self): self.dist=True; self.world=2; self.group=None; self._stage_via_cpu=False
    def _all_gather(self, t): return self._g
sq = FakeSQ()
# merge
if prec == "f32":
    keys = [ (cd.view(torch.int32).to(torch.int64) << 32) | (ci.to(torch.int64) & 0xFFFFFFFF) for cd,ci,_ in loc]
    sq._g = keys; 
    class D: pass
    md, mi = ShardedQuery._merge(sq, loc[0][0], loc[0][1])
else:
    # two gathers in sequence: emulate by swapping _g
    calls = [[l[0] for l in loc], [l[1] for l in loc]]
    it = iter(calls)
    sq._all_gather = lambda t: next(it)
    md, mi = ShardedQuery._merge(sq, loc[0][0], loc[0][1])
top_i, top_d, fl = engs[0].finalize(md, mi, loc[0][2])
print("flagged", len(fl), "of", len(y))
rows = [e.stage1_rows(yt, False, codes, fl) for e in engs]
dd = torch.minimum(rows[0][1], rows[1][1])
engs[0].exact_select(1, rows[0][0], dd, fl, top_i, top_d)
r2 = [e.stage2_rows(yt, False, top_i, top_d) for e in engs]
dd2 = torch.minimum(r2[0][1], r2[1][1])
oi = torch.empty_like(top_i); od = torch.empty_like(top_d)
engs[0].exact_select(2, r2[0][0], dd2, None, oi, od)
got = (oi.to(torch.int64) & 0xFFFFFFFF).numpy().astype(np.uint64)
print("match", np.array_equal(got, want[0]), "rows differing", int(np.sum((got != want[0]).any(axis=1))))
