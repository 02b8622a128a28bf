"""Time ONE rank's share of a G-way point-sharded query step on a SINGLE GPU.

The index is re-sharded so that this device owns 1/G of the rows and the batch is G x 10k queries, exactly what
rank 0 of a G-GPU node would execute; the collectives are loop-back stand-ins (the other ranks' candidates are
padding), so RESULTS ARE MEANINGLESS and communication time is NOT included -- only the per-rank kernel work is
representative.  Used to project scaling before a multi-GPU node is available.

    python tools/emulate_rank.py
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import torch
import approximatenn_amd as A
from approximatenn_amd.sharded import ShardedQuery

class Loopback:
    class ReduceOp: MIN = "min"
    def __init__(self, world, rank): self.world, self.rank = world, rank
    def is_initialized(self): return True
    def get_world_size(self, g=None): return self.world
    def get_rank(self, g=None): return self.rank
    def get_backend(self, g=None): return "loopback"
    def all_gather_into_tensor(self, out, t, group=None):
        o = out.view(self.world, -1)
        if t.dim() == 2 and t.shape[1] == 11:  # stage-1 candidates: the other ranks contribute padding, not copies
            if t.dtype == torch.float32: o.fill_(float("inf"))
            else: o.fill_(-1)
            o[0].copy_(t.reshape(-1))
        else:
            o.copy_(t.reshape(1, -1).expand(self.world, -1))
    def reduce_scatter_tensor(self, out, t, op=None, group=None):
        n = out.shape[0]; out.copy_(t[self.rank * n:(self.rank + 1) * n])
    def all_reduce(self, t, op=None, group=None): pass
    def all_gather(self, outs, t, group=None):
        for o in outs: o.copy_(t)

n, d, k, T = 10_000_000, 128, 10, 10
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(12345)
points = torch.randn((n, d), device=dev, dtype=torch.float32, generator=gen)
ctypes.CDLL("libc.so.6").srandom(12345)
ix = A.Index.precomp(points, k, T)
for G in (1, 2, 4, 8):
    Q = 10_000 * G
    ys = [torch.randn((Q, d), device=dev, dtype=torch.float32, generator=gen) for _ in range(6)]
    if G == 1:
        run = lambda y: ix.query(y)
    else:
        lo, hi = 0, n // G
        ix.reshard(points[lo:hi], lo, hi)
        sq = ShardedQuery(ix, Loopback(G, 0))
        assert sq.fast
        run = lambda y: sq.query(y)
    run(ys[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for y in ys[1:]: run(y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("G=%d  Q=%6d  per-rank step %.3f ms  -> %.2f M q/s aggregate (compute only), scaling %.2fx" % (G, Q, dt * 1e3, Q / dt / 1e6, (Q / dt) / (10_000 / 1.389e-3)))

# ---- segment timing of the G=8 step with CUDA events
import types
G = 8; Q = 80_000
ix.reshard(points[0:n // G], 0, n // G)
sq = ShardedQuery(ix, Loopback(G, 0))
e = sq.eng
y = torch.randn((Q, d), device=dev, dtype=torch.float32, generator=gen)
def ev():
    x = torch.cuda.Event(enable_timing=True); x.record(); return x
for rep in range(3):
    marks = [("start", ev())]
    qs = Q // G
    codes = sq._gather_stacked(e.codes(y[0:qs].contiguous())).reshape(-1); marks.append(("codes+gather", ev()))
    cd, ci, nv = e.stage1_local(y, False, codes); marks.append(("stage1_local", ev()))
    cd, ci = sq._merge(cd, ci); marks.append(("gather+merge", ev()))
    t0 = time.perf_counter(); top_i, top_d, fl = e.finalize(cd, ci, nv); tfin = time.perf_counter() - t0; marks.append(("finalize(sync)", ev()))
    if fl.shape[0]:
        ids, dd = e.stage1_rows(y, False, codes, fl); e.exact_select(1, ids, dd, fl, top_i, top_d)
    marks.append(("fallback(%d)" % fl.shape[0], ev()))
    ids2, dd2 = e.stage2_rows(y, False, top_i, top_d); marks.append(("stage2_rows", ev()))
    mine = torch.empty((qs, dd2.shape[1]), dtype=dd2.dtype, device=dev); sq.dist.reduce_scatter_tensor(mine, dd2)
    loc_i = torch.empty((qs, k), dtype=torch.int32, device=dev); loc_d = torch.empty((qs, k), dtype=torch.float32, device=dev)
    e.exact_select(2, ids2[0:qs].contiguous(), mine, None, loc_i, loc_d); marks.append(("rs+select", ev()))
    oi = sq._gather_stacked(loc_i); od = sq._gather_stacked(loc_d); marks.append(("gather out", ev()))
    torch.cuda.synchronize()
    if rep == 2:
        for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
            print("  %-18s %.3f ms" % (b, ea.elapsed_time(eb)))
        print("  total %.3f ms ; finalize host wall %.3f ms" % (marks[0][1].elapsed_time(marks[-1][1]), tfin * 1e3))
