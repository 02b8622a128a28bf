"""Time ONE rank's share of a G-way point-sharded query step on a SINGLE GPU (owner protocol, sharded.py).

The index is re-sharded so that this device owns 1/G of the rows and the batch is G x 10k queries, exactly what
rank 0 of a G-GPU node would execute; the collectives are loop-back stand-ins (the other ranks' contributions are
padding / copies), so RESULTS ARE MEANINGLESS and communication time is NOT included -- only the per-rank kernel work
and the host-side orchestration are representative.  Used to project scaling before a multi-GPU node is available
(DESIGN.md section 4).

    python tools/emulate_rank.py [--points N] [--queries Q_per_rank] [--rccl] [--tune]

--rccl: every stand-in collective additionally pushes its payload through a real RCCL kernel (a one-rank NCCL process
group's all_to_all_single on the communicator's own stream), so that the placement of RCCL's workgroups beside the
saturating gather -- and the stream hand-offs around them -- are part of what is timed.  --tune: print
ShardedQuery.autotune()'s table (lanes in use, two-half issue, CUs the gathers leave free) for each world size.
"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import torch

import approximatenn_amd as A
from approximatenn_amd.sharded import ShardedQuery

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=10_000)
ap.add_argument("--worlds", type=str, default="1,2,4,8")
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--rccl", action="store_true")
ap.add_argument("--tune", action="store_true")
ap.add_argument("--schedules", type=str, default="",
                help="semicolon list of depth,split,reserve,pieces[,slots] settings to time (ShardedQuery.configure); "
                     "split 2 = stage-major issue order")
ap.add_argument("--slots", type=str, default="",
                help="comma list: also time the pipelined step with the gather as a persistent grid of that many waves per "
                     "SIMD (annhip_index_set_gather_slots; 0 = one workgroup per query)")
args = ap.parse_args()

real = None
if args.rccl:
    import torch.distributed as real
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    torch.cuda.set_device(0)
    opts = real.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    real.init_process_group("nccl", rank=0, world_size=1, pg_options=opts, device_id=torch.device("cuda", 0))


class Loopback:
    """Stand-in for torch.distributed: this process is rank 0 of `world`; nobody else exists."""

    class ReduceOp:
        MIN = "min"

    def __init__(self, world):
        self.world = world
        self._scratch = {}

    def _kernel(self, t):
        """The payload once through an RCCL kernel (one-rank all-to-all = a copy done by rcclGenericKernel)."""
        if real is None:
            return
        key = (t.numel() * t.element_size())
        if key not in self._scratch:
            self._scratch[key] = torch.empty(key, dtype=torch.uint8, device=t.device)
        real.all_to_all_single(self._scratch[key], t.contiguous().view(torch.uint8).view(-1))

    def barrier(self, group=None): pass

    def is_initialized(self): return True
    def get_world_size(self, g=None): return self.world
    def get_rank(self, g=None): return 0
    def get_backend(self, g=None): return "loopback"

    def all_gather_into_tensor(self, out, t, group=None):   # every rank "contributed" what this one did
        self._kernel(t)
        out.view(self.world, -1).copy_(t.reshape(1, -1).expand(self.world, -1))

    def all_to_all_single(self, out, t, group=None):        # own part arrives; the peers' parts are padding
        self._kernel(t)
        n = t.shape[0] // self.world
        if t.dtype == torch.int64:
            out.fill_((0x7F800000 << 32) | 0xFFFFFFFF)       # key(+inf, no id) of the float build
        else:
            out.fill_(float("inf"))
        out[:n].copy_(t[:n])

    def all_reduce(self, t, op=None, group=None):
        if t.is_cuda:
            self._kernel(t)


n, d, k, T = args.points, 128, 10, 10
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(12345)
points = torch.randn((n, d), device=dev, dtype=torch.float32, generator=gen)
ctypes.CDLL("libc.so.6").srandom(12345)
ix = A.Index.precomp(points, k, T)
base = None
for G in [int(g) for g in args.worlds.split(",")]:
    Q = args.queries * G
    ys = [torch.randn((Q, d), device=dev, dtype=torch.float32, generator=gen) for _ in range(4 + args.steps)]
    if G == 1:
        def run_all(batches):
            for y in batches:
                ix.query(y)
        label = "annhip_query"
    else:
        lo, hi = 0, n // G
        ix.reshard(points[lo:hi], lo, hi)
        sq = ShardedQuery(ix, Loopback(G), exchange="alltoall", lanes=args.lanes)
        lanes_used = args.lanes

        if args.tune:
            tuned = sq.autotune(ys[0], batches=10)
            for row in tuned["table"]:
                print("   tune G=%d: lanes %d  split %d  reserve %2d CUs  gather in %d -> %.3f ms/batch" %
                      (G, row["depth"], row["split"], row["reserve_cus"], row["pieces"], row["ms"]), flush=True)
            lanes_used = sq.depth
        host = {"submit": 0.0, "collect": 0.0, "n": 0}

        def run_all(batches):     # several batches in flight, as bench.py --gpus N drives it
            pend = []
            for y in batches:
                if len(pend) == lanes_used:
                    t0 = time.perf_counter()
                    sq.collect(pend.pop(0))
                    host["collect"] += time.perf_counter() - t0
                t0 = time.perf_counter()
                pend.append(sq.submit(y))
                host["submit"] += time.perf_counter() - t0
                host["n"] += 1
            while pend:
                sq.collect(pend.pop(0))
        label = "submit/collect, %d lanes" % lanes_used
    if G > 1 and args.schedules:
        for sch in args.schedules.split(";"):
            c = [int(v) for v in sch.split(",")]
            sq.configure(*c)
            keep, lanes_used = lanes_used, sq.depth
            run_all(ys[:8])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_all(ys[4:])
            torch.cuda.synchronize()
            print("   G=%d schedule %s: per-rank step %.3f ms" % (G, sch, (time.perf_counter() - t0) / args.steps * 1e3), flush=True)
            lanes_used = keep
        fin = os.environ.get("EMUL_FINAL_SCHEDULE")
        if fin:
            sq.configure(*[int(v) for v in fin.split(",")])
            lanes_used = sq.depth
        else:
            sq.configure(lanes_used, 1, 0, 1, 0)
    if G > 1 and args.slots:
        for sl in [int(v) for v in args.slots.split(",")]:
            sq.configure(lanes_used, sq.split, sq.reserve_cus, sq.pieces, sl)
            run_all(ys[:4])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_all(ys[4:])
            torch.cuda.synchronize()
            print("   G=%d gather slots %d waves/SIMD: per-rank step %.3f ms" % (G, sl, (time.perf_counter() - t0) / args.steps * 1e3),
                  flush=True)
        sq.configure(lanes_used, sq.split, sq.reserve_cus, sq.pieces, int(os.environ.get("EMUL_FINAL_SLOTS", "0")))
    run_all(ys[:4])
    torch.cuda.synchronize()
    if G > 1:
        host.update(submit=0.0, collect=0.0, n=0)
    t0 = time.perf_counter()
    run_all(ys[4:])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if base is None:
        base = Q / dt
    print("G=%d  Q=%6d  per-rank step %.3f ms (%s)  -> %.2f M q/s aggregate (compute only), scaling %.2fx"
          % (G, Q, dt * 1e3, label, Q / dt / 1e6, (Q / dt) / base), flush=True)
    if G > 1:
        print("   host time per step: submit %.3f ms, collect (incl. waiting for the GPU) %.3f ms"
              % (host["submit"] / host["n"] * 1e3, host["collect"] / host["n"] * 1e3))
    if G > 1:   # the same step strictly serial (one lane), with CUDA events between the stages
        sq1 = ShardedQuery(ix, Loopback(G), exchange="alltoall", lanes=1)
        e, L = sq1.eng, sq1._lanes[0]
        y = ys[0]
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids, dd = sq1.query(y)
            torch.cuda.synchronize()
            serial = time.perf_counter() - t0
        qs = (Q + G - 1) // G

        def ev():
            x = torch.cuda.Event(enable_timing=True)
            x.record(torch.cuda.current_stream())
            return x
        with e.use(L.stream):
            marks = [("start", ev())]
            e.sh_codes(y, 0, qs, L.codes_slice); sq1._gather_cat(L.codes_all, L.codes_slice); marks.append(("codes + all-gather", ev()))
            e.sh_stage1(y, False, L.codes_all, L.keys, L.nvalid, L.nown); marks.append(("stage1 (own rows, all queries)", ev()))
            sq1._to_owner(L.keys_in, L.keys); marks.append(("all-to-all keys", ev()))
            e.sh_merge_finalize(G, Q, 0, qs, L.keys_in, L.nvalid, L.top_i, L.top_d); marks.append(("merge + finalize (owner)", ev()))
            sq1._gather_cat(L.top_all, L.top_i); marks.append(("all-gather top ids", ev()))
            e.sh_exact1_begin(y, False, L.codes_all, L.top_all, sq1.fcap, L.flist, L.xrows_i, L.xrows_d)
            sq1._all_min(L.xrows_d)
            e.sh_exact1_end(Q, 0, qs, sq1.fcap, L.flist, L.xrows_i, L.xrows_d, L.top_all, L.top_d_all, L.top_i, L.top_d)
            marks.append(("exact stage 1 of the flagged", ev()))
            e.sh_stage2(y, False, L.top_all, L.s2, L.flagged); marks.append(("stage-2 distances (all queries)", ev()))
            sq1._to_owner(L.s2_in, L.s2); marks.append(("all-to-all partial rows", ev()))
            e.sh_final(G, Q, 0, qs, L.top_i, L.top_d, L.s2_in, L.out_i_slice, L.out_d_slice); marks.append(("min + network (owner)", ev()))
            sq1._gather_cat(L.pack_all, L.pack.view(1, -1)); marks.append(("all-gather results", ev()))
        torch.cuda.synchronize()
        print("   serial step (1 lane) %.3f ms wall; %d flagged queries; stages by events:" % (serial * 1e3, int(L.flist[1].item())))
        for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
            print("     %-34s %.3f ms" % (b, ea.elapsed_time(eb)))
        print("     %-34s %.3f ms" % ("total", marks[0][1].elapsed_time(marks[-1][1])), flush=True)
