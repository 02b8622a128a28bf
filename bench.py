#!/usr/bin/env python3
"""bench.py -- queries/s of the query() hot path on MI355X, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], "cfg3"): N=10M points, d=128, k=10, tries=10, float, Q=10k queries per GPU per
step.  Data: iid N(0,1) by Box-Muller on libc random() seeded with --seed -- the reference drivers' own stream
(time_results.c:10-13,94,103; randNorm.c:9-21): points first, then precomp's rotation draws, then one draw per query
batch.  --data randn uses torch.randn on the device instead (same distribution, seconds faster).
One step = one query() batch: hash codes -> candidate gather + squared L2 + top-k selection (+ exact fallback) ->
neighbour-of-neighbour refinement -> ids/distances, inputs and index resident in HBM.

N > 1: the point rows are sharded across the ranks (each GPU gathers only rows it owns); per-shard top-(k+1)
candidates travel to each query's owner rank by an RCCL all-to-all over xGMI and are merged there, the owners' top-k
ids are all-gathered, partial stage-2 rows go back to the owners the same way, results are all-gathered
(approximatenn_amd/sharded.py).  The batch grows with N (Q = 10k x N queries per step, every rank sees all of them),
so per-GPU gather work stays fixed: "scaling": "weak".  value = total queries / max-over-ranks time.  The extra object
"strong" reports the same job with the batch FIXED at 10k queries in total.  The index is built by all ranks together
(precomp_sharded: every rank holds all rows during the build, the distance passes are dealt out by bucket).

The JSON line also carries
  roofline     : the dominant kernel (stage1_select) priced at its ALGORITHMIC bytes / HIP-event duration
  cpu_baseline : the reference's own query_cpu (oracle/_ref) or the oracle (CPU restatement), 1 core, timed on a bounded
                 sample of the same workload on the same index, on rank 0 at N=1 only -- and compared with the GPU's
                 results (parity).
  host_api     : the same workload through the reference's host-pointer ABI (query_gpu, ann.h:61-62) and through the
                 pipelined host API (annhip_stream_*), PCIe included -- reported beside `value`, never as `value`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", dest="n", type=int, default=10_000_000)
    ap.add_argument("--dim", dest="d", type=int, default=128)
    ap.add_argument("--knn", dest="k", type=int, default=10)
    ap.add_argument("--tries", type=int, default=10)
    ap.add_argument("--queries", dest="q", type=int, default=10_000, help="queries per GPU per step")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--data", choices=["randnorm", "randn"], default="randnorm",
                    help="randnorm = the reference drivers' Box-Muller stream on libc random() (host, ~20 s for cfg3's "
                         "points); randn = torch.randn on the device")
    ap.add_argument("--streams", type=int, default=1,
                    help="single GPU: 1 = strictly serial steps (default; per-launch kernel times are meaningful); "
                         "N > 1 = independent batches alternate over N HIP streams/workspaces, so the latency-bound tail "
                         "of one step and the workgroup tail of its gather hide under the next step's gather")
    ap.add_argument("--no-overlap-extra", action="store_true",
                    help="skip the extra 2-stream throughput measurement reported under 'overlap'")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32", help="f64 = the reference's stock double build")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-pointer (query_gpu / annhip_stream) extra")
    ap.add_argument("--fixed-mode-timing", action="store_true",
                    help="also time one full batch in the opt-in fixed mode (recall_sample.fixed_mode.ms_per_step)")
    ap.add_argument("--no-strong-extra", action="store_true", help="N > 1: skip the fixed-batch (strong scaling) extra")
    ap.add_argument("--no-replicas-extra", action="store_true",
                    help="N > 1: skip the query-sharded (every GPU holds all rows) extra reported under 'replicas'")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--tune-seconds", type=float, default=20.0,
                    help="N > 1: wall-clock budget of the in-place schedule measurement before the warm-up (0 = keep the "
                         "pinned schedule)")
    return ap.parse_args()


def describe(n, d, k, T, Q, dtype):
    """Workload label built from the actual sizes; the cfgN tag only when they are BASELINE.json's."""
    word = "float" if dtype == "f32" else "double"
    tag = {(100_000, 32, 10, "f32"): "cfg1", (1_000_000, 64, 10, "f32"): "cfg2", (10_000_000, 128, 10, "f32"): "cfg3",
           (40_000_000, 128, 10, "f32"): "cfg4", (10_000_000, 256, 100, "f64"): "cfg5"}.get((n, d, k, dtype))

    def short(v):
        return "%dM" % (v // 1_000_000) if v % 1_000_000 == 0 else "%dk" % (v // 1000) if v % 1000 == 0 else str(v)
    return tag, "N=%s d=%d k=%d Q=%s %s" % (short(n), d, k, short(Q), word), \
        "%sN=%d d=%d k=%d tries=%d Q=%d/step %s" % (tag + ": " if tag else "", n, d, k, T, Q, word)


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a FRESH child process -- nothing in this
    process has touched the GPU (torch is not even imported yet), and the child is a child, never an exec -- relay its
    output (the one JSON line included) and return its exit status."""
    import socket
    import subprocess
    with socket.socket() as so:                 # a free rendezvous port on the loop-back interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # this pool's driver: dmabuf IPC only (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    for out_line in child.stdout:
        sys.stdout.write(out_line)
        sys.stdout.flush()
    return child.wait()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    import approximatenn_amd as A
    from approximatenn_amd._lib import park_random, random_state_restore, random_state_snapshot

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d inside a job of %d ranks: start it as `python bench.py --gpus N` (it launches its own "
                 "ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus" % (args.gpus, world))
    shared_gpu = os.environ.get("ANN_BENCH_SHARED_GPU") == "1"  # rehearsal: all ranks on GPU 0, gloo collectives
    rehearse_rccl = os.environ.get("ANN_SHARD_FORCE_DIST") == "1" and "RANK" in os.environ  # 1 rank, real RCCL calls
    dev_index = 0 if shared_gpu else local_rank
    os.environ["ANN_HIP_DEVICE"] = str(dev_index)
    if not shared_gpu and dev_index >= torch.cuda.device_count():
        sys.exit("rank %d: no GPU %d on this node (%d visible)" % (rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    sharded = world > 1 or rehearse_rccl
    rccl = None
    if sharded:
        if shared_gpu:
            dist.init_process_group(backend="gloo")
        else:  # "nccl" is RCCL on ROCm.  Its kernels run beside the other batch's gather, which would otherwise take
            # every freed wave slot first: give the communicator's stream the high priority the batches' streams have.
            import datetime
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            # a rank that dies must not leave the others waiting for the default 10 minutes
            dist.init_process_group(backend="nccl", pg_options=opts, device_id=device, timeout=datetime.timedelta(seconds=240))
        # who is really in this job: backend, communicator size and the physical devices behind the ranks
        props = torch.cuda.get_device_properties(dev_index)
        mine = {"rank": rank, "device": dev_index, "uuid": str(getattr(props, "uuid", "")), "name": props.name}
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        distinct = len({(e["device"], e["uuid"]) for e in everyone})
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001  (version query only; no collective involved)
            ver = None
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver,
                "distinct_devices": distinct, "device_name": props.name, "high_priority_stream": not shared_gpu}
        if dist.get_world_size() != args.gpus or (not shared_gpu and distinct != args.gpus):
            sys.exit("rank %d: --gpus %d but the communicator has %d ranks on %d distinct devices"
                     % (rank, args.gpus, dist.get_world_size(), distinct))
    coll_dev = "cpu" if shared_gpu else device      # where the small control tensors of the collectives live

    n, d, k, T = args.n, args.d, args.k, args.tries
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    Q = args.q * world  # weak scaling: the batch grows with the number of shards
    libc = ctypes.CDLL("libc.so.6")
    nbatches = args.warmup + args.steps

    def bcast(t):
        """rank 0's tensor to every rank (device tensor; staged through the host for the gloo rehearsal)."""
        if not sharded or dist.get_world_size() == 1:
            return t
        if shared_gpu:
            c = t.cpu()
            dist.broadcast(c, src=0)
            t.copy_(c)
        else:
            dist.broadcast(t, src=0)
        return t

    # ---- synthetic data + index.  The libc random() stream belongs to the workload (points -> precomp's rotations ->
    #      one draw per batch, time_results.c:94-103).  RANK 0 ALONE generates the points and the batches and broadcasts
    #      them (N ranks x 1.3 G Box-Muller values on the same host cores would take N times as long and N x 5 GB of
    #      host memory); the stream's state after the points travels too, so that every rank's precomp draws the SAME
    #      rotations from it (Q12).  The HIP runtime draws from random() whenever it feels like it (observed at
    #      initialisation), so the stream is parked while torch talks to the GPU -- the library does the same inside.
    with park_random():
        torch.zeros(1, device=device)           # the runtime is fully up before the stream is seeded
        torch.cuda.synchronize()
    libc.srandom(args.seed)
    host_pts = None
    t0 = time.time()
    gen = None
    if args.data == "randnorm":
        if rank == 0:
            host_pts = A.synth_randnorm(n * d, args.dtype, reset=True).reshape(n, d)   # time_results.c:94
        with park_random():
            points = torch.from_numpy(host_pts).to(device) if rank == 0 else torch.empty((n, d), dtype=tdt, device=device)
            torch.cuda.synchronize()
    else:
        with park_random():
            gen = torch.Generator(device=device)
            gen.manual_seed(args.seed)
            points = torch.randn((n, d), device=device, dtype=tdt, generator=gen) if rank == 0 else \
                torch.empty((n, d), dtype=tdt, device=device)
            torch.cuda.synchronize()
    if sharded:
        snap = torch.zeros(256, dtype=torch.uint8)
        if rank == 0:
            state = random_state_snapshot()
            snap[: len(state)] = torch.frombuffer(bytearray(state), dtype=torch.uint8)
            snap[255] = len(state)
        with park_random():
            bcast(points)
            snap = snap.to(coll_dev)
            bcast(snap)
            torch.cuda.synchronize()
            snap = snap.cpu()
        if rank != 0:
            random_state_restore(bytes(snap[: int(snap[255])].tolist()))
    datagen_s = time.time() - t0
    t0 = time.time()
    if sharded:                                 # distance passes dealt to the ranks by bucket; three collectives
        from approximatenn_amd.sharded import precomp_sharded, same_everywhere
        ix = precomp_sharded(points, k, T, dist=dist)
    else:
        ix = A.Index.precomp(points, k, T)      # draws its rotations from the same random() stream (Q12)
    with park_random():
        torch.cuda.synchronize()
        precomp_s = time.time() - t0
        ix.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.data == "randnorm":                 # one genRand per batch, in order (time_results.c:103)
        if rank == 0:
            host_batches = [A.synth_randnorm(Q * d, args.dtype).reshape(Q, d) for _ in range(nbatches)]
        with park_random():
            batches = [torch.from_numpy(host_batches[i]).to(device) if rank == 0 else torch.empty((Q, d), dtype=tdt, device=device)
                       for i in range(nbatches)]
        host_batches = None
    else:
        with park_random():
            batches = [torch.randn((Q, d), device=device, dtype=tdt, generator=gen) if rank == 0 else
                       torch.empty((Q, d), dtype=tdt, device=device) for _ in range(nbatches)]
    if sharded:
        with park_random():
            for b in batches:
                bcast(b)
            torch.cuda.synchronize()
            # Every rank must now hold the same index and the same batches: results depend on both, and ranks that
            # disagree would answer wrongly or hang in mismatched collectives.  Checked, not assumed (exit status 3).
            sums = same_everywhere(dist, [ix.checksum(), A.checksum(batches[0], args.dtype), A.checksum(batches[-1], args.dtype)],
                                   "index / batch checksums", device=coll_dev)
            rccl["index_checksum"] = "%016x" % sums[0]

    runner = None
    pin = os.environ.get("ANN_SHARD_TUNE")
    if sharded:
        from approximatenn_amd.sharded import ShardedQuery
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        with park_random():
            shard = points[lo:hi].clone()
            ix.reshard(shard, lo, hi)
            if args.no_replicas_extra:
                del points
            torch.cuda.empty_cache()
        runner = ShardedQuery(ix, dist, lanes=7)   # exchange agreed on by all ranks at start-up (all-to-all, else all-gather)
        # Up to three batches in flight: the exchanges of batch i run under the gathers of i+1, i+2.  The job starts from
        # ONE pinned schedule (ShardedQuery.PINNED); lanes, issue order and launches per gather are then measured in
        # place within --tune-seconds of wall clock (untimed batches before the warm-up, every decision agreed on by
        # all ranks; no result bit depends on it).  ANN_SHARD_TUNE=depth,split,reserve,pieces pins a schedule,
        # ANN_SHARD_TUNE=off keeps the pinned default without measuring.
        if pin and pin != "off":
            runner.configure(*[int(v) for v in pin.split(",")])
        elif pin == "off" or args.tune_seconds <= 0:
            runner.configure(*ShardedQuery.PINNED)
        else:
            runner.autotune(batches[0], budget_s=args.tune_seconds)

        def run_steps(ys):
            runner.pump(ys)
    else:
        ns = max(1, args.streams)
        out_ids = [torch.empty((Q, k), dtype=torch.int64, device=device) for _ in range(ns)]
        out_d = [torch.empty((Q, k), dtype=tdt, device=device) for _ in range(ns)]
        lanes = [(ix.workspace(), torch.cuda.Stream(device=device)) for _ in range(ns)] if ns > 1 else None

        def run_steps(ys):
            for i, y in enumerate(ys):
                if ns == 1:
                    ix.query(y, out_ids=out_ids[0], out_dists=out_d[0])
                else:
                    j = i % ns
                    ix.query(y, out_ids=out_ids[j], out_dists=out_d[j], ws=lanes[j][0], stream=lanes[j][1])

    def barrier():
        if sharded:
            dist.barrier()

    def timed(ys, run=None):
        """barrier + synchronize, run, synchronize + barrier; max over ranks."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        (run or run_steps)(ys)
        submit = time.perf_counter() - t0
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        if sharded:
            tt = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, submit

    run_steps(batches[:args.warmup])
    torch.cuda.synchronize()
    ix.stats(reset=True)
    # Inside the timed region: ONLY the HIP-event pair around each stage-1 launch (the roofline's kernel time, measured on
    # the steps that count).  Every event record costs ~5 us of stream time, so the per-stage marks (8 more events per
    # step) and the gathered-row statistics (one more small kernel per step) come from a SEPARATE, untimed pass over the
    # same K batches right afterwards -- same batches, same rows, same flagged queries.
    ix.profile(2 if os.environ.get("ANN_BENCH_NO_EVENTS") != "1" else 0)
    elapsed, submit_s = timed(batches[args.warmup:])
    st = ix.stats()
    ix.profile(True)
    ix.stats(reset=True)
    run_steps(batches[args.warmup:])
    torch.cuda.synchronize()
    st_full = ix.stats()
    stage_ms = ix.stage_ms() if not sharded else None
    ix.profile(False)
    if not st["s1_launches"]:  # ANN_BENCH_NO_EVENTS=1: no kernel time from the timed region; report the separate pass's
        st["s1_launches"], st["s1_ms"] = st_full["s1_launches"], st_full["s1_ms"]
    for key in ("s1_rows", "other_rows", "queries", "exact_queries", "tie_queries"):
        st[key] = st_full[key]

    # ---- roofline of the dominant kernel (stage1_select): algorithmic bytes / HIP-event time
    launches = max(st["s1_launches"], 1.0)
    v1 = st["s1_rows"] / max(st["queries"], 1.0)            # rows THIS device gathered per query
    kern_ms = st["s1_ms"] / launches
    bytes_per_query = v1 * d * esz + ix.P1 * 4 + d * esz + T * 4 + (k + 1) * (esz + 4)
    bytes_per_launch = bytes_per_query * Q
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "stage1_select_kernel<%d>" % d, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "rows_gathered_per_query": round(v1, 1)}
    if sharded:
        roofline["note"] = "rank 0's kernel; with several batches in flight it overlaps the other batches's small kernels"

    # PMC counters cannot be read in-process: for the default workload the figure is the committed rocprofv3 --pmc
    # measurement of this same command (profiles/traffic.json says which run); any other workload reports null.
    tag, wl_short, wl_long = describe(n, d, k, T, Q, args.dtype)
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if not sharded and (n, d, k, T, Q, args.dtype) == (10_000_000, 128, 10, 10, 10_000, "f32"):
            roofline["traffic"] = tr["traffic_bytes_per_launch"]
            roofline["traffic_source"] = "static, from the committed rocprofv3 --pmc passes: " + tr["source"]
    except (OSError, ValueError, KeyError):
        pass

    value = Q * args.steps / elapsed
    line = {"metric": "queries/sec, %s (query(): hash + candidate gather + L2 + top-k + refine)" % wl_short,
            "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": wl_long,
                       "data_generator": ("iid N(0,1), Box-Muller on libc random() -- the reference drivers' stream "
                                          "(randNorm.c:9-21), srandom(%d)" % args.seed) if args.data == "randnorm"
                       else "iid N(0,1), torch.randn on the device, seed %d" % args.seed,
                       "points_sharding": "rows/%d" % world,
                       "streams": (max(1, args.streams) if not sharded else runner.depth), "d_short": ix.d_short, "L1": ix.L1, "P1": ix.P1,
                       "L2": ix.L2, "P2": ix.P2, "sum_par_maxes": ix.sum_pm, "datagen_s": round(datagen_s, 2),
                       "precomp_s": round(precomp_s, 2),
                       "exact_path_queries_per_step": round(st["exact_queries"] / args.steps, 2),
                       "tie_path_queries_per_step": round(st.get("tie_queries", 0) / args.steps, 2),
                       "host_submit_ms_per_step": round(submit_s / args.steps * 1e3, 4)},
            "roofline": roofline}
    if stage_ms:
        line["config"]["stage_ms_per_step"] = {k_: round(v / args.steps, 4) for k_, v in stage_ms.items()}
        line["config"]["stage_ms_source"] = ("a separate untimed pass over the same batches with 8 more HIP events per step (~5 us "
                                             "each); the timed region carries only the event pair around stage 1")
    if runner is not None:
        line["config"]["exchange"] = runner.exchange
        line["config"]["rccl"] = rccl
        line["config"]["queries_per_step_total"] = Q
        line["config"]["batches_in_flight"] = runner.depth
        line["config"]["schedule"] = runner.tuned or {"depth": runner.depth, "split": runner.split,
                                                      "reserve_cus": runner.reserve_cus, "pieces": runner.pieces,
                                                      "pinned": True}
        line["config"]["note"] = ("steps are pipelined (%d batches in flight, all K steps complete inside the timed "
                                  "region); the like-for-like single-GPU figure is the N=1 line's `overlap.value` (2 batches "
                                  "in flight), its `value` is strictly serial" % runner.depth)

    # ---- N > 1 extra: strong scaling -- the batch FIXED at --queries in total (10k), sharded the same way
    if sharded and not args.no_strong_extra:
        Qs = args.q
        ys = [b[:Qs].contiguous() for b in batches]
        if not pin and args.tune_seconds > 0:
            runner.autotune(ys[0], budget_s=args.tune_seconds / 2)   # a batch this small wants its own schedule
        run_steps(ys[:args.warmup])
        el_s, _ = timed(ys[args.warmup:])
        line["strong"] = {"queries_per_step_total": Qs, "value": round(Qs * args.steps / el_s, 1), "unit": "queries/s",
                          "ms_per_step": round(el_s / args.steps * 1e3, 4),
                          "schedule": {k_: v for k_, v in (runner.tuned or {}).items() if k_ != "table"},
                          "note": "same job with the batch fixed at %d queries in total (results differ from the weak "
                                  "run's: they depend on the batch, SURVEY Q2)" % Qs}

    # ---- N > 1 extra: the other way to split the job: every GPU keeps ALL rows (cfg4's 20 GB fit each
    #      288-GB GPU) and answers 1/N of the queries of every batch; one all-gather of the hash codes is the only exchange
    #      before the results.  Not the mandated row-sharded line -- the bound that line should be judged against.
    if sharded and not args.no_replicas_extra:
        from approximatenn_amd.sharded import ReplicaQuery
        with park_random():
            ix.reshard(points, 0, n)            # all rows again
            rq = ReplicaQuery(ix, dist, lanes=2, gather=True)
        rq.pump(batches[:max(args.warmup, 2)])
        el_r, _ = timed(batches[args.warmup:], rq.pump)
        line["replicas"] = {"queries_per_step_total": Q, "value": round(Q * args.steps / el_r, 1), "unit": "queries/s",
                            "ms_per_step": round(el_r / args.steps * 1e3, 4), "batches_in_flight": rq.depth,
                            "note": "query-sharded: every GPU holds all %d rows and answers Q/N queries of each batch; "
                                    "exchanges per step: all-gather of the hash codes (%d B/query), all-gather of the "
                                    "results; same results" % (n, 4 * T)}

    # ---- extra: the same K steps with consecutive batches overlapped on two streams (annhip_query_on); reported
    #      beside `value`, never instead of it: per-launch kernel times are not meaningful while gathers overlap
    if not sharded and rank == 0 and max(1, args.streams) == 1 and not args.no_overlap_extra:
        lanes2 = [(ix.workspace(), torch.cuda.Stream(device=device)) for _ in range(2)]
        o_ids = [torch.empty((Q, k), dtype=torch.int64, device=device) for _ in range(2)]
        o_d = [torch.empty((Q, k), dtype=tdt, device=device) for _ in range(2)]
        for i in range(nbatches):
            if i == args.warmup:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            ix.query(batches[i], out_ids=o_ids[i % 2], out_dists=o_d[i % 2], ws=lanes2[i % 2][0], stream=lanes2[i % 2][1])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        line["overlap"] = {"streams": 2, "value": round(Q * args.steps / dt, 1), "unit": "queries/s",
                           "ms_per_step": round(dt / args.steps * 1e3, 4),
                           "note": "consecutive batches on 2 HIP streams/workspaces; same work, same results"}
    # ---- quality of the answers (not part of the metric): exact-rank recall of a 512-query sample, by GPU brute force
    if not sharded and rank == 0:
        qs = min(512, Q)
        g_ids, _, _ = ix.query(batches[0][:qs].contiguous())
        rk = A.recall_ranks(points, batches[0][:qs].contiguous(), g_ids)
        line["config"]["recall_sample"] = {kk: round(v, 4) for kk, v in A.recall_summary(rk, k).items()}
        line["config"]["recall_sample"]["queries"] = qs
        # the same sample in the opt-in fixed mode (own hash codes, every candidate slot; include/ann_hip.h): not the
        # reference's results and never part of `value` -- evidence of what the two accidents cost
        ix.set_fixed(True)
        f_ids, _, _ = ix.query(batches[0][:qs].contiguous())
        f_ms = None
        if args.fixed_mode_timing:      # a full batch too (off by default: it would sit among the profiled launches)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ix.query(batches[0])
            torch.cuda.synchronize()
            f_ms = round((time.perf_counter() - t1) * 1e3, 4)
        ix.set_fixed(False)
        f_ids = torch.where(f_ids >= n, torch.zeros_like(f_ids), f_ids)     # (n, +inf) fillers: any wrong id will do
        rk = A.recall_ranks(points, batches[0][:qs].contiguous(), f_ids)
        line["config"]["recall_sample"]["fixed_mode"] = dict({kk: round(v, 4) for kk, v in A.recall_summary(rk, k).items()},
                                                             ms_per_step=f_ms)
    # ---- host-pointer API + CPU baseline share one exported save_t and one host copy of the points
    if not sharded and rank == 0 and not (args.no_cpu_baseline and args.no_host_api):
        if host_pts is None:
            host_pts = points.cpu().numpy()
        save = ix.export()
        if not args.no_host_api:
            line["host_api"] = host_api(args, ix, save, host_pts, batches, np, torch, A)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, ix, save, host_pts, batches[0], np, torch)
        save.free()
    if rank == 0:
        print(json.dumps(line), flush=True)
    ix.close()
    if sharded:
        dist.destroy_process_group()


def host_api(args, ix, save, host_pts, batches, np, torch, A):
    """The metric's workload through the reference's own ABI: query(save, points, ycnt, y, &dists) with HOST pointers
    in and malloc'd host arrays out (ann.h:61-62, timed like time_results.c:104-109: around the call only), warm
    residency cache, the cold first call apart; and the same batches through annhip_stream_* (3 lanes, pinned staging)."""
    Q, k = batches[0].shape[0], ix.k
    hy = [np.ascontiguousarray(b.cpu().numpy()) for b in batches[: max(4, min(len(batches), 10))]]
    lib = A._lib.load(args.dtype)
    lib.annhip_cache_clear()
    t0 = time.perf_counter()
    ids0, dd0 = A.query(save, host_pts, hy[0])          # cold: uploads points, tables, graph
    cold = time.perf_counter() - t0
    g_ids, g_d, _ = ix.query(batches[0])
    torch.cuda.synchronize()
    same = bool(np.array_equal(g_ids.cpu().numpy().astype(np.uint64), ids0) and
                np.array_equal(g_d.cpu().numpy().view(np.uint8), dd0.view(np.uint8)))
    A.query(save, host_pts, hy[1])
    t0 = time.perf_counter()
    for y in hy[1:]:
        A.query(save, host_pts, y)
    warm = (time.perf_counter() - t0) / (len(hy) - 1)
    lib.annhip_cache_clear()
    hs = ix.host_stream(Q, lanes=3)
    reps = hy * 3
    for _ in hs.map(hy[:3]):
        pass
    t0 = time.perf_counter()
    for _ in hs.map(reps):
        pass
    piped = (time.perf_counter() - t0) / len(reps)
    hs.close()
    return {"query_gpu": {"value": round(Q / warm, 1), "unit": "queries/s", "ms_per_call": round(warm * 1e3, 4),
                          "cold_first_call_s": round(cold, 3), "same_results_as_resident_path": same},
            "stream_3_lanes": {"value": round(Q / piped, 1), "unit": "queries/s", "ms_per_batch": round(piped * 1e3, 4)},
            "note": "host pointers in, host arrays out, PCIe included; never `value`"}


def cpu_baseline(args, ix, save, host_pts, y_dev, np, torch):
    """CPU column on the GPU box's host cores, single thread, on a bounded sample of the same batch and the same
    (GPU-built) index, with the results compared against the GPU's.  kind = "reference": the reference's own query_cpu
    (oracle/_ref, compiled from /root/reference in the authoring container; it materialises Q*L1*d values, so the
    sample is small); otherwise kind = "port": the oracle (oracle/ann_oracle.c).  The oracle's rate is reported too."""
    from oracle import oracle_py as O
    arrays = save.to_dict()
    npdt = np.float32 if args.dtype == "f32" else np.float64

    def timed(backend, qs):
        y = np.ascontiguousarray(y_dev[:qs].cpu().numpy())
        hs = O.HostSave(arrays, args.dtype)
        t0 = time.perf_counter()
        ids, dd = backend.query(hs, host_pts, y)
        dt = time.perf_counter() - t0
        g_ids, g_d, _ = ix.query(y_dev[:qs].contiguous())
        torch.cuda.synchronize()
        same = {"ids_bit_exact": bool(np.array_equal(g_ids.cpu().numpy().astype(np.uint64), ids)),
                "dists_bit_exact": bool(np.array_equal(g_d.cpu().numpy().view(np.uint8), dd.view(np.uint8)))}
        return qs / dt, dt, same

    def sized(backend, budget_s, probe, cap, derate=1.0):
        rate, _, _ = timed(backend, probe)
        qs = int(max(probe, min(cap, len(y_dev), rate * budget_s * derate)))
        return (qs,) + timed(backend, qs)

    orc = O.CpuBackend(args.dtype, "oracle")
    o_qs, o_rate, o_dt, o_same = sized(orc, args.cpu_seconds, 64, len(y_dev))
    out = {"value": round(o_rate, 2), "unit": "queries/s", "cores": 1, "kind": "port",
           "sample": "%d-query batch of the same workload on the GPU-built index (oracle, 1 thread, %.1f s)" % (o_qs, o_dt),
           "host_cores_available": os.cpu_count(), "parity_on_sample": o_same}
    if O.have_ref():
        # memory of the reference's diffs tensor: Q * L1 * d values -- keep it under ~8 GB
        cap = max(8, int(8e9 / (ix.L1 * ix.d * np.dtype(npdt).itemsize)))
        ref = O.CpuBackend(args.dtype, "ref")
        # its per-query cost grows with the batch (strided z-outer loops, ocl2c.h:18-22): derate the probe's estimate
        r_qs, r_rate, r_dt, r_same = sized(ref, args.cpu_seconds, 32, min(cap, 1000), derate=0.35)
        out = {"value": round(r_rate, 2), "unit": "queries/s", "cores": 1, "kind": "reference",
               "sample": "%d-query batch of the same workload on the GPU-built index (the reference's query_cpu, "
                         "oracle/_ref, 1 thread, %.1f s)" % (r_qs, r_dt),
               "host_cores_available": os.cpu_count(), "parity_on_sample": r_same,
               "oracle_port": {"value": round(o_rate, 2), "sample_queries": o_qs, "parity_on_sample": o_same}}
    return out


if __name__ == "__main__":
    main()
