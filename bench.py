#!/usr/bin/env python3
"""bench.py -- queries/s of the query() hot path on MI355X, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], "cfg3"): N=10M points, d=128, k=10, tries=10, float, Q=10k queries per
GPU per step, synthetic N(0,1) data generated on the device (no dataset exists for this path).  One step =
one query() batch: hash codes -> candidate gather + squared L2 + top-k selection (+ exact fallback) ->
neighbour-of-neighbour refinement -> ids/distances, inputs and index resident in HBM.

N > 1: the point rows are sharded across the ranks (each GPU gathers only rows it owns), per-shard top-(k+1)
candidates are all-gathered over RCCL and merged, stage-2 distance rows are min-all-reduced.  The batch grows
with N (Q = 10k x N queries per step, every rank sees all of them), so per-GPU gather work stays fixed:
"scaling": "weak".  value = total queries / max-over-ranks time.

The JSON line also carries
  roofline     : the dominant kernel (stage1_select) priced at its ALGORITHMIC bytes / HIP-event duration
  cpu_baseline : the oracle (CPU restatement, 1 core) timed on a bounded sample of the same workload on the
                 same index, on rank 0 at N=1 only -- and its results are compared with the GPU's (parity).
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
    ap.add_argument("--streams", type=int, default=1,
                    help="single GPU: 1 = strictly serial steps (default; per-launch kernel times are meaningful); "
                         "N > 1 = independent batches alternate over N HIP streams/workspaces, so the latency-bound tail "
                         "of one step and the workgroup tail of its gather hide under the next step's gather")
    ap.add_argument("--no-overlap-extra", action="store_true",
                    help="skip the extra 2-stream throughput measurement reported under 'overlap'")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32", help="f64 = the reference's stock double build")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import approximatenn_amd as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    shared_gpu = os.environ.get("ANN_BENCH_SHARED_GPU") == "1"  # rehearsal: all ranks on GPU 0, gloo collectives
    dev_index = 0 if shared_gpu else local_rank
    os.environ["ANN_HIP_DEVICE"] = str(dev_index)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        backend = "gloo" if shared_gpu else "nccl"  # "nccl" is RCCL on ROCm
        dist.init_process_group(backend=backend)

    n, d, k, T = args.n, args.d, args.k, args.tries
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    Q = args.q * world  # weak scaling: the batch grows with the number of shards
    libc = ctypes.CDLL("libc.so.6")

    # ---- synthetic data + index (identical on every rank: same seeds, deterministic build)
    gen = torch.Generator(device=device)
    gen.manual_seed(args.seed)
    points = torch.randn((n, d), device=device, dtype=tdt, generator=gen)
    libc.srandom(args.seed)
    torch.cuda.synchronize()
    t0 = time.time()
    ix = A.Index.precomp(points, k, T)
    torch.cuda.synchronize()
    precomp_s = time.time() - t0
    ix.set_stream(torch.cuda.current_stream().cuda_stream)
    batches = [torch.randn((Q, d), device=device, dtype=tdt, generator=gen)
               for _ in range(args.warmup + args.steps)]

    if world > 1:
        from approximatenn_amd.sharded import ShardedQuery
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        shard = points[lo:hi].clone()
        ix.reshard(shard, lo, hi)
        del points
        torch.cuda.empty_cache()
        runner = ShardedQuery(ix, dist)
        try:  # the lean collectives (all_gather_into_tensor / reduce_scatter MIN) first; plain ones if RCCL objects
            runner.query(batches[0])
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            if rank == 0:
                print("bench: fast collectives failed (%r); using all_gather/all_reduce" % (exc,), file=sys.stderr)
            runner = ShardedQuery(ix, dist, fast=False)
        step = lambda y: runner.query(y)
    else:
        ns = max(1, args.streams)
        out_ids = [torch.empty((Q, k), dtype=torch.int64, device=device) for _ in range(ns)]
        out_d = [torch.empty((Q, k), dtype=tdt, device=device) for _ in range(ns)]
        if ns == 1:
            step = lambda y: ix.query(y, out_ids=out_ids[0], out_dists=out_d[0])
        else:
            lanes = [(ix.workspace(), torch.cuda.Stream(device=device)) for _ in range(ns)]
            counter = [0]

            def step(y):
                j = counter[0] % ns
                counter[0] += 1
                ix.query(y, out_ids=out_ids[j], out_dists=out_d[j], ws=lanes[j][0], stream=lanes[j][1])

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(batches[i])
    torch.cuda.synchronize()
    ix.stats(reset=True)
    ix.profile(os.environ.get("ANN_BENCH_NO_EVENTS") != "1")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(batches[args.warmup + i])
    submit_s = time.perf_counter() - t0  # host time to enqueue K steps (the path is asynchronous on one GPU)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if not shared_gpu else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = ix.stats()
    stage_ms = ix.stage_ms() if world == 1 else None
    ix.profile(False)

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

    # PMC traffic cannot be read in-process; for the default workload it comes from the committed rocprofv3 passes
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if world == 1 and (n, d, k, T, Q, args.dtype) == (10_000_000, 128, 10, 10, 10_000, "f32"):
            roofline["traffic"] = tr["traffic_bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
    except (OSError, ValueError, KeyError):
        pass

    value = Q * args.steps / elapsed
    line = {"metric": "queries/sec, N=10M d=128 k=10 Q=10k float (query(): hash + candidate gather + L2 + top-k + refine)",
            "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%sN=%d d=%d k=%d tries=%d Q=%d/step float, randn points+queries"
                       % ("cfg3: " if (n, d, k, args.q, args.dtype) == (10_000_000, 128, 10, 10_000, "f32") else "", n, d, k, T, Q),
                       "points_sharding": "rows/%d" % world, "streams": (max(1, args.streams) if world == 1 else 1), "d_short": ix.d_short, "L1": ix.L1, "P1": ix.P1,
                       "L2": ix.L2, "P2": ix.P2, "sum_par_maxes": ix.sum_pm, "precomp_s": round(precomp_s, 2),
                       "exact_path_queries_per_step": round(st["exact_queries"] / args.steps, 2),
                       "host_submit_ms_per_step": round(submit_s / args.steps * 1e3, 4)},
            "roofline": roofline}
    if stage_ms:
        line["config"]["stage_ms_per_step"] = {k_: round(v / args.steps, 4) for k_, v in stage_ms.items()}

    # ---- extra: the same K steps with consecutive batches overlapped on two streams (annhip_query_on); reported
    #      beside `value`, never instead of it: per-launch kernel times are not meaningful while gathers overlap
    if world == 1 and rank == 0 and max(1, args.streams) == 1 and not args.no_overlap_extra:
        lanes = [(ix.workspace(), torch.cuda.Stream(device=device)) for _ in range(2)]
        o_ids = [torch.empty((Q, k), dtype=torch.int64, device=device) for _ in range(2)]
        o_d = [torch.empty((Q, k), dtype=tdt, device=device) for _ in range(2)]
        for i in range(args.warmup + args.steps):
            if i == args.warmup:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            ix.query(batches[i], out_ids=o_ids[i % 2], out_dists=o_d[i % 2], ws=lanes[i % 2][0], stream=lanes[i % 2][1])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        line["overlap"] = {"streams": 2, "value": round(Q * args.steps / dt, 1), "unit": "queries/s",
                           "ms_per_step": round(dt / args.steps * 1e3, 4),
                           "note": "consecutive batches on 2 HIP streams/workspaces; same work, same results"}
    # ---- quality of the answers (not part of the metric): exact-rank recall of a 512-query sample, by GPU brute force
    if world == 1 and rank == 0:
        qs = min(512, Q)
        g_ids, _, _ = ix.query(batches[0][:qs].contiguous())
        rk = A.recall_ranks(points, batches[0][:qs].contiguous(), g_ids)
        line["config"]["recall_sample"] = {kk: round(v, 4) for kk, v in A.recall_summary(rk, k).items()}
        line["config"]["recall_sample"]["queries"] = qs
    # ---- CPU baseline + full-size parity sample (rank 0, single GPU only)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args, ix, points, batches[0], libc)
    if rank == 0:
        print(json.dumps(line), flush=True)
    ix.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, ix, points, y_dev, libc):
    """CPU column on the GPU box's host cores, single thread, on a bounded sample of the same batch and the same
    (GPU-built) index, with the results compared against the GPU's.  kind = "reference": the reference's own query_cpu
    (oracle/_ref, compiled from /root/reference in the authoring container; it materialises Q*L1*d values, so the
    sample is small); otherwise kind = "port": the oracle (oracle/ann_oracle.c).  The oracle's rate is reported too."""
    import numpy as np
    import torch

    from oracle import oracle_py as O
    save = ix.export()
    arrays = save.to_dict()
    save.free()
    host_pts = points.cpu().numpy()
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
