"""GPU: bench.py prints exactly ONE JSON line with the contract's fields (small configuration, seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", "200000", "--queries", "2000", "--steps", "4",
                          "--warmup", "1", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "queries/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2000 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0
    assert c["parity_on_sample"] == {"ids_bit_exact": True, "dists_bit_exact": True}
    # labels come from the actual sizes, not from the default workload
    assert "N=200k d=128 k=10 Q=2k float" in d["metric"] and "N=200000 d=128 k=10 tries=10 Q=2000/step float" in d["config"]["workload"]
    assert "cfg" not in d["config"]["workload"] and r["traffic"] is None      # PMC traffic only for the profiled default
    assert "random()" in d["config"]["data_generator"]
    h = d["host_api"]   # the reference's host-pointer ABI, beside (never as) value
    assert h["query_gpu"]["value"] > 0 and h["query_gpu"]["same_results_as_resident_path"] is True
    assert h["stream_3_lanes"]["value"] > 0


def test_bench_labels_follow_dtype_and_shape():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", "60000", "--dim", "64", "--knn", "12",
                          "--queries", "500", "--steps", "2", "--warmup", "1", "--dtype", "f64", "--data", "randn",
                          "--no-cpu-baseline", "--no-host-api"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["dtype"] == "f64" and "N=60k d=64 k=12 Q=500 double" in d["metric"] and "double" in d["config"]["workload"]
    assert "torch.randn" in d["config"]["data_generator"] and "cpu_baseline" not in d and "host_api" not in d


def test_bench_multi_rank_branch_two_ranks_sharing_the_gpu():
    """The --gpus N branch (row sharding, staged calls, collectives, max-over-ranks timing) with 2 ranks on this box's
    single GPU and gloo collectives (ANN_BENCH_SHARED_GPU=1): the path the driver runs on a multi-GPU node over RCCL."""
    env = dict(os.environ, ANN_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29633", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points", "200000", "--queries", "1000",
           "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["points_sharding"] == "rows/2" and "Q=2000/step" in d["config"]["workload"]
    assert d["config"]["exchange"] == "alltoall" and d["config"]["queries_per_step_total"] == 2000
    assert "cpu_baseline" not in d          # rank 0 at N=1 only
    s = d["strong"]                         # the same job with the batch fixed at --queries in total
    assert s["queries_per_step_total"] == 1000 and s["value"] > 0
    r = d["replicas"]                       # the query-sharded alternative (every GPU holds all rows), beside the mandated line
    assert r["queries_per_step_total"] == 2000 and r["value"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun around it -- how the driver starts every bench line: the script
    starts its ranks as a fresh child process before anything touches the GPU, relays the one JSON line and the exit status."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ANN_BENCH_SHARED_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points", "200000", "--queries", "1000",
           "--steps", "2", "--warmup", "1", "--no-strong-extra", "--tune-seconds", "5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    r = d["config"]["rccl"]
    assert r["world_size"] == 2 and r["backend"] == "gloo" and len(r["index_checksum"]) == 16
    sch = d["config"]["schedule"]
    assert sch["table"][0]["depth"] == 3 and sch["table"][0]["split"] is True      # the pinned schedule is measured first
    # failing ranks give a non-zero exit status and no JSON line (here: an empty point set, which the library refuses)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points", "0"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT, env=env)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_multi_rank_fallback_exchange():
    """ANN_SHARD_EXCHANGE=allgather: the exchange every rank falls back to, together, when all_to_all_single is not
    available on the backend (decided collectively at start-up, sharded.py)."""
    env = dict(os.environ, ANN_BENCH_SHARED_GPU="1", ANN_SHARD_EXCHANGE="allgather")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29634", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points", "100000", "--queries", "500",
           "--steps", "2", "--warmup", "1", "--data", "randn", "--no-strong-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["exchange"] == "allgather" and "strong" not in d


def test_bench_single_rank_over_rccl():
    """ANN_SHARD_FORCE_DIST=1 with ONE rank: the sharded host with its real RCCL collectives (all_to_all_single on packed
    keys, all_gather_into_tensor on u32 / bytes, MIN all_reduce, the start-up probe, the high-priority communicator) on
    this single-GPU box -- the calls the driver's multi-GPU run makes, short of a second GPU."""
    env = dict(os.environ, ANN_SHARD_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29635", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--points", "200000", "--queries", "2000",
           "--steps", "3", "--warmup", "1", "--data", "randn"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["exchange"] == "alltoall" and d["value"] > 0 and d["strong"]["value"] > 0
    r = d["config"]["rccl"]         # proof of what the collectives ran on
    assert r["backend"] == "nccl" and r["world_size"] == 1 and r["distinct_devices"] == 1 and r["rccl_version"]
