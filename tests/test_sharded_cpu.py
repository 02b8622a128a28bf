"""The point-sharded multi-rank orchestration (approximatenn_amd/sharded.py) under gloo on CPU, world_size 2 and 3,
with the oracle-backed CpuShardEngine standing in for the per-shard HIP kernels.  Result on every rank must be
bit-identical to the reference's single-device answer (golden vectors)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from approximatenn_amd.sharded import ShardedQuery
from oracle import oracle_py as O
from tests.cpu_engine import CpuShardEngine
from tests.util import bits_equal, load_golden


def _bounds(n, world, rank):
    return (n * rank) // world, (n * (rank + 1)) // world


def _tie_case():
    """Every point exists twice (two ids, identical coordinates): ties between different ids everywhere,
    so nearly every query is rejected by the selection proof and takes the exact, min-all-reduced path."""
    orc = O.CpuBackend("f32", "oracle")
    O.srandom(99)
    orc.rand_norm_reset()
    half = orc.gen_rand(150 * 16).reshape(150, 16)
    pts = np.ascontiguousarray(np.concatenate([half, half]))
    y = orc.gen_rand(12 * 16).reshape(12, 16)
    O.srandom(5)
    _, _, save = orc.precomp(pts, 4, 3)
    ids, dd = orc.query(save, pts, y)
    return pts, y, save, ids, dd


def _worker(rank, world, port, case, mode="plain"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if mode != "plain":
            return _worker_modes(rank, world, case, mode)
        if case == "ties":
            pts, y, save, want_ids, want_d = _tie_case()
            prec = "f32"
        else:
            g = load_golden(case)
            pts, y, save, want_ids, want_d, prec = g["points"], g["y"], g["save"], g["query_ids"], g["query_dists"], g["prec"]
        lo, hi = _bounds(len(pts), world, rank)
        eng = CpuShardEngine(save, pts, lo, hi, prec)
        # ties: nearly every query is flagged; fcap = 3 leaves most of them to the host-driven repair after the step
        sq = ShardedQuery(eng, dist, fcap=3 if case == "ties" else 32)
        ids, dd = sq.query(torch.from_numpy(np.ascontiguousarray(y)))
        assert np.array_equal(ids.numpy().astype(np.uint64), want_ids), "rank %d ids" % rank
        assert bits_equal(dd.numpy(), want_d), "rank %d dists" % rank
        if case == "ties":
            assert sq.last_exact > 3
            sq2 = ShardedQuery(eng, dist)       # default fcap: all of them on the device-driven path
            ids, dd = sq2.query(torch.from_numpy(np.ascontiguousarray(y)))
            assert np.array_equal(ids.numpy().astype(np.uint64), want_ids) and bits_equal(dd.numpy(), want_d)
            assert sq2.last_exact == sq.last_exact
    finally:
        dist.destroy_process_group()


def _worker_modes(rank, world, case, mode):
    g = load_golden(case)
    pts, y, save, want_ids, want_d, prec = g["points"], g["y"], g["save"], g["query_ids"], g["query_dists"], g["prec"]
    lo, hi = _bounds(len(pts), world, rank)
    eng = CpuShardEngine(save, pts, lo, hi, prec)
    yt = torch.from_numpy(np.ascontiguousarray(y))
    if mode == "allgather":      # the fallback exchange: every rank receives everything and keeps its slice
        sq = ShardedQuery(eng, dist, exchange="allgather")
        assert sq.exchange == "allgather"
        ids, dd = sq.query(yt)
    elif mode == "exact":        # every query through the repair path (what k > P1 or ANN_HIP_EXACT selects)
        sq = ShardedQuery(eng, dist, exact_all=True)
        ids, dd = sq.query(yt)
        assert sq.last_exact == len(y)
    elif mode == "pipelined":    # two batches in flight; the second is a ragged prefix of the first's queries
        sq = ShardedQuery(eng, dist, lanes=2)
        assert sq.exchange == "alltoall"          # probed and agreed on at start-up
        t0 = sq.submit(yt)
        with pytest.raises(RuntimeError):
            sq.submit(yt), sq.submit(yt)          # only two lanes
        ids, dd = sq.collect(t0)
        sq.collect(t0 + 1)
    elif mode == "autotune":     # the schedule is measured and agreed on (MAX over ranks per candidate); results unchanged
        sq = ShardedQuery(eng, dist, lanes=3)
        tuned = sq.autotune(yt, batches=2, candidates=[(3, True, 0, 1), (2, False, 0, 4), (1, False, 0, 1)])
        assert len(tuned["table"]) == 3 and (tuned["depth"], tuned["split"], tuned["pieces"]) in [(3, True, 1), (2, False, 4), (1, False, 1)]
        mine = torch.tensor([tuned["depth"], int(tuned["split"]), tuned["pieces"]], dtype=torch.int64)
        lo_, hi_ = mine.clone(), mine.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        assert torch.equal(lo_, hi_), "ranks disagree on the schedule"
        res = sq.pump([yt, yt, yt, yt])
        assert len(res) == 4
        for ids, dd in res:
            assert np.array_equal(ids.numpy().astype(np.uint64), want_ids) and bits_equal(dd.numpy(), want_d)
    elif mode == "stagewise":    # stage-major issue order (split = 2): seven lanes, a batch advances one stage per submit
        sq = ShardedQuery(eng, dist, lanes=7)
        sq.configure(7, 2, 0, 1)
        res = sq.pump([yt] * 10)                       # steady state: collect() finds its batch finished
        assert len(res) == 10
        for ids, dd in res:
            assert np.array_equal(ids.numpy().astype(np.uint64), want_ids) and bits_equal(dd.numpy(), want_d)
        t0, t1 = sq.submit(yt), sq.submit(yt)          # fewer batches than the pipeline is deep: collect() flushes
        ids, dd = sq.collect(t0)
        ids1, dd1 = sq.collect(t1)
        assert np.array_equal(ids1.numpy(), ids.numpy()) and bits_equal(dd1.numpy(), dd.numpy())
    elif mode == "autotune_budget":   # no budget left after the pinned candidate: every rank stops there, together
        sq = ShardedQuery(eng, dist, lanes=3)
        tuned = sq.autotune(yt, batches=2, budget_s=0.0)
        assert len(tuned["table"]) == 1 and "stopped" in tuned
        assert (tuned["depth"], tuned["split"], tuned["reserve_cus"], tuned["pieces"]) == ShardedQuery.PINNED
        ids, dd = sq.pump([yt])[0]
    else:
        raise AssertionError(mode)
    assert np.array_equal(ids.numpy().astype(np.uint64), want_ids), "rank %d ids (%s)" % (rank, mode)
    assert bits_equal(dd.numpy(), want_d), "rank %d dists (%s)" % (rank, mode)


@pytest.mark.parametrize("world,case,mode", [(2, "tiny_appendixA_f32", "allgather"), (3, "few_candidates_f64", "allgather"),
                                             (2, "odd_everything_f32", "exact"), (3, "tiny_appendixA_f32", "pipelined"), (2, "tiny_appendixA_f32", "autotune"),
                                             (2, "tiny_appendixA_f32", "autotune_budget"), (2, "tiny_appendixA_f32", "stagewise"),
                                             (3, "odd_everything_f32", "stagewise")])
def test_sharded_query_modes(world, case, mode):
    port = 29500 + (os.getpid() + hash((case, mode))) % 2000
    mp.spawn(_worker, args=(world, port, case, mode), nprocs=world, join=True)


def test_batch_not_divisible_by_world_size():
    """Ragged owner slices (the last rank owns fewer queries, or none): Q = 7 over 3 ranks and Q = 2 over 3 ranks."""
    port = 29500 + (os.getpid() + 777) % 2000
    mp.spawn(_ragged_worker, args=(3, port), nprocs=3, join=True)


def _ragged_worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("tiny_appendixA_f32")
        orc = O.CpuBackend("f32", "oracle")
        lo, hi = _bounds(len(g["points"]), world, rank)
        eng = CpuShardEngine(g["save"], g["points"], lo, hi, "f32")
        sq = ShardedQuery(eng, dist)
        for Q in (7, 2):
            y = np.ascontiguousarray(g["y"][:Q])
            want_ids, want_d = orc.query(g["save"], g["points"], y)     # results depend on the batch (Q2): ask the oracle
            ids, dd = sq.query(torch.from_numpy(y))
            assert np.array_equal(ids.numpy().astype(np.uint64), want_ids) and bits_equal(dd.numpy(), want_d), (rank, Q)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "tiny_appendixA_f32"), (2, "odd_everything_f32"), (3, "few_candidates_f64"),
                                        (2, "k17_d100_f64"), (2, "ties")])
def test_sharded_query_matches_reference(world, case):
    port = 29500 + (os.getpid() + hash(case)) % 2000
    mp.spawn(_worker, args=(world, port, case), nprocs=world, join=True)


def test_single_rank_engine_matches_reference():
    g = load_golden("one_try_one_query_f32")
    eng = CpuShardEngine(g["save"], g["points"], 0, g["cfg"]["n"], "f32")
    ids, dd = ShardedQuery(eng, None).query(torch.from_numpy(g["y"]))
    assert np.array_equal(ids.numpy().astype(np.uint64), g["query_ids"]) and bits_equal(dd.numpy(), g["query_dists"])


def _same_worker(rank, world, port, differ):
    from approximatenn_amd.sharded import same_everywhere
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vals = [0xFEDCBA9876543210, 7, (1 << 63) + 5]        # 64-bit patterns incl. the sign bit
        assert same_everywhere(dist, vals, "test values") == vals
        if differ:
            vals[2] += rank                                   # rank 1 holds another checksum: EVERY rank must stop
            with pytest.raises(SystemExit) as ei:
                same_everywhere(dist, vals, "test values")
            assert ei.value.code == 3
    finally:
        dist.destroy_process_group()


def test_ranks_with_different_inputs_stop_together():
    """bench.py's start-up check (same index, same batches on every rank): MIN/MAX all-reduce of checksums; a
    difference ends every rank with exit status 3 instead of silently different answers or a hang."""
    port = 29500 + (os.getpid() + 4242) % 2000
    mp.spawn(_same_worker, args=(2, port, True), nprocs=2, join=True)


def test_random_stream_hand_over_between_processes():
    """bench.py generates the points on rank 0 only; the libc random() state after them is shipped to the other ranks
    (approximatenn_amd._lib.random_state_snapshot / _restore) so that every rank's precomp draws the same rotations."""
    import subprocess
    import sys
    from approximatenn_amd import _lib
    O.srandom(2024)
    for _ in range(1001):
        O.libc_random()
    snap = _lib.random_state_snapshot()
    want = [O.libc_random() for _ in range(8)]
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from approximatenn_amd import _lib\nfrom oracle import oracle_py as O\n"
            "O.srandom(1)\n_lib.random_state_restore(bytes.fromhex(%r))\n"
            "print([O.libc_random() for _ in range(8)])" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), snap.hex()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == str(want)
    # parking is nestable and leaves the caller's stream where it was
    _lib.random_state_restore(snap)
    with _lib.park_random():
        O.libc_random()
        with _lib.park_random():
            O.libc_random()
    assert [O.libc_random() for _ in range(8)] == want
