"""GPU: the row-sharded kernels (owned-range filtering, +inf for foreign slots, per-shard candidates) and the
sharded orchestration, with G shard indexes living on the ONE GPU of the test box.  Each shard runs in its own
thread; ThreadDist is an in-process stand-in for the subset of torch.distributed that sharded.py uses."""
import threading

import numpy as np
import pytest
import torch

import approximatenn_amd as A
from approximatenn_amd.sharded import ShardedQuery
from oracle import oracle_py as O
from tests.util import bits_equal, load_golden

pytestmark = pytest.mark.gpu


class ThreadDist:
    """The subset of torch.distributed that sharded.py uses, for G threads of one process sharing one GPU."""

    class ReduceOp:
        MIN = "min"

    def __init__(self, world):
        self.world, self.bar, self.slots, self.tl = world, threading.Barrier(world), [None] * world, threading.local()

    def is_initialized(self):
        return True

    def get_world_size(self, group=None):
        return self.world

    def get_backend(self, group=None):
        return "threads"

    def get_rank(self, group=None):
        return self.tl.rank

    def _publish(self, t):
        torch.cuda.current_stream().synchronize()   # producers ran on this thread's lane stream
        self.slots[self.tl.rank] = t
        self.bar.wait()

    def _done(self):
        torch.cuda.current_stream().synchronize()
        self.bar.wait()

    def all_gather_into_tensor(self, out, t, group=None):
        self._publish(t)
        out.copy_(torch.cat([s_.reshape((-1,) + tuple(out.shape[1:])) for s_ in self.slots]).reshape(out.shape))
        self._done()

    def all_to_all_single(self, out, t, group=None):
        self._publish(t)
        n = t.shape[0] // self.world
        r = self.tl.rank
        out.copy_(torch.cat([s_[r * n:(r + 1) * n] for s_ in self.slots]))
        self._done()

    def all_reduce(self, t, op=None, group=None):
        self._publish(t.clone())
        res = self.slots[0]
        for s_ in self.slots[1:]:
            res = torch.minimum(res, s_)
        self._done()
        t.copy_(res)
        self._done()


LAST_TIE_QUERIES = {}


def _run_sharded(prec, save_arrays, pts, y, world, alias=False, fast=True, pipelined=False, schedule=None):
    save = A.Save.from_dict(prec, save_arrays)
    td = ThreadDist(world)
    results, errors = [None] * world, []
    yt = torch.from_numpy(np.ascontiguousarray(y)).cuda()

    def work(rank):
        try:
            td.tl.rank = rank
            lo, hi = (len(pts) * rank) // world, (len(pts) * (rank + 1)) // world
            ix = A.Index.from_save(save, torch.from_numpy(np.ascontiguousarray(pts[lo:hi])).cuda(), lo, hi)
            sq = ShardedQuery(ix, td, exchange="alltoall" if fast else "allgather", lanes=3 if schedule else 2)
            if schedule:     # (lanes in use, two-half issue, CUs kept free of the gathers, launches per gather)
                sq.configure(*schedule)
                res = sq.pump([yt] * 5, alias=alias)
                ids, dd = res[0]
                for ids1, dd1 in res[1:]:
                    assert torch.equal(ids, ids1) and torch.equal(dd.view(torch.uint8), dd1.view(torch.uint8))
            elif pipelined:  # two batches in flight on two HIP streams
                t0, t1 = sq.submit(yt, alias=alias), sq.submit(yt, alias=alias)
                ids, dd = sq.collect(t0)
                ids1, dd1 = sq.collect(t1)
                assert torch.equal(ids, ids1) and torch.equal(dd.view(torch.uint8), dd1.view(torch.uint8))
            else:
                ids, dd = sq.query(yt, alias=alias)
            torch.cuda.synchronize()
            results[rank] = (ids.cpu().numpy().astype(np.uint64), dd.cpu().numpy(), sq.last_exact)
            LAST_TIE_QUERIES[rank] = ix.stats()["tie_queries"]  # flagged rows this rank answered without the network
            ix.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            td.bar.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    return results


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("name,world", [("pow2_d128_f32", 2), ("pow2_d128_f64", 3), ("defaults_d80_f32", 2),
                                        ("pow2_d32_f32", 4), ("k17_d100_f64", 2), ("few_candidates_f32", 2),
                                        ("pow2_d64_f32", 8)])
def test_sharded_on_one_gpu_matches_golden(name, world, fast):
    # owner protocol with the all-to-all exchange (fast) and with the all-gather fallback; Q divisible by the world size and not
    g = load_golden(name)
    for ids, dd, _ in _run_sharded(g["prec"], g["save"], g["points"], g["y"], world, fast=fast):
        assert np.array_equal(ids, g["query_ids"]) and bits_equal(dd, g["query_dists"])


def _dup_dataset(prec, n_half, d, k, T, Q):
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(321)
    orc.rand_norm_reset()
    half = orc.gen_rand(n_half * d).reshape(n_half, d)
    pts = np.ascontiguousarray(np.concatenate([half, half]))   # every point twice => ties between different ids
    y = orc.gen_rand(Q * d).reshape(Q, d)
    O.srandom(17)
    o_ids, o_d, o_save = orc.precomp(pts, k, T)
    return orc, pts, y, o_ids, o_d, o_save


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_ties_take_the_exact_path_and_still_match(prec):
    orc, pts, y, o_ids, o_d, o_save = _dup_dataset(prec, 700, 32, 6, 4, 90)
    O.srandom(17)
    ids, dists, save = A.precomp(pts, 6, 4)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dists, o_d)
        want = orc.query(o_save, pts, y)
        got = A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
        ix = A.Index.from_save(save, pts)
        r_ids, r_d, _ = ix.query(torch.from_numpy(y).cuda())
        nex = ix.stats()["exact_queries"]
        assert nex > len(y) // 2, "duplicated points must trip the tie test for most queries (got %d)" % nex
        assert np.array_equal(r_ids.cpu().numpy().astype(np.uint64), want[0])
        ix.close()
        for s_ids, s_d, s_ex in _run_sharded(prec, o_save, pts, y, 2) + _run_sharded(prec, o_save, pts, y, 3, fast=False):
            assert np.array_equal(s_ids, want[0]) and bits_equal(s_d, want[1]) and s_ex > 0
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


def test_sharded_alias_query():
    g = load_golden("pow2_d64_f32")
    qa = len(g["alias_ids"])
    for ids, dd, _ in _run_sharded("f32", g["save"], g["points"], g["points"][:qa], 2, alias=True):
        assert np.array_equal(ids, g["alias_ids"]) and bits_equal(dd, g["alias_dists"])


@pytest.mark.parametrize("name,world", [("pow2_d128_f32", 4), ("k17_d100_f64", 3)])
def test_two_batches_in_flight(name, world):
    g = load_golden(name)
    for ids, dd, _ in _run_sharded(g["prec"], g["save"], g["points"], g["y"], world, pipelined=True):
        assert np.array_equal(ids, g["query_ids"]) and bits_equal(dd, g["query_dists"])


@pytest.mark.parametrize("schedule", [(3, True, 0, 4), (2, False, 8, 3), (1, False, 0, 7), (3, False, 0, 1)])
def test_schedules_do_not_change_results(schedule):
    # what ShardedQuery.autotune() chooses between: lanes, issue order, a CU-masked gather stream, the gather in pieces
    g = load_golden("pow2_d128_f32")
    for ids, dd, _ in _run_sharded(g["prec"], g["save"], g["points"], g["y"], 2, schedule=schedule):
        assert np.array_equal(ids, g["query_ids"]) and bits_equal(dd, g["query_dists"])


def test_sharded_exact_everywhere(monkeypatch):
    """ANN_HIP_EXACT=1: every query is flagged and goes through the repair pass (rows, MIN all-reduce, network)."""
    monkeypatch.setenv("ANN_HIP_EXACT", "1")
    A._lib.reload_env()
    g = load_golden("pow2_d32_f32")
    for ids, dd, nex in _run_sharded("f32", g["save"], g["points"], g["y"], 2):
        assert np.array_equal(ids, g["query_ids"]) and bits_equal(dd, g["query_dists"]) and nex == len(g["y"])


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_small_shards_with_a_heavy_bucket(prec):
    """Shards of 1/4 and 1/8 of the rows scan through the inline 32-byte bucket records; a cluster of near-identical
    points makes buckets with far more than the 7 owned ids a record holds, so the table-row continuation runs too."""
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(2025)
    orc.rand_norm_reset()
    n, d, k, T, Q = 4000, 32, 5, 3, 96
    pts = orc.gen_rand(n * d).reshape(n, d)
    rng = np.random.default_rng(3)
    heavy = rng.choice(n, size=120, replace=False)
    pts[heavy] = pts[heavy[0]] + (1e-3 * orc.gen_rand(120 * d).reshape(120, d)).astype(pts.dtype)
    y = orc.gen_rand(Q * d).reshape(Q, d)
    y[:32] = pts[heavy[0]] + (1e-2 * orc.gen_rand(32 * d).reshape(32, d)).astype(pts.dtype)   # these probe the heavy bucket
    pts, y = np.ascontiguousarray(pts), np.ascontiguousarray(y)
    O.srandom(8)
    _, _, o_save = orc.precomp(pts, k, T)
    assert max(int(v) for v in o_save["par_maxes"]) > 60 and max(int(v) for v in o_save["par_maxes"]) <= 255
    want = orc.query(o_save, pts, y)
    for world in (4, 8):
        for ids, dd, _ in _run_sharded(prec, o_save, pts, y, world):
            assert np.array_equal(ids, want[0]) and bits_equal(dd, want[1]), world


@pytest.mark.parametrize("name,world", [("pow2_d64_f32", 2), ("pow2_d128_f64", 3), ("pow2_d32_f32", 4), ("defaults_d80_f32", 2),
                                        ("k17_d100_f64", 2)])
def test_sharded_precomp_matches_golden(name, world):
    """precomp with its distance passes dealt to the ranks by bucket (power-of-two d) -- or run redundantly where the
    bucket kernel does not apply (d = 80, 100) -- gives the reference's index on every rank, field for field."""
    from approximatenn_amd.sharded import precomp_sharded
    from tests.util import assert_save_equal
    g = load_golden(name)
    c, prec = g["cfg"], g["prec"]
    pts = torch.from_numpy(np.ascontiguousarray(g["points"])).cuda()
    td = ThreadDist(world)
    lock = threading.Lock()
    results, errors = [None] * world, []

    def seed():    # every "rank" starts from the libc stream position the golden generator had after drawing the points
        O.srandom(c["seed"])
        orc = O.CpuBackend(prec, "oracle")
        orc.rand_norm_reset()
        orc.gen_rand(c["n"] * c["d"] + (c["n"] * c["d"]) % 2)

    def work(rank):
        try:
            td.tl.rank = rank
            lock.acquire()     # ranks share one process, hence one random() stream: seed + draw one rank at a time
            ix = _precomp_with_lock(precomp_sharded, lock, pts, c, dict(dist=td, want_dists=True, _begin_hook=seed))
            save = ix.export()
            results[rank] = (save.to_dict(), ix.graph_dists.cpu().numpy())
            y = torch.from_numpy(np.ascontiguousarray(g["y"])).cuda()
            ids, dd, _ = ix.query(y)
            torch.cuda.synchronize()
            assert np.array_equal(ids.cpu().numpy().astype(np.uint64), g["query_ids"]) and bits_equal(dd.cpu().numpy(), g["query_dists"])
            save.free()
            ix.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            td.bar.abort()
            if lock.locked():
                try:
                    lock.release()
                except RuntimeError:
                    pass
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for sd, gdists in results:
        assert_save_equal(sd, g["save"])
        assert bits_equal(gdists, g["precomp_dists"])


def _precomp_with_lock(precomp_sharded, lock, pts, c, kw):
    """Run precomp_sharded with the caller HOLDING `lock`; the lock is released as soon as annhip_precomp_begin (the only
    consumer of the process-wide random() stream) has returned, i.e. at the first collective."""
    td = kw["dist"]
    released = []
    orig = td.all_gather_into_tensor

    def first_collective(out, t, group=None):
        if not released:
            released.append(1)
            lock.release()
        return orig(out, t, group=group)
    proxy = _Proxy(td, first_collective)
    kw = dict(kw, dist=proxy)
    return precomp_sharded(pts, c["k"], c["tries"], c["rb"], c["rlb"], c["ra"], c["rla"], **kw)


class _Proxy:
    def __init__(self, td, ag):
        self._td, self.all_gather_into_tensor = td, ag

    def __getattr__(self, name):
        return getattr(self._td, name)


@pytest.mark.parametrize("name,world", [("pow2_d128_f32", 2), ("pow2_d64_f32", 3), ("defaults_d80_f64", 2), ("pow2_d32_f32", 8),
                                        ("one_try_one_query_f32", 2)])
def test_replica_hosts_match_golden(name, world):
    """Query-sharded hosts (approximatenn_amd.sharded.ReplicaQuery, annhip_query_slice): every rank holds all rows, hashes its
    slice, ONE all-gather of the codes, each rank answers its slice with the single-GPU path; results all-gathered.  Every
    rank must return the reference's answer for the whole batch (results depend on the other queries' codes, Q2)."""
    from approximatenn_amd.sharded import ReplicaQuery
    g = load_golden(name)
    save = A.Save.from_dict(g["prec"], g["save"])
    pts = torch.from_numpy(np.ascontiguousarray(g["points"])).cuda()
    yt = torch.from_numpy(np.ascontiguousarray(g["y"])).cuda()
    td = ThreadDist(world)
    results, errors = [None] * world, []

    def work(rank):
        try:
            td.tl.rank = rank
            ix = A.Index.from_save(save, pts)
            rq = ReplicaQuery(ix, td, lanes=2)
            res = rq.pump([yt, yt, yt])
            own = ReplicaQuery(ix, td, lanes=1, gather=False).query(yt)
            torch.cuda.synchronize()
            results[rank] = ([(i.cpu().numpy().astype(np.uint64), d_.cpu().numpy()) for i, d_ in res],
                             (own[0].cpu().numpy().astype(np.uint64), own[1].cpu().numpy()))
            ix.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            td.bar.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    Q = len(g["y"])
    qs = (Q + world - 1) // world
    for rank, (full, own) in enumerate(results):
        for ids, dd in full:
            assert np.array_equal(ids, g["query_ids"]) and bits_equal(dd, g["query_dists"])
        lo, hi = min(Q, rank * qs), min(Q, rank * qs + qs)
        assert np.array_equal(own[0], g["query_ids"][lo:hi]) and bits_equal(own[1], g["query_dists"][lo:hi])
