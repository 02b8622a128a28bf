"""GPU parity: the HIP path (through the C-ABI) against the golden vectors recorded from the compiled
reference and against the oracle on fresh seeded inputs.  Bit-exact ids AND bit-exact distances: the
kernels use the reference's operation order, so the 1e-5 relative tolerance north_star allows for float
distances is never needed (it is asserted as the outer bound anyway)."""
import os

import numpy as np
import pytest

import approximatenn_amd as A
from oracle import oracle_py as O
from tests.util import assert_save_equal, bits_equal, golden_cases, load_golden

pytestmark = pytest.mark.gpu


def _check(ids, dists, g_ids, g_dists, what):
    assert np.array_equal(ids, g_ids), "%s: ids differ in %d places" % (what, int(np.sum(ids != g_ids)))
    fin = np.isfinite(g_dists)
    np.testing.assert_allclose(dists[fin], g_dists[fin], rtol=1e-5, err_msg=what)
    assert bits_equal(dists, g_dists), "%s: distances not bit-identical" % what


@pytest.fixture(params=["select", "select-unfused", "select-classic", "exact"])
def path_mode(request):
    """'select' = production path (selection + exact fallback; small batches run stage 2 in the tail of the stage-1
    workgroup); 'select-unfused' = stage 2 as its own fused kernel (what large batches take); 'select-classic' = stage 2
    as separate rows / network / widen kernels (what sharded indexes and long stage-2 rows take); 'exact' = the
    reference network for every row."""
    for v in ("ANN_HIP_EXACT", "ANN_HIP_FUSE", "ANN_HIP_TAIL"):
        os.environ.pop(v, None)
    if request.param == "exact":
        os.environ["ANN_HIP_EXACT"] = "1"
    elif request.param == "select-unfused":
        os.environ["ANN_HIP_FUSE"] = "0"
    elif request.param == "select-classic":
        os.environ["ANN_HIP_FUSE"], os.environ["ANN_HIP_TAIL"] = "0", "0"
    A._lib.reload_env()
    yield request.param
    for v in ("ANN_HIP_EXACT", "ANN_HIP_FUSE", "ANN_HIP_TAIL"):
        os.environ.pop(v, None)
    A._lib.reload_env()


@pytest.mark.parametrize("name", golden_cases())
def test_query_matches_golden(name, path_mode):
    g = load_golden(name)
    save = A.Save.from_dict(g["prec"], g["save"])
    pts = np.ascontiguousarray(g["points"])
    ids, dists = A.query(save, pts, g["y"])
    _check(ids, dists, g["query_ids"], g["query_dists"], name + " query")
    qa = len(g["alias_ids"])
    ids, dists = A.query(save, pts, pts[:qa])  # same buffer => self excluded (Q3)
    _check(ids, dists, g["alias_ids"], g["alias_dists"], name + " alias")
    ids, dists = A.query(save, pts, pts[:qa].copy())
    _check(ids, dists, g["copy_ids"], g["copy_dists"], name + " copy")
    A._lib.load(g["prec"]).annhip_cache_clear()


@pytest.mark.parametrize("name", golden_cases())
def test_precomp_matches_golden(name, path_mode):
    g = load_golden(name)
    c = g["cfg"]
    pts = np.ascontiguousarray(g["points"])
    # same libc stream position as the generator had after drawing the points
    O.srandom(c["seed"])
    orc = O.CpuBackend(g["prec"], "oracle")
    orc.rand_norm_reset()
    orc.gen_rand(c["n"] * c["d"] + (c["n"] * c["d"]) % 2)
    ids, dists, save = A.precomp(pts, c["k"], c["tries"], c["rb"], c["rlb"], c["ra"], c["rla"])
    try:
        _check(ids, dists, g["precomp_ids"], g["precomp_dists"], name + " precomp")
        assert_save_equal(save.to_dict(), g["save"])
        # the freshly built (resident) index answers queries like the reference's
        q_ids, q_d = A.query(save, pts, g["y"])
        _check(q_ids, q_d, g["query_ids"], g["query_dists"], name + " query after precomp")
    finally:
        A._lib.load(g["prec"]).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("case", [(3000, 64, 10, 10, 500), (2500, 128, 8, 6, 300), (1800, 80, 10, 10, 120),
                                  (4000, 16, 5, 8, 700), (1200, 256, 10, 4, 64), (900, 40, 20, 3, 50),
                                  # k*tries just above a power of two: precomp skips the distance pass of the tries whose
                                  # merged columns lie beyond the sorted prefix (Q1) -- 22 of 30, 13 of 14, all 9 scored
                                  (2000, 32, 3, 30, 100), (1200, 32, 5, 14, 50), (1500, 64, 8, 9, 64)])
def test_against_oracle_fresh_inputs(prec, case):
    """Seeded inputs never seen by the fixtures: GPU precomp+query vs the oracle, all fields bit-exact."""
    n, d, k, T, Q = case
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(1000 + n + d)
    orc.rand_norm_reset()
    pts = orc.gen_rand(n * d).reshape(n, d)
    y = orc.gen_rand(Q * d).reshape(Q, d)
    O.srandom(31)
    o_ids, o_d, o_save = orc.precomp(pts, k, T)
    O.srandom(31)
    ids, dists, save = A.precomp(pts, k, T)
    try:
        _check(ids, dists, o_ids, o_d, "precomp")
        assert_save_equal(save.to_dict(), o_save)
        oq = orc.query(o_save, pts, y)
        gq = A.query(save, pts, y)
        _check(gq[0], gq[1], oq[0], oq[1], "query")
        oa = orc.query(o_save, pts, min(Q, n), alias=True)
        ga = A.query(save, pts, pts[: min(Q, n)])
        _check(ga[0], ga[1], oa[0], oa[1], "alias query")
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


def test_resident_index_device_tensors():
    import torch
    g = load_golden("pow2_d128_f32")
    save = A.Save.from_dict("f32", g["save"])
    pts = torch.from_numpy(g["points"]).cuda()
    ix = A.Index.from_save(save, pts)
    y = torch.from_numpy(g["y"]).cuda()
    ix.profile(True)  # gathered-row statistics are only collected while profiling
    for mode in (0, 1):
        ids, dists, nex = ix.query(y, mode=mode)
        torch.cuda.synchronize()
        _check(ids.cpu().numpy().astype(np.uint64), dists.cpu().numpy(), g["query_ids"], g["query_dists"], "resident")
    st = ix.stats()
    assert st["queries"] == 2 * len(g["y"]) and st["s1_rows"] > 0
    ix.close()


def test_precomp_is_first_gpu_call_in_a_fresh_process():
    """HIP runtime start-up consumes libc random(); the index must not depend on whether the runtime was
    already up when precomp() drew its rotations (regression: first precomp after srandom() differed)."""
    import subprocess
    import sys
    code = ("import __graft_entry__ as g; g.smoke()")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "smoke ok" in out.stdout


def test_overlapped_batches_on_two_streams_match():
    """annhip_query_on: independent batches on their own workspace + stream give the serial answers."""
    import torch
    g = load_golden("pow2_d128_f32")
    save = A.Save.from_dict("f32", g["save"])
    ix = A.Index.from_save(save, torch.from_numpy(g["points"]).cuda())
    y = torch.from_numpy(g["y"]).cuda()
    ys = [y, y.flip(0).contiguous(), y[:17].contiguous(), y]
    want = [ix.query(t)[:2] for t in ys]
    torch.cuda.synchronize()
    lanes = [(ix.workspace(), torch.cuda.Stream()) for _ in range(2)]
    got = []
    for rep in range(3):
        got = [ix.query(t, ws=lanes[i % 2][0], stream=lanes[i % 2][1])[:2] for i, t in enumerate(ys)]
    torch.cuda.synchronize()
    for (wi, wd), (gi, gd) in zip(want, got):
        assert torch.equal(wi, gi) and torch.equal(wd.view(torch.int32), gd.view(torch.int32))
    _check(got[0][0].cpu().numpy().astype(np.uint64), got[0][1].cpu().numpy(), g["query_ids"], g["query_dists"], "overlapped")
    ix.close()


def test_host_stream_pipeline_matches_query():
    """annhip_stream_*: host batches through the pinned, multi-lane pipeline give query()'s answers, in order."""
    import torch
    g = load_golden("pow2_d64_f32")
    save = A.Save.from_dict("f32", g["save"])
    pts = np.ascontiguousarray(g["points"])
    ix = A.Index.from_save(save, pts)
    rng = np.random.default_rng(5)
    batches = [g["y"]] + [rng.standard_normal((n, pts.shape[1])).astype(np.float32) for n in (64, 1, 33, 64, 17, 64)]
    want = [A.query(save, pts, b) for b in batches]
    hs = ix.host_stream(max_ycnt=64, lanes=3)
    got = list(hs.map(batches))
    hs.close()
    for (wi, wd), (gi, gd) in zip(want, got):
        assert np.array_equal(wi, gi) and bits_equal(wd, gd)
    _check(got[0][0], got[0][1], g["query_ids"], g["query_dists"], "host stream")
    ix.close()
    A._lib.load("f32").annhip_cache_clear()


def test_randomised_shapes_against_oracle():
    """A bounded run of tools/fuzz_parity.py: random shapes over every row layout (power of two, static 3/5-lane,
    run-time lane groups, LDS tree), k, tries, rotations, duplicated points -- precomp, query and aliased query bit for
    bit against the oracle, both precisions.  (The unbounded sweep, 480 shapes x 2 precisions + 80 sharded, is clean.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--cases", "24", "--seed", "5"],
                         cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
