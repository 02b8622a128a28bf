"""GPU: the point-sharded step driven from ONE process behind the drop-in surface (csrc/ann_multi_host.h).

query_gpu()/precomp_gpu() -- the symbols the reference's drivers bind (/root/reference/algg.h:5-11) -- with the point
rows sharded over G shards: ANN_HIP_VIRTUAL_SHARDS=G / annhip_set_devices(0, G) puts the shards on this box's one GPU with
loop-back exchanges; ANN_HIP_DEVICES=1 + ANN_HIP_FORCE_RCCL=1 runs the same host code over real RCCL calls
(ncclCommInitAll, grouped all-gather / all-to-all / MIN all-reduce) with a communicator of one.  Everything must be
bit-identical to the goldens recorded from the compiled reference, and the reference's own compare_results must print 0."""
import os
import re
import subprocess

import numpy as np
import pytest

import approximatenn_amd as A
from oracle import oracle_py as O
from tests.util import assert_save_equal, bits_equal, golden_cases, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = os.path.join(ROOT, "tests", "harness")
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture
def shards(request):
    """annhip_set_devices(0, G) on both libraries for the duration of a test."""
    def set_(G):
        for prec in ("f32", "f64"):
            lib = A._lib.load(prec)
            lib.annhip_cache_clear()
            lib.annhip_set_devices(0, G)
    yield set_
    for prec in ("f32", "f64"):
        lib = A._lib.load(prec)
        lib.annhip_cache_clear()
        lib.annhip_set_devices(0, 0)
    for v in ("ANN_HIP_EXACT",):
        os.environ.pop(v, None)
    A._lib.reload_env()


def _same(ids, dists, g_ids, g_dists, what):
    assert np.array_equal(ids, g_ids), "%s: ids differ in %d places" % (what, int(np.sum(ids != g_ids)))
    assert bits_equal(dists, g_dists), "%s: distances not bit-identical" % what


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("name", golden_cases())
def test_sharded_query_gpu_matches_golden(name, G, shards):
    g = load_golden(name)
    shards(G)
    save = A.Save.from_dict(g["prec"], g["save"])
    pts = np.ascontiguousarray(g["points"])
    ids, dists = A.query(save, pts, g["y"])
    assert A._lib.load(g["prec"]).annhip_host_shards(save.c) == G
    _same(ids, dists, g["query_ids"], g["query_dists"], "%s query over %d shards" % (name, G))
    qa = len(g["alias_ids"])
    ids, dists = A.query(save, pts, pts[:qa])          # same buffer => self excluded (Q3)
    _same(ids, dists, g["alias_ids"], g["alias_dists"], name + " alias")
    ids, dists = A.query(save, pts, pts[:qa].copy())
    _same(ids, dists, g["copy_ids"], g["copy_dists"], name + " copy")


@pytest.mark.parametrize("G", [2, 4])
@pytest.mark.parametrize("name", ["pow2_d64_f32", "pow2_d128_f64", "defaults_d80_f32", "k17_d100_f64", "tiny_appendixA_f32",
                                  "odd_everything_f64"])
def test_sharded_precomp_gpu_matches_golden(name, G, shards):
    """precomp_gpu with its distance passes dealt to the shards by bucket (or run redundantly where the bucket kernel does
    not apply): every save_t field, the graph and its distances as the reference's; the sharded index it leaves resident
    answers the golden queries."""
    g = load_golden(name)
    c = g["cfg"]
    shards(G)
    pts = np.ascontiguousarray(g["points"])
    O.srandom(c["seed"])
    orc = O.CpuBackend(g["prec"], "oracle")
    orc.rand_norm_reset()
    orc.gen_rand(c["n"] * c["d"] + (c["n"] * c["d"]) % 2)
    ids, dists, save = A.precomp(pts, c["k"], c["tries"], c["rb"], c["rlb"], c["ra"], c["rla"])
    after = O.libc_random()
    try:
        _same(ids, dists, g["precomp_ids"], g["precomp_dists"], name + " precomp")
        assert_save_equal(save.to_dict(), g["save"])
        assert A._lib.load(g["prec"]).annhip_host_shards(save.c) == G      # left resident, sharded
        q_ids, q_d = A.query(save, pts, g["y"])
        _same(q_ids, q_d, g["query_ids"], g["query_dists"], name + " query after precomp")
    finally:
        A._lib.load(g["prec"]).annhip_cache_clear()
        save.free()
    # the build consumed the caller's random() stream exactly once (shard 0 draws, the others reuse its transforms)
    A._lib.load(g["prec"]).annhip_set_devices(0, 0)
    O.srandom(c["seed"])
    orc.rand_norm_reset()
    orc.gen_rand(c["n"] * c["d"] + (c["n"] * c["d"]) % 2)
    _, _, save1 = A.precomp(pts, c["k"], c["tries"], c["rb"], c["rlb"], c["ra"], c["rla"])
    assert O.libc_random() == after
    save1.free()


def test_sharded_exact_everywhere_and_repair(shards):
    """ANN_HIP_EXACT=1: every query is flagged, 32 take the device-driven exact path and the rest the host-driven repair
    (MIN all-reduces of their full rows); and a duplicated-point dataset whose ties reject most selection proofs."""
    g = load_golden("pow2_d64_f32")
    os.environ["ANN_HIP_EXACT"] = "1"
    A._lib.reload_env()
    shards(3)
    save = A.Save.from_dict("f32", g["save"])
    pts = np.ascontiguousarray(g["points"])
    ids, dists = A.query(save, pts, g["y"])
    _same(ids, dists, g["query_ids"], g["query_dists"], "exact everywhere")
    os.environ.pop("ANN_HIP_EXACT")
    A._lib.reload_env()
    A._lib.load("f32").annhip_cache_clear()
    # ties: every point twice
    orc = O.CpuBackend("f32", "oracle")
    O.srandom(99)
    orc.rand_norm_reset()
    half = orc.gen_rand(1500 * 32).reshape(1500, 32)
    pts = np.ascontiguousarray(np.concatenate([half, half]))
    y = orc.gen_rand(400 * 32).reshape(400, 32)
    O.srandom(5)
    _, _, o_save = orc.precomp(pts, 6, 5)
    want = orc.query(o_save, pts, y)
    save = A.Save.from_dict("f32", o_save)
    for G in (2, 5):
        shards(G)
        got = A.query(save, pts, y)
        _same(got[0], got[1], want[0], want[1], "ties over %d shards" % G)


def test_sharded_ragged_and_tiny_batches(shards):
    """Q not divisible by the shard count, Q smaller than the shard count, Q = 1 (results depend on the batch, Q2:
    the oracle is asked per batch)."""
    g = load_golden("pow2_d32_f32")
    orc = O.CpuBackend("f32", "oracle")
    save = A.Save.from_dict("f32", g["save"])
    pts = np.ascontiguousarray(g["points"])
    shards(4)
    for Q in (1, 3, 7, 0):
        y = np.ascontiguousarray(g["y"][:Q])
        got = A.query(save, pts, y)
        if Q == 0:
            assert got[0].shape == (0, g["cfg"]["k"])
            continue
        want = orc.query(O.HostSave(g["save"], "f32"), pts, y)
        _same(got[0], got[1], want[0], want[1], "Q=%d" % Q)


def _run(cmd, env=None):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("G", [2, 8])
def test_c_harness_compare_results_sharded(prec, G):
    """tests/harness/compare_results -V G: precomp and query of a C program through include/ann.h, rows sharded."""
    out = _run([os.path.join(H, "compare_results_" + prec), "-o", "2", "-S", "41", "-V", str(G)])
    assert "Average diffs for comp: 0" in out and "PASS" in out and "0 not bit-identical" in out
    out = _run([os.path.join(H, "compare_results_" + prec), "-n", "5000", "-d", "64", "-y", "200", "-o", "2", "-S", "42", "-V", str(G)])
    assert "Average diffs for query: 0" in out and "PASS" in out and "0 not bit-identical" in out


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("G", [2, 8])
def test_reference_own_compare_results_sharded(prec, G):
    """The reference's UNMODIFIED compare_results.c + ann.c + algc.c linked against this library (oracle/Makefile dropin),
    with ONE environment variable set: 0 differences between its CPU path and the sharded GPU path."""
    exe = os.path.join(REF, "compare_results_dropin_" + prec)
    if not os.path.exists(exe):
        pytest.skip("not built (no /root/reference at build time)")
    env = {"ANN_HIP_VIRTUAL_SHARDS": str(G)}
    m = re.search(r"Average diffs for comp: (\S+)", _run([exe, "-o", "2"], env))
    assert m and float(m.group(1)) == 0.0
    m = re.search(r"Average diffs for query: (\S+)", _run([exe, "-n", "3000", "-d", "64", "-y", "100", "-o", "2"], env))
    assert m and float(m.group(1)) == 0.0


def test_time_results_prints_per_shard_bandwidth():
    out = _run([os.path.join(H, "time_results_f32"), "-n", "40000", "-d", "64", "-k", "10", "-y", "2000", "-o", "3", "-S", "7",
                "-V", "2", "-F"])
    assert "shard 0 of 2" in out and "shard 1 of 2" in out and "GB/s" in out
    assert "residency-cache fingerprint" in out


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_one_device_set_over_real_rccl(prec):
    """ANN_HIP_DEVICES=1 ANN_HIP_FORCE_RCCL=1: librccl.so loaded on demand, ncclCommInitAll, and every exchange of the step
    and of the build as a real RCCL call inside ncclGroupStart/End -- with a communicator of one, all a single-GPU box allows."""
    env = {"ANN_HIP_DEVICES": "1", "ANN_HIP_FORCE_RCCL": "1"}
    out = _run([os.path.join(H, "compare_results_" + prec), "-o", "1", "-S", "41"], env)
    assert "Average diffs for comp: 0" in out and "PASS" in out
    out = _run([os.path.join(H, "compare_results_" + prec), "-n", "5000", "-d", "64", "-y", "200", "-o", "2", "-S", "42"], env)
    assert "Average diffs for query: 0" in out and "PASS" in out
