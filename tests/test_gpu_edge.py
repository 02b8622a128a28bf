"""GPU: edge cases of the domain and miniature versions of BASELINE.json's other configurations."""
import ctypes as C

import numpy as np
import pytest
import torch

import approximatenn_amd as A
from oracle import oracle_py as O
from tests.util import assert_save_equal, bits_equal, load_golden

pytestmark = pytest.mark.gpu


def _data(prec, n, d, Q, seed):
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(seed)
    orc.rand_norm_reset()
    return orc, orc.gen_rand(n * d).reshape(n, d), orc.gen_rand(Q * d).reshape(Q, d)


def _both(prec, pts, y, k, T, seed=3, **rot):
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(seed)
    o = orc.precomp(pts, k, T, **rot)
    O.srandom(seed)
    names = dict(rb="rots_before", rlb="rot_len_before", ra="rots_after", rla="rot_len_after")
    g = A.precomp(pts, k, T, **{names[a]: v for a, v in rot.items()})
    return orc, o, g


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_cfg5_miniature_large_k(prec):
    """BASELINE configs[4] in small: d=256, k=100 (K1=101 selection buffers, L2=10100 -> 8193-entry network)."""
    orc, pts, y = _data(prec, 1500, 256, 24, 55)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both(prec, pts, y, 100, 2)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        assert_save_equal(save.to_dict(), o_save)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("prec,d,k,dup", [("f32", 64, 40, False), ("f64", 80, 33, False), ("f32", 33, 36, False),
                                          ("f32", 32, 45, True), ("f64", 128, 32, True)])
def test_long_stage2_rows_by_selection(prec, d, k, dup):
    """k >= 32: the stage-2 row (k(k+1) entries) no longer fits the fused kernel's LDS row; precomp's graph stage and
    query()'s stage 2 run stage2_select_kernel (selection + proof, literal path for the rows it flags).  dup: every point
    twice => equal distances between different ids everywhere => (nearly) every row is flagged and takes the fallback."""
    orc, pts, y = _data(prec, 2600, d, 40, 900 + d + k)
    if dup:
        pts[1300:] = pts[:1300]
        y[:10] = pts[5:15]
    pts, y = np.ascontiguousarray(pts), np.ascontiguousarray(y)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both(prec, pts, y, k, 3)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        assert_save_equal(save.to_dict(), o_save)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
        want, got = orc.query(o_save, pts, 300, alias=True), A.query(save, pts, pts[:300])
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("prec,d,k,group", [("f32", 64, 10, "4"), ("f64", 128, 70, "5"), ("f32", 32, 100, "7"), ("f64", 256, 64, None)])
def test_bucket_kernel_member_groups_and_two_key_lanes(prec, d, k, group, monkeypatch):
    """precomp's bucket-centric distance pass: k+1 > 64 keeps two keys per lane (wave_topk_insert2); buckets whose
    members' lists do not fit the LDS are scored in groups (forced here by ANN_HIP_BK_GROUP).  Few hash bits (large
    buckets): n = 3000 points in 2^d_short buckets with d_short = 5..6."""
    if group:
        monkeypatch.setenv("ANN_HIP_BK_GROUP", group)
        A._lib.reload_env()
    orc, pts, y = _data(prec, 3000, d, 16, 400 + d + k)
    pts[2900:] = pts[:100]          # some exact duplicates: ties inside the lists
    pts = np.ascontiguousarray(pts)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both(prec, pts, y, k, 2)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        assert_save_equal(save.to_dict(), o_save)
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("prec,d", [("f32", 36), ("f32", 50), ("f32", 77), ("f32", 100), ("f64", 100), ("f32", 150),
                                    ("f32", 250), ("f64", 72), ("f64", 77), ("f32", 200), ("f32", 300), ("f64", 150), ("f32", 1000), ("f64", 300)])
def test_row_lengths_folded_and_neighbours(prec, d):
    """Row lengths of the folded layouts (2, 3 or -- unaligned only -- 4 tree levels inside a lane), with odd levels at
    different depths (77: levels 1 and 3; 100: level 3; 150: level 2; 250: levels 2 and 4), and their neighbours that keep
    the lanes-per-row layout (200) or the tree through LDS (300)."""
    orc, pts, y = _data(prec, 900, d, 40, 300 + d)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both(prec, pts, y, 6, 3)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
        want, got = orc.query(o_save, pts, 200, alias=True), A.query(save, pts, pts[:200])
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("prec,d", [("f32", 96), ("f32", 160), ("f64", 48), ("f64", 80), ("f32", 24), ("f32", 40), ("f64", 24),
                                    ("f32", 192), ("f32", 320), ("f64", 96), ("f64", 160), ("f32", 384), ("f64", 192)])
def test_row_lengths_static_lane_groups(prec, d):
    """Row lengths whose 16-byte chunks split as 3 or 5 times a power of two: lane groups of 3, 5 (few chunks per lane)
    or 6, 10 lanes (3 x 8 and 5 x 8 chunks run as 6 x 4 and 10 x 4: d = 96 / 160 float, 48 / 80 double -- the reference
    drivers' default row in the stock double build; 3 x 16 and 5 x 16 chunks as 6 x 8 and 10 x 8: d = 192 / 320 float,
    96 / 160 double; 3 x 32 chunks as 12 x 8: d = 384 float, 192 double) inside the 16-lane DPP rows, tail of the tree unrolled at compile time.
    Precomp, query and aliased query against the oracle, bit for bit."""
    orc, pts, y = _data(prec, 1500, d, 60, 500 + d)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both(prec, pts, y, 7, 4)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        assert_save_equal(save.to_dict(), o_save)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
        want, got = orc.query(o_save, pts, 300, alias=True), A.query(save, pts, pts[:300])
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()


@pytest.mark.parametrize("d", [16, 512, 1024, 48, 130])
def test_row_lengths_fast_and_generic(d):
    """smallest / largest register-tiled d, and two generic (non power of two) ones; float."""
    orc, pts, y = _data("f32", 700, d, 30, 70 + d)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both("f32", pts, y, 5, 3)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load("f32").annhip_cache_clear()
        save.free()


def test_k_larger_than_sorted_prefix_forces_exact_path():
    """P(L1) < k: the top-k includes entries the network never sorts; only the literal path is right."""
    orc = O.CpuBackend("f32", "oracle")
    found = None
    for seed in range(200):
        O.srandom(seed)
        orc.rand_norm_reset()
        pts = orc.gen_rand(30 * 16).reshape(30, 16)
        O.srandom(seed)
        _, _, sv = orc.precomp(pts, 20, 1, ra=0)   # d_short is 1 here: a post-Walsh rotation needs 2 coordinates
        L1 = (sv["d_short"] + 1) * int(sv["par_maxes"][0])
        if (1 << int(np.floor(np.log2(L1)))) < 20:
            found = (seed, pts, sv)
            break
    assert found, "no seed with P(L1) < k"
    seed, pts, o_save = found
    y = orc.gen_rand(6 * 16).reshape(6, 16)
    O.srandom(seed)
    o_ids, o_d, _ = orc.precomp(pts, 20, 1, ra=0)
    O.srandom(seed)
    ids, dd, save = A.precomp(pts, 20, 1, rots_after=0)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    finally:
        A._lib.load("f32").annhip_cache_clear()
        save.free()


def test_empty_and_single_query_batches():
    g = load_golden("pow2_d32_f32")
    save = A.Save.from_dict("f32", g["save"])
    pts = np.ascontiguousarray(g["points"])
    ids, dd = A.query(save, pts, np.zeros((0, pts.shape[1]), dtype=np.float32))
    assert ids.shape == (0, 10) and dd.shape == (0, 10)
    orc = O.CpuBackend("f32", "oracle")
    one = g["y"][:1]
    want, got = orc.query(g["save"], pts, one), A.query(save, pts, one)    # Q=1: no scramble (Q2)
    assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    A._lib.load("f32").annhip_cache_clear()


def test_non_default_rotation_parameters():
    orc, pts, y = _data("f64", 900, 64, 40, 91)
    orc, (o_ids, o_d, o_save), (ids, dd, save) = _both("f64", pts, y, 7, 4, rb=3, rlb=8, ra=2, rla=3)
    try:
        assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
        assert_save_equal(save.to_dict(), o_save)
    finally:
        A._lib.load("f64").annhip_cache_clear()
        save.free()


def test_lifecycle_register_cleanup_and_reinit():
    lib = A._lib.load("f32")
    calls = []
    cb = C.CFUNCTYPE(None)(lambda: calls.append(1))
    lib.gpu_init()
    lib.register_cleanup(cb)          # initialised: runs at gpu_cleanup (gpu_comp.c:93-101)
    assert calls == []
    lib.gpu_cleanup()
    assert calls == [1]
    lib.register_cleanup(cb)          # not initialised: runs immediately
    assert calls == [1, 1]
    g = load_golden("tiny_appendixA_f32")
    save = A.Save.from_dict("f32", g["save"])
    ids, dd = A.query(save, np.ascontiguousarray(g["points"]), g["y"])   # re-initialises by itself
    assert np.array_equal(ids, g["query_ids"])
    lib.annhip_cache_clear()


def test_residency_cache_detects_changed_content():
    """Same host addresses, different content => the cached index must not be reused."""
    g = load_golden("pow2_d64_f32")
    save = A.Save.from_dict("f32", g["save"])
    pts = np.ascontiguousarray(g["points"]).copy()
    ids, _ = A.query(save, pts, g["y"])
    assert np.array_equal(ids, g["query_ids"])
    orc = O.CpuBackend("f32", "oracle")
    pts[:] = pts[::-1].copy()         # in place: same pointer, reversed rows
    want = orc.query(g["save"], pts, g["y"])
    got = A.query(save, pts, g["y"])
    assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
    A._lib.load("f32").annhip_cache_clear()


@pytest.mark.parametrize("exact_bytes", ["20000", "100000"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_network_in_hbm_and_chunked_host_driven_exact_path(prec, exact_bytes, monkeypatch):
    """Paths that only very large shapes reach (full-size cfg5 rows do not fit LDS; cfg4's Q=100k exceeds the exact
    path's workspace budget): force them with the size hooks on a ties-heavy dataset, in both path modes.
    exact_bytes 20000: a workspace of a few rows, batch > 8 workspaces => host-driven chunks; 100000: ~20 rows =>
    the device-driven path walks the flagged list in several passes over the bounded workspace."""
    orc = O.CpuBackend(prec, "oracle")
    O.srandom(77)
    orc.rand_norm_reset()
    half = orc.gen_rand(600 * 32).reshape(600, 32)
    pts = np.ascontiguousarray(np.concatenate([half, half]))     # duplicated points: most queries get rejected
    y = orc.gen_rand(70 * 32).reshape(70, 32)
    O.srandom(9)
    o_ids, o_d, o_save = orc.precomp(pts, 6, 4)
    want = orc.query(o_save, pts, y)
    monkeypatch.setenv("ANN_HIP_LDS_ROW_MAX", "64")              # every exact-path row sorts in HBM
    monkeypatch.setenv("ANN_HIP_EXACT_BYTES", exact_bytes)
    for exact in (False, True):
        if exact:
            monkeypatch.setenv("ANN_HIP_EXACT", "1")
        A._lib.reload_env()
        O.srandom(9)
        ids, dd, save = A.precomp(pts, 6, 4)
        try:
            assert np.array_equal(ids, o_ids) and bits_equal(dd, o_d)
            got = A.query(save, pts, y)
            assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
        finally:
            A._lib.load(prec).annhip_cache_clear()
            save.free()


@pytest.mark.parametrize("name", ["pow2_d64_f32", "k17_d100_f64", "defaults_d80_f32"])
def test_index_file_roundtrip_into_the_hip_path(name, tmp_path):
    """SURVEY 8(f)-1 end to end: build on the GPU, write the index file, read it back in a fresh Save, create a resident
    index from the FILE and match the reference's recorded answers bit for bit."""
    g = load_golden(name)
    c, prec = g["cfg"], g["prec"]
    pts = np.ascontiguousarray(g["points"])
    O.srandom(c["seed"])
    orc = O.CpuBackend(prec, "oracle")
    orc.rand_norm_reset()
    orc.gen_rand(c["n"] * c["d"] + (c["n"] * c["d"]) % 2)   # same libc stream position as the generator had
    ids, dd, save = A.precomp(pts, c["k"], c["tries"], c["rb"], c["rlb"], c["ra"], c["rla"])
    path = tmp_path / "index.ann"
    try:
        assert np.array_equal(ids, g["precomp_ids"])
        save.write(path)
    finally:
        A._lib.load(prec).annhip_cache_clear()
        save.free()
    back = A.Save.read(prec, path)
    try:
        assert_save_equal(back.to_dict(), g["save"])
        ix = A.Index.from_save(back, torch.from_numpy(pts).cuda())
        r_ids, r_d, _ = ix.query(torch.from_numpy(np.ascontiguousarray(g["y"])).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(r_ids.cpu().numpy().astype(np.uint64), g["query_ids"])
        assert bits_equal(r_d.cpu().numpy(), g["query_dists"])
        ix.close()
        q_ids, q_d = A.query(back, pts, g["y"])   # and through the drop-in symbol, host pointers
        assert np.array_equal(q_ids, g["query_ids"]) and bits_equal(q_d, g["query_dists"])
    finally:
        A._lib.load(prec).annhip_cache_clear()
        back.free()


def test_residency_cache_contract(monkeypatch):
    """One resident index per (save, points) key; rebuilt indexes replace older ones; edits of sampled content and of
    bases / row_means are detected; an edit of an UNSAMPLED table entry is detected in strict mode and after
    annhip_cache_drop(), and -- the documented contract -- not otherwise."""
    lib = A._lib.load("f32")
    lib.annhip_cache_clear()
    orc = O.CpuBackend("f32", "oracle")
    O.srandom(4242)
    orc.rand_norm_reset()
    n, d, k, T, Q = 6000, 32, 5, 3, 64
    pts = orc.gen_rand(n * d).reshape(n, d)
    y = orc.gen_rand(Q * d).reshape(Q, d)
    saves = []
    try:
        for rep in range(3):                      # the reference's drivers: same &save / points every iteration
            O.srandom(11 + rep)
            ids, dd, save = A.precomp(pts, k, T)
            saves.append(save)
            assert lib.annhip_cache_size() == rep + 1   # distinct save_t addresses here (Save objects are kept alive)
        lib.annhip_cache_clear()
        save = saves[-1]
        arrays = save.to_dict()
        want = orc.query(arrays, pts, y)
        got = A.query(save, pts, y)
        assert np.array_equal(got[0], want[0]) and bits_equal(got[1], want[1])
        assert lib.annhip_cache_size() == 1
        A.query(save, pts, y)
        assert lib.annhip_cache_size() == 1      # hit, not a second upload
        # --- an unsampled table entry: swap two ids inside one bucket row far from the strided samples
        wp0 = np.ctypeslib.as_array(save.c.which_par[0], shape=((1 << int(save.c.d_short)) * int(save.c.par_maxes[0]),))
        cnt = wp0.size
        sampled = {(cnt - 1) * i // 1023 for i in range(1024)} if cnt > 1024 else set(range(cnt))
        pm0 = int(save.c.par_maxes[0])
        pos = next(p for p in range(0, cnt - 1, pm0) if wp0[p] < n and wp0[p + 1] < n and p not in sampled and p + 1 not in sampled)
        wp0[pos], wp0[pos + 1] = wp0[pos + 1], wp0[pos]          # in place, same addresses
        want2 = orc.query(save.to_dict(), pts, y)
        stale = A.query(save, pts, y)
        assert np.array_equal(stale[0], want[0])                 # documented: sampled fingerprint does not see it
        lib.annhip_cache_drop(save.c)
        assert lib.annhip_cache_size() == 0
        fresh = A.query(save, pts, y)
        assert np.array_equal(fresh[0], want2[0]) and bits_equal(fresh[1], want2[1])
        wp0[pos], wp0[pos + 1] = wp0[pos + 1], wp0[pos]          # back
        monkeypatch.setenv("ANN_HIP_CACHE", "strict")
        A._lib.reload_env()
        lib.annhip_cache_clear()
        a1 = A.query(save, pts, y)
        assert np.array_equal(a1[0], want[0])
        wp0[pos], wp0[pos + 1] = wp0[pos + 1], wp0[pos]
        a2 = A.query(save, pts, y)                               # strict: full content hash sees the swap
        assert np.array_equal(a2[0], want2[0]) and bits_equal(a2[1], want2[1])
        assert lib.annhip_cache_size() == 1
        monkeypatch.setenv("ANN_HIP_CACHE", "off")
        A._lib.reload_env()
        lib.annhip_cache_clear()
        a3 = A.query(save, pts, y)
        assert np.array_equal(a3[0], want2[0]) and lib.annhip_cache_size() == 0
    finally:
        monkeypatch.delenv("ANN_HIP_CACHE", raising=False)
        A._lib.reload_env()
        lib.annhip_cache_clear()
        for s_ in saves:
            s_.free()
