"""Loader for the HIP backend's shared libraries (one per precision, see csrc/Makefile).

The product has no CPU implementation: if the library is missing or there is no HIP device, calls fail
loudly (ImportError here, "No GPU found." + exit(1) from gpu_init() in the library).
"""
import ctypes as C
import os
import subprocess
import threading

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_libs = {}


class SaveT(C.Structure):
    """save_t -- include/ann.h (layout of /root/reference/ann.h:8-12)."""
    _fields_ = [("tries", C.c_int), ("n", C.c_size_t), ("k", C.c_size_t),
                ("d_short", C.c_size_t), ("d_long", C.c_size_t),
                ("which_par", C.POINTER(C.POINTER(C.c_size_t))),
                ("par_maxes", C.POINTER(C.c_size_t)), ("graph", C.POINTER(C.c_size_t)),
                ("row_means", C.c_void_p), ("bases", C.c_void_p)]


def build(verbose=False):
    """Compile every HIP extension for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "all"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def reload_env():
    """Make every loaded library re-read the ANN_HIP_* switches (they are cached after the first call)."""
    for lib in _libs.values():
        lib.annhip_reload_env()


def lib_path(prec):
    # ANN_HIP_LIBDIR: load the backend from another directory (A/B builds of a kernel variant, tools/ab_build.sh)
    return os.path.join(os.environ.get("ANN_HIP_LIBDIR", CSRC), "libapproxnn_hip_%s.so" % prec)


def load(prec="f32"):
    assert prec in ("f32", "f64")
    if prec in _libs:
        return _libs[prec]
    # One HIP runtime per process: torch preloads its bundled libamdhip64.so.7 by path, so it has to be in the
    # process BEFORE this library's NEEDED libamdhip64.so.7 is resolved (same SONAME => the loaded one is reused).
    # Loading in the other order leaves two runtimes and the second one finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path(prec)
    if not os.path.exists(path):
        raise ImportError("HIP backend %s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C approximatenn_amd/csrc`" % path)
    lib = C.CDLL(path)
    vp, sz, u32p = C.c_void_p, C.c_size_t, C.c_void_p
    lib.annhip_precision.restype = C.c_char_p
    lib.annhip_index_create.restype = vp
    lib.annhip_index_create.argtypes = [C.POINTER(SaveT), vp, C.c_int, sz, sz]
    lib.annhip_index_destroy.argtypes = [vp]
    lib.annhip_index_info.argtypes = [vp, C.POINTER(sz * 12)]
    lib.annhip_index_set_stream.argtypes = [vp, vp]
    lib.annhip_index_set_gather_pieces.argtypes = [vp, C.c_int]
    lib.annhip_index_set_gather_slots.argtypes = [vp, C.c_int]
    lib.annhip_index_set_fixed.argtypes = [vp, C.c_int]
    lib.annhip_index_export.argtypes = [vp, C.POINTER(SaveT)]
    lib.annhip_index_reshard.argtypes = [vp, vp, sz, sz]
    lib.annhip_save_write.restype = C.c_int
    lib.annhip_save_write.argtypes = [C.POINTER(SaveT), C.c_char_p]
    lib.annhip_save_read.restype = C.c_int
    lib.annhip_save_read.argtypes = [C.c_char_p, C.POINTER(SaveT)]
    lib.annhip_precomp_index.restype = vp
    lib.annhip_precomp_index.argtypes = [sz, sz, sz, vp, C.c_int, C.c_int, sz, sz, sz, sz, vp]
    lib.annhip_precomp_begin.restype = vp
    lib.annhip_precomp_begin.argtypes = [sz, sz, sz, vp, C.c_int, C.c_int, sz, sz, sz, sz, C.c_int, C.c_int]
    lib.annhip_precomp_info.argtypes = [vp, C.POINTER(sz * 6)]
    lib.annhip_precomp_init_merged.argtypes = [vp, u32p, vp]
    lib.annhip_precomp_hash.argtypes = [vp, C.c_int, sz, sz, u32p]
    lib.annhip_precomp_try.argtypes = [vp, C.c_int, u32p, u32p, vp]
    lib.annhip_precomp_merge.argtypes = [vp, u32p, vp]
    lib.annhip_precomp_graph.argtypes = [vp, sz, sz, u32p, vp]
    lib.annhip_precomp_finish.restype = vp
    lib.annhip_precomp_finish.argtypes = [vp, u32p]
    lib.annhip_query.restype = C.c_long
    lib.annhip_query.argtypes = [vp, sz, vp, C.c_int, C.c_int, vp, vp]
    lib.annhip_workspace_create.restype = vp
    lib.annhip_workspace_create.argtypes = [vp]
    lib.annhip_workspace_destroy.argtypes = [vp]
    lib.annhip_query_on.restype = C.c_long
    lib.annhip_query_on.argtypes = [vp, vp, vp, sz, vp, C.c_int, C.c_int, vp, vp]
    lib.annhip_query_slice.restype = C.c_long
    lib.annhip_query_slice.argtypes = [vp, vp, vp, sz, sz, sz, vp, u32p, C.c_int, vp, vp]
    lib.annhip_stream_open.restype = vp
    lib.annhip_stream_open.argtypes = [vp, sz, C.c_int]
    lib.annhip_stream_submit.restype = C.c_long
    lib.annhip_stream_submit.argtypes = [vp, sz, vp, C.c_int]
    lib.annhip_stream_collect.restype = C.c_int
    lib.annhip_stream_collect.argtypes = [vp, C.c_long, vp, vp]
    lib.annhip_stream_close.argtypes = [vp]
    lib.annhip_key_bytes.restype = sz
    lib.annhip_key_bytes.argtypes = []
    lib.annhip_stream_create_reserving.restype = vp
    lib.annhip_stream_create_reserving.argtypes = [C.c_int]
    lib.annhip_stream_destroy.argtypes = [vp]
    lib.annhip_sh_codes.argtypes = [vp, vp, sz, vp, sz, sz, u32p]
    lib.annhip_sh_stage1.argtypes = [vp, vp, sz, vp, C.c_int, u32p, vp, u32p, u32p]
    lib.annhip_sh_merge_finalize.argtypes = [vp, vp, C.c_int, sz, sz, sz, vp, u32p, u32p, vp]
    lib.annhip_sh_exact1_begin.argtypes = [vp, vp, sz, vp, C.c_int, u32p, u32p, sz, u32p, u32p, vp]
    lib.annhip_sh_exact1_end.argtypes = [vp, vp, sz, sz, sz, sz, u32p, u32p, vp, u32p, vp, u32p, vp]
    lib.annhip_sh_stage2.argtypes = [vp, vp, sz, vp, C.c_int, u32p, vp, u32p]
    lib.annhip_sh_final.argtypes = [vp, vp, C.c_int, sz, sz, sz, u32p, vp, vp, u32p, vp]
    lib.annhip_stage1_rows.argtypes = [vp, sz, vp, C.c_int, u32p, u32p, sz, u32p, vp]
    lib.annhip_stage2_rows_list.argtypes = [vp, sz, vp, C.c_int, u32p, sz, u32p, vp, u32p, vp]
    lib.annhip_exact_select.argtypes = [vp, C.c_int, sz, u32p, vp, u32p, u32p, vp]
    lib.annhip_test_sort_rows.argtypes = [sz, sz, sz, u32p, vp, vp, u32p, u32p, vp, vp, C.c_int]
    lib.annhip_recall_ranks.argtypes = [sz, sz, sz, vp, sz, vp, vp, C.c_int, vp]
    lib.annhip_recall_ranks_host.argtypes = [sz, sz, sz, vp, sz, vp, vp, C.c_int, vp]
    lib.annhip_checksum_dev.restype = C.c_ulonglong
    lib.annhip_checksum_dev.argtypes = [vp, sz, vp]
    lib.annhip_index_checksum.restype = C.c_ulonglong
    lib.annhip_index_checksum.argtypes = [vp]
    lib.annhip_profile.argtypes = [vp, C.c_int]
    lib.annhip_stats.argtypes = [vp, C.POINTER(C.c_double * 8), C.c_int]
    lib.annhip_stage_ms.argtypes = [vp, C.POINTER(C.c_double * 6)]
    lib.annhip_cache_clear.argtypes = []
    lib.annhip_fingerprint_ms.restype = C.c_double
    lib.annhip_fingerprint_ms.argtypes = [C.POINTER(SaveT), vp, C.c_int]
    lib.annhip_host_profile.argtypes = [C.c_int]
    lib.annhip_host_stats.restype = C.c_int
    lib.annhip_host_stats.argtypes = [C.POINTER(SaveT), C.POINTER(C.c_double * 8), C.c_int]
    lib.annhip_host_shards.restype = C.c_int
    lib.annhip_host_shards.argtypes = [C.POINTER(SaveT)]
    lib.annhip_host_stats_shard.restype = C.c_int
    lib.annhip_host_stats_shard.argtypes = [C.POINTER(SaveT), C.c_int, C.POINTER(C.c_double * 8), C.c_int]
    lib.annhip_set_devices.argtypes = [C.c_int, C.c_int]
    lib.annhip_cache_drop.argtypes = [C.POINTER(SaveT)]
    lib.annhip_cache_size.restype = sz
    lib.annhip_cache_size.argtypes = []
    lib.annhip_reload_env.argtypes = []
    lib.annhip_synth_randnorm.argtypes = [sz, vp]
    lib.annhip_synth_reset.argtypes = []
    lib.gpu_init.argtypes = []
    lib.gpu_cleanup.argtypes = []
    lib.register_cleanup.argtypes = [C.CFUNCTYPE(None)]
    lib.query_gpu.restype = C.POINTER(sz)
    lib.query_gpu.argtypes = [C.POINTER(SaveT), vp, sz, vp, C.POINTER(vp)]
    lib.precomp_gpu.restype = C.POINTER(sz)
    lib.precomp_gpu.argtypes = [sz, sz, sz, vp, C.c_int, sz, sz, sz, sz, C.POINTER(SaveT), C.POINTER(vp)]
    _libs[prec] = lib
    return lib


# every symbol include/*.h declares for the backend library (checked by tests/test_abi.py)
EXPORTED = ["gpu_init", "gpu_cleanup", "register_cleanup", "query_gpu", "precomp_gpu", "annhip_precision",
            "annhip_index_create", "annhip_index_destroy", "annhip_index_info", "annhip_index_set_stream", "annhip_index_set_gather_pieces", "annhip_index_set_gather_slots", "annhip_index_set_fixed",
            "annhip_index_export", "annhip_index_reshard", "annhip_save_write", "annhip_save_read", "annhip_precomp_index", "annhip_precomp_begin", "annhip_precomp_info", "annhip_precomp_init_merged", "annhip_precomp_hash",
            "annhip_precomp_try", "annhip_precomp_merge", "annhip_precomp_graph", "annhip_precomp_finish", "annhip_query", "annhip_workspace_create", "annhip_workspace_destroy", "annhip_query_on", "annhip_query_slice", "annhip_stream_open", "annhip_stream_submit", "annhip_stream_collect", "annhip_stream_close",
            "annhip_key_bytes", "annhip_stream_create_reserving", "annhip_stream_destroy", "annhip_sh_codes", "annhip_sh_stage1", "annhip_sh_merge_finalize", "annhip_sh_exact1_begin", "annhip_sh_exact1_end", "annhip_sh_stage2",
            "annhip_sh_final", "annhip_stage1_rows", "annhip_stage2_rows_list", "annhip_exact_select", "annhip_test_sort_rows",
            "annhip_recall_ranks", "annhip_recall_ranks_host", "annhip_checksum_dev", "annhip_index_checksum", "annhip_profile", "annhip_stats", "annhip_stage_ms",
            "annhip_cache_clear", "annhip_fingerprint_ms", "annhip_cache_drop", "annhip_cache_size", "annhip_reload_env",
            "annhip_synth_randnorm", "annhip_synth_reset", "annhip_host_profile", "annhip_host_stats",
            "annhip_host_shards", "annhip_host_stats_shard", "annhip_set_devices"]
DISPATCH_EXPORTED = ["precomp", "query", "free_save"]


# ---- the caller's libc random() stream (the workload's data and precomp's rotations are drawn from it) ----
_libc = C.CDLL("libc.so.6")
_libc.initstate.restype = C.c_void_p
_libc.initstate.argtypes = [C.c_uint, C.c_void_p, C.c_size_t]
_libc.setstate.restype = C.c_void_p
_libc.setstate.argtypes = [C.c_void_p]
_RAND_WORDS = (1, 8, 16, 32, 64)   # glibc random_r.c: state words (degree + 1) of TYPE_0..TYPE_4; word 0 = rear*5 + type
_restored = []                      # state buffers handed to setstate() must outlive the call


_park_lock = threading.Lock()
_park_depth = 0
_park_old = None


class park_random:
    """Context manager: libc random() draws made inside (the HIP runtime draws while it initialises) do not touch the
    caller's stream.  The library's own entry points do the same (RandGuard, csrc/ann_host.hip).  Nestable, and several
    host threads may be inside at once (rank threads of the tests): only the outermost one swaps the state."""

    def __enter__(self):
        global _park_depth, _park_old
        with _park_lock:
            if _park_depth == 0:
                # 128 bytes = the generator type of libc's default state (TYPE_3): a thread that seeds and draws while
                # another one holds the park (rank threads of the tests) gets the sequence it would get unparked
                buf = C.create_string_buffer(128)
                _restored.append(buf)               # a state buffer glibc has seen is never freed
                _park_old = _libc.initstate(1, buf, 128)
            _park_depth += 1
        return self

    def __exit__(self, *exc):
        global _park_depth
        with _park_lock:
            _park_depth -= 1
            if _park_depth == 0:
                _libc.setstate(_park_old)


def random_state_snapshot():
    """The complete state of the process's current libc random() stream as bytes (glibc layout: switching states makes
    glibc store the rear pointer into word 0 of the state it leaves).  random_state_restore() on ANY process continues
    the stream from exactly there -- how one rank hands the benchmark's stream to the others."""
    scratch = C.create_string_buffer(256)
    cur = _libc.initstate(1, scratch, 256)          # leaves the current state; its word 0 now holds rear*5 + type
    word0 = C.cast(cur, C.POINTER(C.c_int32))[0]
    nbytes = 4 * _RAND_WORDS[word0 % 5]
    snap = C.string_at(cur, nbytes)
    _libc.setstate(cur)
    return snap


def random_state_restore(snap):
    buf = C.create_string_buffer(bytes(snap), len(snap))
    _restored.append(buf)
    _libc.setstate(buf)
