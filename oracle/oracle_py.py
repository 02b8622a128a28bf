"""ctypes bindings for the CPU checker libraries.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module (see the header of ``oracle/ann_oracle.c``).  It wraps

* ``oracle/liboracle_{f32,f64}.so`` -- the clean-room restatement (``kind="oracle"``), and
* ``oracle/_ref/libref_{f32,f64}.so`` -- the reference's own CPU path compiled from
  ``/root/reference`` by ``oracle/Makefile`` (``kind="ref"``; exists only where it was built).

Both expose the same Python surface so tests can run one against the other.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libc = C.CDLL("libc.so.6")
_libc.srandom.argtypes = [C.c_uint]
_libc.random.restype = C.c_long
_libc.free.argtypes = [C.c_void_p]


def srandom(seed):
    _libc.srandom(seed)


def libc_random():
    return _libc.random()


class SaveT(C.Structure):
    """save_t, /root/reference/ann.h:8-12 (pointer fields are precision-agnostic here)."""
    _fields_ = [("tries", C.c_int), ("n", C.c_size_t), ("k", C.c_size_t),
                ("d_short", C.c_size_t), ("d_long", C.c_size_t),
                ("which_par", C.POINTER(C.POINTER(C.c_size_t))),
                ("par_maxes", C.POINTER(C.c_size_t)), ("graph", C.POINTER(C.c_size_t)),
                ("row_means", C.c_void_p), ("bases", C.c_void_p)]


class QueryStats(C.Structure):
    _fields_ = [("L1", C.c_size_t), ("P1", C.c_size_t), ("L2", C.c_size_t), ("P2", C.c_size_t),
                ("valid1", C.c_ulonglong), ("valid2", C.c_ulonglong)]


def build(force=False):
    """Compile the checker (gcc).  Building the checker is not using it."""
    need = force or not all(os.path.exists(os.path.join(HERE, f))
                            for f in ("liboracle_f32.so", "liboracle_f64.so"))
    src_newer = False
    try:
        src_newer = os.path.getmtime(os.path.join(HERE, "ann_oracle.c")) > \
            os.path.getmtime(os.path.join(HERE, "liboracle_f32.so"))
    except OSError:
        pass
    if need or src_newer:
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle_f32.so", "liboracle_f64.so"])
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(HERE, "_ref", "libref_f32.so")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_f32.so"))


def save_to_arrays(save, prec):
    """Deep-copy a C save_t into numpy arrays (dict)."""
    ft = np.float32 if prec == "f32" else np.float64
    T, n, k, ds, d = save.tries, save.n, save.k, save.d_short, save.d_long
    pm = np.ctypeslib.as_array(save.par_maxes, shape=(T,)).copy()
    out = {"tries": T, "n": n, "k": k, "d_short": ds, "d_long": d, "par_maxes": pm.astype(np.uint64)}
    out["graph"] = np.ctypeslib.as_array(save.graph, shape=(n, k)).copy().astype(np.uint64)
    out["which_par"] = [np.ctypeslib.as_array(save.which_par[t], shape=(1 << ds, int(pm[t]))).copy().astype(np.uint64)
                        for t in range(T)]
    out["row_means"] = np.ctypeslib.as_array(C.cast(save.row_means, C.POINTER(np.ctypeslib.as_ctypes_type(ft))),
                                             shape=(d,)).copy()
    out["bases"] = np.ctypeslib.as_array(C.cast(save.bases, C.POINTER(np.ctypeslib.as_ctypes_type(ft))),
                                         shape=(T, ds, d)).copy()
    return out


class HostSave:
    """A save_t whose memory is owned by numpy arrays (so any backend can consume it)."""

    def __init__(self, arrays, prec):
        self.prec = prec
        self.a = arrays
        T = int(arrays["tries"])
        self._wp = [np.ascontiguousarray(w, dtype=np.uint64) for w in arrays["which_par"]]
        self._pm = np.ascontiguousarray(arrays["par_maxes"], dtype=np.uint64)
        self._graph = np.ascontiguousarray(arrays["graph"], dtype=np.uint64)
        ft = np.float32 if prec == "f32" else np.float64
        self._means = np.ascontiguousarray(arrays["row_means"], dtype=ft)
        self._bases = np.ascontiguousarray(arrays["bases"], dtype=ft)
        self._wp_ptrs = (C.POINTER(C.c_size_t) * T)(*[w.ctypes.data_as(C.POINTER(C.c_size_t)) for w in self._wp])
        s = SaveT()
        s.tries = T
        s.n, s.k = int(arrays["n"]), int(arrays["k"])
        s.d_short, s.d_long = int(arrays["d_short"]), int(arrays["d_long"])
        s.which_par = C.cast(self._wp_ptrs, C.POINTER(C.POINTER(C.c_size_t)))
        s.par_maxes = self._pm.ctypes.data_as(C.POINTER(C.c_size_t))
        s.graph = self._graph.ctypes.data_as(C.POINTER(C.c_size_t))
        s.row_means = self._means.ctypes.data
        s.bases = self._bases.ctypes.data
        self.c = s


class CpuBackend:
    """precomp/query on the CPU through the oracle (kind='oracle') or the compiled reference (kind='ref')."""

    def __init__(self, prec="f32", kind="oracle"):
        assert prec in ("f32", "f64") and kind in ("oracle", "ref")
        self.prec, self.kind = prec, kind
        self.ft = np.float32 if prec == "f32" else np.float64
        self.cft = C.c_float if prec == "f32" else C.c_double
        if kind == "oracle":
            build()
            self.lib = C.CDLL(os.path.join(HERE, "liboracle_%s.so" % prec))
            pre = "oracle_"
            self._precomp = self.lib.oracle_precomp
            self._query = self.lib.oracle_query
            self._free_save = self.lib.oracle_free_save
            self.lib.oracle_query_ex.restype = C.POINTER(C.c_size_t)
            self.lib.oracle_query_ex.argtypes = [C.POINTER(SaveT), C.c_void_p, C.c_size_t, C.c_void_p,
                                                 C.POINTER(C.c_void_p), C.c_int, C.POINTER(QueryStats)]
            self.lib.oracle_gen_rand.argtypes = [C.c_size_t, C.c_void_p]
            self.lib.oracle_sort_net.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
            self.lib.oracle_topk_stage.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
            self.lib.oracle_tree_sum.argtypes = [C.c_size_t, C.c_void_p]
            self.lib.oracle_tree_sum.restype = self.cft
            self.lib.oracle_query_codes.argtypes = [C.POINTER(SaveT), C.c_size_t, C.c_void_p, C.c_void_p]
            self.lib.oracle_rand_perm.restype = C.POINTER(C.c_size_t)
            self.lib.oracle_rand_perm.argtypes = [C.c_size_t, C.c_size_t]
        else:
            self.lib = C.CDLL(os.path.join(HERE, "_ref", "libref_%s.so" % prec))
            self._precomp = self.lib.precomp_cpu
            self._query = self.lib.query_cpu
            self._free_save = None
            self.lib.rand_norm.restype = C.c_double
            self.lib.do_sort_cpu.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
            self.lib.sort_and_uniq_cpu.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
            self.lib.rand_perm.restype = C.POINTER(C.c_size_t)
            self.lib.rand_perm.argtypes = [C.c_size_t, C.c_size_t]
        self._precomp.restype = C.POINTER(C.c_size_t)
        self._precomp.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t,
                                  C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(SaveT), C.POINTER(C.c_void_p)]
        self._query.restype = C.POINTER(C.c_size_t)
        self._query.argtypes = [C.POINTER(SaveT), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]

    # -- data generation from the libc stream (time_results.c:10-13) --
    def gen_rand(self, count):
        out = np.empty(count, dtype=self.ft)
        if self.kind == "oracle":
            self.lib.oracle_gen_rand(count, out.ctypes.data)
        else:
            for i in range(count):
                out[i] = self.lib.rand_norm()
        return out

    def rand_norm_reset(self):
        if self.kind == "oracle":
            self.lib.oracle_rand_norm_reset()

    def _take(self, ptr, count, ctype, dtype):
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).copy().astype(dtype)
        _libc.free(C.cast(ptr, C.c_void_p))
        return arr

    def precomp(self, points, k, tries=10, rb=6, rlb=1, ra=1, rla=1, want_save=True):
        points = np.ascontiguousarray(points, dtype=self.ft)
        n, d = points.shape
        save = SaveT()
        dptr = C.c_void_p()
        ids = self._precomp(n, k, d, points.ctypes.data, tries, rb, rlb, ra, rla,
                            C.byref(save) if want_save else None, C.byref(dptr))
        ids = self._take(ids, n * k, C.c_size_t, np.uint64).reshape(n, k)
        dists = self._take(dptr, n * k, self.cft, self.ft).reshape(n, k)
        arrays = None
        if want_save:
            arrays = save_to_arrays(save, self.prec)
            if self._free_save is not None:
                self._free_save(C.byref(save))
            else:  # ann.c:25-34
                for t in range(save.tries):
                    _libc.free(C.cast(save.which_par[t], C.c_void_p))
                for p in (save.which_par, save.par_maxes, save.graph):
                    _libc.free(C.cast(p, C.c_void_p))
                _libc.free(save.row_means), _libc.free(save.bases)
        return ids, dists, arrays

    def precomp_tables_sample(self, points, k, tries, rows, rb=6, rlb=1, ra=1, rla=1):
        """oracle_precomp_tables_sample: means, bases and the hash codes of the sampled rows, drawing the transforms
        from libc random() exactly as a full precomp does.  Returns (d_short, means[d], bases[T,ds,d], codes[nrows,T])."""
        assert self.kind == "oracle"
        points = np.ascontiguousarray(points, dtype=self.ft)
        n, d = points.shape
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        ds = int(np.ceil(np.log2(self.ft(n) / self.ft(k))))
        d_max = 1
        while d_max < d:
            d_max *= 2
        ds = min(ds, d_max)
        means = np.empty(d, dtype=self.ft)
        bases = np.empty((tries, ds, d), dtype=self.ft)
        codes = np.empty((len(rows), tries), dtype=np.uint64)
        f = self.lib.oracle_precomp_tables_sample
        f.restype = C.c_size_t
        f.argtypes = [C.c_size_t] * 3 + [C.c_void_p, C.c_int] + [C.c_size_t] * 5 + [C.c_void_p] * 4
        got = f(n, k, d, points.ctypes.data, tries, rb, rlb, ra, rla, len(rows), rows.ctypes.data, means.ctypes.data,
                bases.ctypes.data, codes.ctypes.data)
        assert got == ds, (got, ds)
        return ds, means, bases, codes

    def precomp_graph_rows(self, save_arrays, points, rows):
        """oracle_precomp_graph_rows: the graph rows precomp returns for the sampled points, from a complete save_t."""
        assert self.kind == "oracle"
        hs = save_arrays if isinstance(save_arrays, HostSave) else HostSave(save_arrays, self.prec)
        points = np.ascontiguousarray(points, dtype=self.ft)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        k = int(hs.c.k)
        ids = np.empty((len(rows), k), dtype=np.uint64)
        dd = np.empty((len(rows), k), dtype=self.ft)
        f = self.lib.oracle_precomp_graph_rows
        f.restype = C.c_int
        f.argtypes = [C.POINTER(SaveT), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        if f(C.byref(hs.c), points.ctypes.data, len(rows), rows.ctypes.data, ids.ctypes.data, dd.ctypes.data) != 0:
            raise ValueError("a point is missing from a bucket table")
        return ids, dd

    def query(self, save_arrays, points, y, alias=False, stats=False):
        """alias=True reproduces the y==points pointer-equality self exclusion (Q3)."""
        hs = save_arrays if isinstance(save_arrays, HostSave) else HostSave(save_arrays, self.prec)
        points = np.ascontiguousarray(points, dtype=self.ft)
        if alias:
            y = points[: len(y)] if not isinstance(y, int) else points[:y]
            assert y.ctypes.data == points.ctypes.data
        else:
            y = np.array(y, dtype=self.ft, order="C", copy=True)
        ycnt = y.shape[0]
        k = int(hs.c.k)
        dptr = C.c_void_p()
        if stats:
            assert self.kind == "oracle"
            st = QueryStats()
            ids = self.lib.oracle_query_ex(C.byref(hs.c), points.ctypes.data, ycnt, y.ctypes.data,
                                           C.byref(dptr), int(alias), C.byref(st))
        else:
            ids = self._query(C.byref(hs.c), points.ctypes.data, ycnt, y.ctypes.data, C.byref(dptr))
        ids = self._take(ids, ycnt * k, C.c_size_t, np.uint64).reshape(ycnt, k)
        dists = self._take(dptr, ycnt * k, self.cft, self.ft).reshape(ycnt, k)
        if stats:
            return ids, dists, {f: getattr(st, f) for f, _ in QueryStats._fields_}
        return ids, dists

    # -- sub-function probes --
    def sort_net(self, ids, keys):
        ids = np.array(ids, dtype=np.uint64)
        keys = np.array(keys, dtype=self.ft)
        if self.kind == "oracle":
            self.lib.oracle_sort_net(len(ids), ids.ctypes.data, keys.ctypes.data)
        else:
            self.lib.do_sort_cpu(len(ids), 1, ids.ctypes.data, keys.ctypes.data)
        return ids, keys

    def topk_stage(self, ids, keys):
        ids = np.array(ids, dtype=np.uint64)
        keys = np.array(keys, dtype=self.ft)
        if self.kind == "oracle":
            self.lib.oracle_topk_stage(len(ids), ids.ctypes.data, keys.ctypes.data)
        else:
            self.lib.sort_and_uniq_cpu(1, len(ids), ids.ctypes.data, keys.ctypes.data)
        return ids, keys

    def rand_perm(self, d_pre, d_post):
        f = self.lib.oracle_rand_perm if self.kind == "oracle" else self.lib.rand_perm
        p = f(d_pre, d_post)
        return self._take(p, d_post, C.c_size_t, np.uint64)

    def tree_sum(self, values):
        assert self.kind == "oracle"
        v = np.array(values, dtype=self.ft)
        return self.ft(self.lib.oracle_tree_sum(len(v), v.ctypes.data))

    def query_codes(self, save_arrays, y):
        assert self.kind == "oracle"
        hs = save_arrays if isinstance(save_arrays, HostSave) else HostSave(save_arrays, self.prec)
        y = np.ascontiguousarray(y, dtype=self.ft)
        out = np.empty(y.shape[0] * hs.c.tries, dtype=np.uint64)
        self.lib.oracle_query_codes(C.byref(hs.c), y.shape[0], y.ctypes.data, out.ctypes.data)
        return out
