"""Host-side mirror of the reference's operator interface for the precomp/query path.

Names and argument meaning follow /root/reference/ann.h:46-49,61-62 (precomp, query, save_t, free_save); the
numpy arrays stand where the reference takes malloc'd C arrays.  Everything runs through the C-ABI of
csrc/libapproxnn_hip_{f32,f64}.so -- there is no Python or CPU implementation of the path in this package.

    save = Save.from_arrays(...)            # or: ids, dists, save = precomp(points, k, ...)
    ids, dists = query(save, points, y)     # drop-in symbols precomp_gpu / query_gpu (host pointers)
    ix = Index.from_save(save, points)      # resident index (HBM) for repeated queries
    ids, dists = ix.query(y_torch)          # device tensors in, device tensors out
"""
import ctypes as C

import numpy as np

from . import _lib

_libc = C.CDLL("libc.so.6")
_libc.free.argtypes = [C.c_void_p]


def _ft(prec):
    return np.float32 if prec == "f32" else np.float64


def _prec_of(arr):
    if arr.dtype == np.float32:
        return "f32"
    if arr.dtype == np.float64:
        return "f64"
    raise TypeError("points must be float32 (USE_FLOAT build) or float64 (stock build), got %s" % arr.dtype)


class Save:
    """A save_t (include/ann.h).  Either owns numpy-backed memory or wraps a malloc'd struct the library filled."""

    def __init__(self, prec):
        self.prec = prec
        self.c = _lib.SaveT()
        self._keep = None
        self._malloced = False

    @classmethod
    def from_arrays(cls, prec, tries, n, k, d_short, d_long, which_par, par_maxes, graph, row_means, bases):
        s = cls(prec)
        ft = _ft(prec)
        wp = [np.ascontiguousarray(w, dtype=np.uint64) for w in which_par]
        pm = np.ascontiguousarray(par_maxes, dtype=np.uint64)
        gr = np.ascontiguousarray(graph, dtype=np.uint64)
        rm = np.ascontiguousarray(row_means, dtype=ft)
        bs = np.ascontiguousarray(bases, dtype=ft)
        ptrs = (C.POINTER(C.c_size_t) * int(tries))(*[w.ctypes.data_as(C.POINTER(C.c_size_t)) for w in wp])
        s._keep = (wp, pm, gr, rm, bs, ptrs)
        s.c.tries = int(tries)
        s.c.n, s.c.k, s.c.d_short, s.c.d_long = int(n), int(k), int(d_short), int(d_long)
        s.c.which_par = C.cast(ptrs, C.POINTER(C.POINTER(C.c_size_t)))
        s.c.par_maxes = pm.ctypes.data_as(C.POINTER(C.c_size_t))
        s.c.graph = gr.ctypes.data_as(C.POINTER(C.c_size_t))
        s.c.row_means = rm.ctypes.data
        s.c.bases = bs.ctypes.data
        return s

    @classmethod
    def from_dict(cls, prec, a):
        return cls.from_arrays(prec, a["tries"], a["n"], a["k"], a["d_short"], a["d_long"], a["which_par"],
                               a["par_maxes"], a["graph"], a["row_means"], a["bases"])

    def to_dict(self):
        """Deep copy of every field into numpy arrays."""
        c, ft = self.c, _ft(self.prec)
        T, n, k, ds, d = c.tries, c.n, c.k, c.d_short, c.d_long
        pm = np.ctypeslib.as_array(c.par_maxes, shape=(T,)).copy()
        cft = np.ctypeslib.as_ctypes_type(ft)
        return dict(tries=T, n=n, k=k, d_short=ds, d_long=d, par_maxes=pm.astype(np.uint64),
                    graph=np.ctypeslib.as_array(c.graph, shape=(n, k)).copy().astype(np.uint64),
                    which_par=[np.ctypeslib.as_array(c.which_par[t], shape=(1 << ds, int(pm[t]))).copy().astype(np.uint64)
                               for t in range(T)],
                    row_means=np.ctypeslib.as_array(C.cast(c.row_means, C.POINTER(cft)), shape=(d,)).copy(),
                    bases=np.ctypeslib.as_array(C.cast(c.bases, C.POINTER(cft)), shape=(T, ds, d)).copy())

    def write(self, path):
        """annhip_save_write: one checksummed index file (include/ann_hip.h)."""
        if _lib.load(self.prec).annhip_save_write(C.byref(self.c), str(path).encode()) != 0:
            raise OSError("could not write index file %s" % path)

    @classmethod
    def read(cls, prec, path):
        """annhip_save_read: load an index file written by the same-precision library."""
        s = cls(prec)
        if _lib.load(prec).annhip_save_read(str(path).encode(), C.byref(s.c)) != 0:
            raise OSError("could not read index file %s" % path)
        s._malloced = True
        return s

    def free(self):
        """free_save (/root/reference/ann.c:25-34) for library-filled structs."""
        if self._malloced:
            for t in range(self.c.tries):
                _libc.free(C.cast(self.c.which_par[t], C.c_void_p))
            for p in (self.c.which_par, self.c.par_maxes, self.c.graph):
                _libc.free(C.cast(p, C.c_void_p))
            _libc.free(self.c.row_means)
            _libc.free(self.c.bases)
            self._malloced = False


def _take(ptr, count, ctype, dtype):
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).copy().astype(dtype)
    _libc.free(C.cast(ptr, C.c_void_p))
    return out


def gpu_init(prec="f32"):
    _lib.load(prec).gpu_init()


def gpu_cleanup(prec="f32"):
    _lib.load(prec).gpu_cleanup()


def precomp(points, k, tries=10, rots_before=6, rot_len_before=1, rots_after=1, rot_len_after=1, want_save=True):
    """precomp_gpu (include/algg.h; /root/reference/algg.h:7-11).  Returns (ids[n,k], sq_dists[n,k], Save|None).
    Draws its random rotations from libc random(): call srandom() first for a reproducible index."""
    points = np.ascontiguousarray(points)
    prec = _prec_of(points)
    lib = _lib.load(prec)
    n, d = points.shape
    save = Save(prec) if want_save else None
    dptr = C.c_void_p()
    ids = lib.precomp_gpu(n, k, d, points.ctypes.data, tries, rots_before, rot_len_before, rots_after, rot_len_after,
                          C.byref(save.c) if want_save else None, C.byref(dptr))
    if want_save:
        save._malloced = True
        save._points_ref = points  # the residency cache is keyed on this host pointer
    cft = C.c_float if prec == "f32" else C.c_double
    return (_take(ids, n * k, C.c_size_t, np.uint64).reshape(n, k),
            _take(dptr, n * k, cft, _ft(prec)).reshape(n, k), save)


def query(save, points, y, want_dists=True):
    """query_gpu (include/algg.h; /root/reference/algg.h:5-6).  `y is points` (same buffer) excludes self."""
    points = np.ascontiguousarray(points)
    prec = _prec_of(points)
    assert prec == save.prec
    lib = _lib.load(prec)
    if y is points or (isinstance(y, np.ndarray) and y.ctypes.data == points.ctypes.data):
        yy = points[: len(y)]
    else:
        yy = np.ascontiguousarray(y, dtype=points.dtype)
    ycnt, k = yy.shape[0], int(save.c.k)
    dptr = C.c_void_p()
    ids = lib.query_gpu(C.byref(save.c), points.ctypes.data, ycnt, yy.ctypes.data, C.byref(dptr) if want_dists else None)
    cft = C.c_float if prec == "f32" else C.c_double
    ids = _take(ids, ycnt * k, C.c_size_t, np.uint64).reshape(ycnt, k)
    if not want_dists:
        return ids, None
    return ids, _take(dptr, ycnt * k, cft, _ft(prec)).reshape(ycnt, k)


def synth_randnorm(count, prec="f32", reset=False):
    """annhip_synth_randnorm: `count` N(0,1) values from the caller's libc random() stream, exactly the values the
    reference's drivers generate (time_results.c:10-13, randNorm.c:9-21).  Seed with libc srandom() first."""
    lib = _lib.load(prec)
    if reset:
        lib.annhip_synth_reset()
    out = np.empty(int(count), dtype=_ft(prec))
    lib.annhip_synth_randnorm(int(count), out.ctypes.data)
    return out


class HostStream:
    """annhip_stream_*: numpy batches in, numpy results out, up to `lanes` batches in flight.

        hs = ix.host_stream(max_ycnt=10000, lanes=3)
        for ids, dists in hs.map(batches):      # results in submission order
            ...
    """

    def __init__(self, ix, max_ycnt, lanes=3):
        self.ix, self.lib, self.lanes = ix, ix.lib, lanes
        self.h = self.lib.annhip_stream_open(ix.h, max_ycnt, lanes)
        self.ft = _ft(ix.prec)
        self._pending = {}

    def submit(self, y, alias=False):
        y = np.ascontiguousarray(y, dtype=self.ft)
        t = self.lib.annhip_stream_submit(self.h, y.shape[0], y.ctypes.data, int(alias))
        if t >= 0:
            self._pending[t] = y.shape[0]
        return t

    def collect(self, ticket):
        n = self._pending.pop(ticket)
        ids = np.empty((n, self.ix.k), dtype=np.uint64)
        dists = np.empty((n, self.ix.k), dtype=self.ft)
        if self.lib.annhip_stream_collect(self.h, ticket, ids.ctypes.data, dists.ctypes.data) != 0:
            raise RuntimeError("unknown ticket %d" % ticket)
        return ids, dists

    def map(self, batches, alias=False):
        inflight = []
        for y in batches:
            if len(inflight) == self.lanes:
                yield self.collect(inflight.pop(0))
            t = self.submit(y, alias)
            assert t >= 0
            inflight.append(t)
        while inflight:
            yield self.collect(inflight.pop(0))

    def close(self):
        if self.h:
            self.lib.annhip_stream_close(self.h)
            self.h = None


def recall_ranks(points, y, guess, self_exclude=False):
    """annhip_recall_ranks: torch device tensors points [n,d], y [Q,d], guess int64 [Q,k] -> int64 ranks [Q,k]
    (number of points strictly closer than each guessed neighbour)."""
    import torch
    prec = "f32" if points.dtype == torch.float32 else "f64"
    lib = _lib.load(prec)
    assert points.is_cuda and y.is_cuda and guess.is_cuda and guess.dtype == torch.int64
    points, y, guess = points.contiguous(), y.contiguous(), guess.contiguous()
    ranks = torch.empty(guess.shape, dtype=torch.int64, device=guess.device)
    lib.annhip_recall_ranks(points.shape[0], points.shape[1], guess.shape[1], points.data_ptr(), y.shape[0], y.data_ptr(),
                            guess.data_ptr(), int(self_exclude), ranks.data_ptr())
    return ranks


def checksum(t, prec="f32"):
    """annhip_checksum_dev: 64-bit content checksum of a contiguous torch device tensor."""
    assert t.is_cuda and t.is_contiguous()
    return int(_lib.load(prec).annhip_checksum_dev(t.data_ptr(), t.numel() * t.element_size(), None))


def recall_summary(ranks, k):
    """The three numbers /root/reference/test_correctness.c:131-139 prints, from a rank tensor [Q,k]:
    average index score (mean rank excess per neighbour), probability correct (rank < k), max index score / k."""
    r = ranks.to("cpu").double()
    per_query = r.sum(dim=1).mean().item()
    return dict(avg_index_score=(per_query - k * (k - 1) / 2) / k,
                prob_correct=1.0 - (r >= k).double().mean().item(),
                max_index_score=r.max().item() / k)


class Index:
    """A device-resident index (include/ann_hip.h).  Tensors are torch CUDA(HIP) tensors; torch is only the
    allocator and stream provider here."""

    def __init__(self, prec, handle, keep=()):
        self.prec, self.h, self._keep = prec, handle, keep
        self.lib = _lib.load(prec)
        info = (C.c_size_t * 12)()
        self.lib.annhip_index_info(self.h, C.byref(info))
        (self.n, self.k, self.d, self.d_short, self.tries, self.L1, self.P1, self.Lc1, self.L2, self.P2, self.Lc2,
         self.sum_pm) = [int(v) for v in info]

    @staticmethod
    def _torch_ft(prec):
        import torch
        return torch.float32 if prec == "f32" else torch.float64

    @classmethod
    def from_save(cls, save, points, row_lo=0, row_hi=None):
        """points: numpy array (copied to HBM) or torch device tensor (borrowed) holding rows [row_lo,row_hi)."""
        lib = _lib.load(save.prec)
        row_hi = int(save.c.n) if row_hi is None else row_hi
        if isinstance(points, np.ndarray):
            pts = np.ascontiguousarray(points, dtype=_ft(save.prec))
            assert pts.shape == (row_hi - row_lo, save.c.d_long)
            h = lib.annhip_index_create(C.byref(save.c), pts.ctypes.data, 0, row_lo, row_hi)
            return cls(save.prec, h)
        assert points.is_cuda and points.is_contiguous() and points.dtype == cls._torch_ft(save.prec)
        assert tuple(points.shape) == (row_hi - row_lo, save.c.d_long)
        h = lib.annhip_index_create(C.byref(save.c), points.data_ptr(), 1, row_lo, row_hi)
        return cls(save.prec, h, keep=(points,))

    @classmethod
    def precomp(cls, points, k, tries=10, rots_before=6, rot_len_before=1, rots_after=1, rot_len_after=1,
                want_dists=False):
        """annhip_precomp_index: build the index on the device from a torch device tensor [n,d] (borrowed)."""
        import torch
        prec = "f32" if points.dtype == torch.float32 else "f64"
        assert points.is_cuda and points.is_contiguous()
        lib = _lib.load(prec)
        n, d = points.shape
        gd = torch.empty((n, k), dtype=points.dtype, device=points.device) if want_dists else None
        h = lib.annhip_precomp_index(n, k, d, points.data_ptr(), 1, tries, rots_before, rot_len_before, rots_after,
                                     rot_len_after, gd.data_ptr() if want_dists else None)
        ix = cls(prec, h, keep=(points,))
        ix.graph_dists = gd
        return ix

    def export(self):
        s = Save(self.prec)
        self.lib.annhip_index_export(self.h, C.byref(s.c))
        s._malloced = True
        return s

    def reshard(self, shard_points, row_lo, row_hi):
        """annhip_index_reshard: own rows [row_lo,row_hi) only, read from the torch device tensor shard_points."""
        assert shard_points.is_cuda and shard_points.is_contiguous() and tuple(shard_points.shape) == (row_hi - row_lo, self.d)
        self.lib.annhip_index_reshard(self.h, shard_points.data_ptr(), row_lo, row_hi)
        self._keep = (shard_points,)
        self.row_lo, self.row_hi = row_lo, row_hi

    def set_stream(self, stream_ptr):
        self.lib.annhip_index_set_stream(self.h, stream_ptr)

    def set_fixed(self, on=True):
        """annhip_index_set_fixed: opt-in non-parity query mode (own hash codes, every candidate slot; include/ann_hip.h)."""
        self.lib.annhip_index_set_fixed(self.h, int(bool(on)))

    def workspace(self):
        """annhip_workspace_create: scratch for one in-flight batch (pass to query(ws=..., stream=...))."""
        ws = self.lib.annhip_workspace_create(self.h)
        self._workspaces = getattr(self, "_workspaces", []) + [ws]
        return ws

    def query(self, y, alias=False, mode=0, out_ids=None, out_dists=None, ws=None, stream=None):
        """annhip_query / annhip_query_on: y torch tensor [Q,d] on the device -> (ids int64 [Q,k], sq dists [Q,k], n_exact).
        ws + stream (a torch.cuda.Stream): run this batch on its own workspace and stream so that it can overlap others."""
        import torch
        assert y.is_cuda and y.is_contiguous() and y.dtype == self._torch_ft(self.prec) and y.shape[1] == self.d
        Q = y.shape[0]
        ids = out_ids if out_ids is not None else torch.empty((Q, self.k), dtype=torch.int64, device=y.device)
        dists = out_dists if out_dists is not None else torch.empty((Q, self.k), dtype=y.dtype, device=y.device)
        if ws is None and stream is None:
            nex = self.lib.annhip_query(self.h, Q, y.data_ptr(), int(alias), mode, ids.data_ptr(), dists.data_ptr())
        else:
            nex = self.lib.annhip_query_on(self.h, ws, stream.cuda_stream if stream is not None else None, Q, y.data_ptr(),
                                           int(alias), mode, ids.data_ptr(), dists.data_ptr())
        return ids, dists, nex

    def host_stream(self, max_ycnt, lanes=3):
        """annhip_stream_open: pipeline for host-resident (numpy) batches; see HostStream."""
        return HostStream(self, max_ycnt, lanes)

    def checksum(self):
        """annhip_index_checksum: 64-bit checksum of everything a query reads except the point rows."""
        return int(self.lib.annhip_index_checksum(self.h))

    def profile(self, on=True):
        self.lib.annhip_profile(self.h, int(on))

    def stats(self, reset=False):
        out = (C.c_double * 8)()
        self.lib.annhip_stats(self.h, C.byref(out), int(reset))
        return dict(s1_launches=out[0], s1_ms=out[1], s1_rows=out[2], other_rows=out[3], exact_queries=out[4],
                    queries=out[5], tie_queries=out[6])  # tie_queries: flagged queries answered without the network (ann_tie.h)

    def stage_ms(self):
        out = (C.c_double * 6)()
        self.lib.annhip_stage_ms(self.h, C.byref(out))
        return dict(zip(("codes", "stage1", "finalize_fallback", "stage2_rows", "stage2_network", "widen"), [float(v) for v in out]))

    def close(self):
        if self.h:
            for ws in getattr(self, "_workspaces", []):
                self.lib.annhip_workspace_destroy(ws)
            self._workspaces = []
            self.lib.annhip_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
