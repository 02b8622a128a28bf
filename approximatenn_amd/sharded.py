"""Point-sharded query() across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-device; this is the multi-GPU design BASELINE.json's north_star asks for:

* rank g owns point rows [g*n/G, (g+1)*n/G) in HBM; bucket tables, graph, projection rows are replicated (ids only);
* every rank sees the whole query batch (results depend on the batch composition) and derives the
  identical candidate row per query, but gathers rows and computes distances only for ids it owns;
* exchange 0: each rank hashes one slice of the batch, the codes (4*T bytes per query) are all-gathered;
* exchange 1: all-gather of each rank's k+1 best distinct (dist,id) candidates per query  (G*(k+1)*8 B/query),
  merged by a device kernel (annhip_merge_candidates) to the global k+1 best; the same proof as on one GPU
  (annhip_stage1_finalize) decides which queries need the exact path; for those (rare) the full distance rows
  are min-all-reduced;
* exchange 2: min-reduce-scatter of the stage-2 distance rows (Lc2 values per query, 65 at k=10): each rank runs
  the reference's network on its slice of the queries only, and the final ids/distances are all-gathered.
  (Plain all-gather / min-all-reduce forms are kept as the fallback: fast=False, gloo, uneven batches.)

Every rank ends with the same ids/distances, bit-identical to the single-GPU / reference result.

`engine` is anything with the HipEngine methods below (tests drive the same orchestration with a CPU engine
built on the oracle under gloo); `dist` is torch.distributed or None for a single process.
"""
import ctypes as C

import torch


def _u32(t):
    """int32 tensor holding u32 bit patterns -> int64 values."""
    return t.to(torch.int64) & 0xFFFFFFFF


class HipEngine:
    """The staged C-ABI of include/ann_hip.h over torch device tensors."""

    def __init__(self, ix):
        self.ix, self.lib, self.h = ix, ix.lib, ix.h
        self.k, self.T, self.Lc1, self.Lc2 = ix.k, ix.tries, ix.Lc1, ix.Lc2
        self.ft = torch.float32 if ix.prec == "f32" else torch.float64

    def _e(self, shape, dtype, like):
        return torch.empty(shape, dtype=dtype, device=like.device)

    def codes(self, y):
        out = self._e((y.shape[0] * self.T,), torch.int32, y)
        self.lib.annhip_codes(self.h, y.shape[0], y.data_ptr(), out.data_ptr())
        return out

    def stage1_local(self, y, alias, codes):
        Q, K1 = y.shape[0], self.k + 1
        cd, ci, nv = self._e((Q, K1), self.ft, y), self._e((Q, K1), torch.int32, y), self._e((Q,), torch.int32, y)
        self.lib.annhip_stage1_local(self.h, Q, y.data_ptr(), int(alias), codes.data_ptr(), cd.data_ptr(), ci.data_ptr(),
                                     nv.data_ptr())
        return cd, ci, nv

    def merge(self, ndev, all_d, all_i):
        """all_d/all_i: [ndev, Q, k+1] gathered candidates -> the k+1 globally best per query."""
        Q, K1 = all_d.shape[1], all_d.shape[2]
        md, mi = self._e((Q, K1), self.ft, all_d), self._e((Q, K1), torch.int32, all_d)
        self.lib.annhip_merge_candidates(self.h, ndev, Q, all_d.data_ptr(), all_i.data_ptr(), md.data_ptr(), mi.data_ptr())
        return md, mi

    def finalize(self, cd, ci, nv):
        Q = cd.shape[0]
        top_i, top_d = self._e((Q, self.k), torch.int32, cd), self._e((Q, self.k), self.ft, cd)
        fl = self._e((Q,), torch.int32, cd)
        nf = self.lib.annhip_stage1_finalize(self.h, Q, cd.data_ptr(), ci.data_ptr(), nv.data_ptr(), top_i.data_ptr(),
                                             top_d.data_ptr(), fl.data_ptr())
        return top_i, top_d, fl[:nf]

    def stage1_rows(self, y, alias, codes, qidx):
        nq = qidx.shape[0]
        ids, dd = self._e((nq, self.Lc1), torch.int32, y), self._e((nq, self.Lc1), self.ft, y)
        self.lib.annhip_stage1_rows(self.h, y.shape[0], y.data_ptr(), int(alias), codes.data_ptr(), qidx.data_ptr(), nq,
                                    ids.data_ptr(), dd.data_ptr())
        return ids, dd

    def stage2_rows(self, y, alias, top_i, top_d):
        Q = y.shape[0]
        ids, dd = self._e((Q, self.Lc2), torch.int32, y), self._e((Q, self.Lc2), self.ft, y)
        self.lib.annhip_stage2_rows(self.h, Q, y.data_ptr(), int(alias), top_i.data_ptr(), top_d.data_ptr(), ids.data_ptr(),
                                    dd.data_ptr())
        return ids, dd

    def exact_select(self, stage, ids, dd, qidx, out_i, out_d):
        """network + rdups + network on each row; first k entries go to row qidx[i] (or i) of out_i/out_d."""
        self.lib.annhip_exact_select(self.h, stage, ids.shape[0], ids.data_ptr(), dd.data_ptr(),
                                     qidx.data_ptr() if qidx is not None else None, out_i.data_ptr(), out_d.data_ptr())


class ShardedQuery:
    """fast=True uses the leaner collectives (query-sharded hash codes, all-gather into one tensor + device merge
    kernel, reduce-scatter + sliced final network + all-gather for stage 2) whenever the batch divides evenly by
    the world size and the backend offers them; otherwise -- and always with fast=False -- the plain
    all-gather / all-reduce forms are used.  Both give identical results."""

    def __init__(self, ix_or_engine, dist=None, group=None, fast=True):
        self.eng = ix_or_engine if hasattr(ix_or_engine, "stage1_local") else HipEngine(ix_or_engine)
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1) else None
        self.group = group
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        backend = self.dist.get_backend(group) if self.dist else None
        self._stage_via_cpu = backend == "gloo"
        self.fast = bool(fast and self.dist and backend != "gloo" and hasattr(self.dist, "all_gather_into_tensor")
                         and hasattr(self.dist, "reduce_scatter_tensor"))
        self.last_exact = 0

    # -- collectives (RCCL on device tensors; gloo stages device tensors through the host) --
    def _all_gather(self, t):
        if not self.dist:
            return [t]
        src = t.cpu() if (self._stage_via_cpu and t.is_cuda) else t
        outs = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(outs, src.contiguous(), group=self.group)
        return [o.to(t.device) for o in outs] if src is not t else outs

    def _gather_stacked(self, t):
        """[world, *t.shape] through one all-gather into a single tensor."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def _all_min(self, t):
        if not self.dist:
            return t
        if self._stage_via_cpu and t.is_cuda:
            c = t.cpu()
            self.dist.all_reduce(c, op=self.dist.ReduceOp.MIN, group=self.group)
            t.copy_(c)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t

    def _merge(self, cd, ci):
        """Global k+1 smallest (dist,id) keys per query from every rank's k+1 (ids are disjoint across ranks)."""
        if not self.dist:
            return cd, ci
        if self.fast and hasattr(self.eng, "merge") and self.world <= 16:
            return self.eng.merge(self.world, self._gather_stacked(cd), self._gather_stacked(ci))
        K1 = cd.shape[1]
        if cd.dtype == torch.float32:  # one 64-bit key per candidate, one all-gather, one sort
            key = (cd.view(torch.int32).to(torch.int64) << 32) | _u32(ci)
            allk = torch.cat(self._all_gather(key), dim=1)
            best = torch.sort(allk, dim=1).values[:, :K1]
            md = (best >> 32).to(torch.int32).view(torch.float32)
            mi = (best & 0xFFFFFFFF).to(torch.int32)
            return md.contiguous(), mi.contiguous()
        alld = torch.cat(self._all_gather(cd), dim=1)
        alli = torch.cat(self._all_gather(ci), dim=1)
        o1 = torch.sort(_u32(alli), dim=1, stable=True).indices           # secondary key: id
        d1 = torch.gather(alld, 1, o1).view(torch.int64)                  # non-negative doubles order like int64
        o2 = torch.sort(d1, dim=1, stable=True).indices[:, :K1]           # primary key: distance bits
        sel = torch.gather(o1, 1, o2)
        return torch.gather(alld, 1, sel).contiguous(), torch.gather(alli, 1, sel).contiguous()

    def query(self, y, alias=False):
        """y: [Q,d] (identical on every rank).  Returns (ids int64 [Q,k], squared distances [Q,k])."""
        e, G, r = self.eng, self.world, self.rank
        Q = y.shape[0]
        even = self.fast and Q % G == 0 and Q >= G
        qs = Q // G if even else Q
        # hash codes: every rank needs all of them (Q2 scramble), each computes one slice of the batch
        if even:
            codes = self._gather_stacked(e.codes(y[r * qs:(r + 1) * qs].contiguous())).reshape(-1)
        else:
            codes = e.codes(y)
        cd, ci, nv = e.stage1_local(y, alias, codes)
        cd, ci = self._merge(cd, ci)
        top_i, top_d, flagged = e.finalize(cd, ci, nv)
        self.last_exact = int(flagged.shape[0])
        if flagged.shape[0]:  # identical list on every rank: it is a function of the merged candidates only
            ids, dd = e.stage1_rows(y, alias, codes, flagged)
            self._all_min(dd)
            e.exact_select(1, ids, dd, flagged, top_i, top_d)
        ids2, dd2 = e.stage2_rows(y, alias, top_i, top_d)
        if even:
            # each rank min-reduces and sorts only its slice of the queries, then the results are all-gathered
            mine = torch.empty((qs, dd2.shape[1]), dtype=dd2.dtype, device=dd2.device)
            self.dist.reduce_scatter_tensor(mine, dd2, op=self.dist.ReduceOp.MIN, group=self.group)
            loc_i = torch.empty((qs, top_i.shape[1]), dtype=top_i.dtype, device=top_i.device)
            loc_d = torch.empty((qs, top_d.shape[1]), dtype=top_d.dtype, device=top_d.device)
            e.exact_select(2, ids2[r * qs:(r + 1) * qs].contiguous(), mine, None, loc_i, loc_d)
            out_i = self._gather_stacked(loc_i).reshape(Q, -1)
            out_d = self._gather_stacked(loc_d).reshape(Q, -1)
        else:
            self._all_min(dd2)
            out_i = torch.empty_like(top_i)
            out_d = torch.empty_like(top_d)
            e.exact_select(2, ids2, dd2, None, out_i, out_d)
        return _u32(out_i), out_d
