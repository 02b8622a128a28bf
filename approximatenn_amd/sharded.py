"""Point-sharded query() across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-device; this is the multi-GPU design BASELINE.json's north_star asks for.

* rank g owns point rows [g*n/G, (g+1)*n/G) in HBM; bucket tables, graph, projection rows are replicated (ids only);
* every rank sees the whole query batch (results depend on the batch composition) and derives the identical
  candidate row per query, but gathers rows and computes distances only for the ids it owns;
* every query has an OWNER rank (contiguous slices of qs = ceil(Q/G) queries) that does its merges and its networks.

One step (5 collectives; the data path never touches the host):

  0. each rank hashes its own slice of the batch                     -> all-gather of the codes (4*T B/query)
  1. stage 1 on ALL queries over the rows this rank owns: k+1 best distinct packed (dist,id) keys per query
                                                                     -> all-to-all: keys go to the query's owner
     ((k+1)*8 B per query and rank; xGMI is point-to-point, so the 7 peers' slices travel on 7 different links)
  2. owner: merge the G lists + the single-GPU selection proof (annhip_sh_merge_finalize); rejected queries are
     flagged in place                                                -> all-gather of the top-k ids (4*k B/query)
  2b. flagged queries (exact ties between different ids, fewer than k candidates: ~0.5 per 10k at cfg3, i.e. a few in
     EVERY 80k-query step) take the exact path on the device: every rank derives the same ascending list, computes its
     part of their full distance rows                                -> MIN all-reduce of a FIXED [fcap][Lc1] buffer
     and runs the literal network; the flags disappear before stage 2.  No host decision is involved.
  3. every rank: distances of the neighbour-of-neighbour slots it owns (annhip_sh_stage2)
                                                                     -> all-to-all: partial rows go to the owner
  4. owner: min over the G partial rows, the reference's network on the stage-2 row (annhip_sh_final)
                                                                     -> all-gather of ids+distances (one packed buffer)

More than fcap (default 32) flagged queries in one step -- duplicate-heavy data, or k beyond the sorted prefix -- are
REPAIRED after the step: all ranks compute their part of the remaining rows, two MIN all-reduces, the literal network,
and the result rows are patched.  The count of those is the only thing the host reads back, and it does so in
collect(), not in the middle of the step.

Streams: with submit()/collect() two batches are in flight.  All stage-1 gathers run back to back on ONE
normal-priority HIP stream; everything else of a batch (hash, merges, stage 2, networks, the collectives' glue) runs on
the batch's own HIGH-priority stream, so the small latency-bound kernels and the exchanges of batch i slip in
underneath the HBM-bound gather of batch i+1 instead of queueing behind it (two gathers running concurrently only
slow each other down -- measured).  Collectives execute in ISSUE order on the communicator's single stream, so a batch
is issued in two halves and submit(i+1) issues [hash, codes all-gather, gather] of batch i+1 BEFORE the exchanges of
batch i (which all wait for gather i): otherwise gather(i+1) would start only when batch i is nearly done.

If the backend cannot do all_to_all_single the same steps run with an all-gather + local slice instead
(exchange="allgather": G times the traffic, identical results); the choice is agreed on by all ranks at start-up.

Every rank ends with the same ids/distances, bit-identical to the single-GPU / reference result.

`engine` is anything with the HipEngine methods below (tests drive the same orchestration with a CPU engine built on
the oracle under gloo); `dist` is torch.distributed (or a stand-in) or None for a single process.
"""
import contextlib
import os

import torch

ID_FLAG = 0xFFFFFFFE  # ann_query_kernels.h: ANN_ID_FLAG


def _u32(t):
    """int32 tensor holding u32 bit patterns -> int64 values."""
    return t.to(torch.int64) & 0xFFFFFFFF


class HipEngine:
    """The staged C-ABI of include/ann_hip.h over torch device tensors."""

    def __init__(self, ix):
        self.ix, self.lib, self.h = ix, ix.lib, ix.h
        self.k, self.T, self.Lc1, self.Lc2, self.P1 = ix.k, ix.tries, ix.Lc1, ix.Lc2, ix.P1
        self.ft = torch.float32 if ix.prec == "f32" else torch.float64
        self.key_words = self.lib.annhip_key_bytes() // 8
        self._stream = None  # raw hipStream_t of the lane being enqueued (None = the default stream)

    # -- streams: one high-priority HIP stream per in-flight batch, one shared stream for the gathers
    def new_stream(self, device, high_priority=False, reserve_cus=0):
        """reserve_cus > 0: a stream that leaves that many compute units to the other streams (the gathers' stream)."""
        if reserve_cus > 0:
            with torch.cuda.device(device):
                ptr = self.lib.annhip_stream_create_reserving(int(reserve_cus))
            if ptr:
                return torch.cuda.ExternalStream(ptr, device=device)
        return torch.cuda.Stream(device=device, priority=-1 if high_priority else 0)

    @contextlib.contextmanager
    def use(self, stream):
        """Launch on `stream` (None = the default stream): the C-ABI calls and torch ops alike.  Nestable."""
        prev = self._stream
        self._stream = stream.cuda_stream if stream is not None else None
        self.lib.annhip_index_set_stream(self.h, self._stream)
        try:
            with torch.cuda.stream(stream):
                yield
        finally:
            self._stream = prev
            self.lib.annhip_index_set_stream(self.h, prev)

    def new_event(self):
        return torch.cuda.Event()

    def set_gather_pieces(self, pieces):
        self.lib.annhip_index_set_gather_pieces(self.h, int(pieces))

    def set_gather_slots(self, waves_per_simd):
        self.lib.annhip_index_set_gather_slots(self.h, int(waves_per_simd))

    def empty(self, shape, dtype, like):
        return torch.empty(shape, dtype=dtype, device=like.device)

    # -- owner protocol
    def sh_codes(self, y, q_lo, q_hi, out):
        self.lib.annhip_sh_codes(self.h, self._stream, y.shape[0], y.data_ptr(), q_lo, q_hi, out.data_ptr())

    def sh_stage1(self, y, alias, codes, keys, nvalid, nown, stream=None):
        """stream: launch there instead of on the lane's stream (the shared gather stream)."""
        s = stream.cuda_stream if stream is not None else self._stream
        self.lib.annhip_sh_stage1(self.h, s, y.shape[0], y.data_ptr(), int(alias), codes.data_ptr(),
                                  keys.data_ptr(), nvalid.data_ptr(), nown.data_ptr())

    def sh_merge_finalize(self, G, Q, q_lo, qs, keys_in, nvalid, top_i, top_d):
        self.lib.annhip_sh_merge_finalize(self.h, self._stream, G, Q, q_lo, qs, keys_in.data_ptr(), nvalid.data_ptr(),
                                          top_i.data_ptr(), top_d.data_ptr())

    def sh_exact1_begin(self, y, alias, codes, top_all, fcap, flist, rows_i, rows_d):
        self.lib.annhip_sh_exact1_begin(self.h, self._stream, y.shape[0], y.data_ptr(), int(alias), codes.data_ptr(),
                                        top_all.data_ptr(), fcap, flist.data_ptr(), rows_i.data_ptr(), rows_d.data_ptr())

    def sh_exact1_end(self, Q, q_lo, qs, fcap, flist, rows_i, rows_d, top_all, top_d_all, top_i, top_d):
        self.lib.annhip_sh_exact1_end(self.h, self._stream, Q, q_lo, qs, fcap, flist.data_ptr(), rows_i.data_ptr(),
                                      rows_d.data_ptr(), top_all.data_ptr(), top_d_all.data_ptr(), top_i.data_ptr(),
                                      top_d.data_ptr())

    def sh_stage2(self, y, alias, top_all, dist_out, flagged):
        self.lib.annhip_sh_stage2(self.h, self._stream, y.shape[0], y.data_ptr(), int(alias), top_all.data_ptr(),
                                  dist_out.data_ptr(), flagged.data_ptr())

    def sh_final(self, G, Q, q_lo, qs, top_i, top_d, dist_in, out_i, out_d):
        self.lib.annhip_sh_final(self.h, self._stream, G, Q, q_lo, qs, top_i.data_ptr(), top_d.data_ptr(), dist_in.data_ptr(),
                                 out_i.data_ptr(), out_d.data_ptr())

    # -- exact path (repair of flagged queries); these run on the index's stream (set by use())
    def stage1_rows(self, y, alias, codes, qidx):
        nq = qidx.shape[0]
        ids, dd = self.empty((nq, self.Lc1), torch.int32, y), self.empty((nq, self.Lc1), self.ft, y)
        self.lib.annhip_stage1_rows(self.h, y.shape[0], y.data_ptr(), int(alias), codes.data_ptr(), qidx.data_ptr(), nq,
                                    ids.data_ptr(), dd.data_ptr())
        return ids, dd

    def stage2_rows_list(self, y, alias, qidx, top_i, top_d):
        nq = qidx.shape[0]
        ids, dd = self.empty((nq, self.Lc2), torch.int32, y), self.empty((nq, self.Lc2), self.ft, y)
        self.lib.annhip_stage2_rows_list(self.h, y.shape[0], y.data_ptr(), int(alias), qidx.data_ptr(), nq, top_i.data_ptr(),
                                         top_d.data_ptr(), ids.data_ptr(), dd.data_ptr())
        return ids, dd

    def exact_select(self, stage, ids, dd, qidx, out_i, out_d):
        """network + rdups + network on each row; first k entries go to row qidx[i] (or i) of out_i/out_d."""
        self.lib.annhip_exact_select(self.h, stage, ids.shape[0], ids.data_ptr(), dd.data_ptr(),
                                     qidx.data_ptr() if qidx is not None else None, out_i.data_ptr(), out_d.data_ptr())


class _Lane:
    """Buffers + stream of one in-flight batch.  Everything is allocated once per batch size: no per-step tensors."""

    def __init__(self, stream):
        self.stream, self.shape, self.busy, self.event, self.next, self.seq = stream, None, False, None, 6, 0
        self.ev_in = self.ev_out = None

    def ensure(self, eng, y, G, qs, fcap):
        Q = y.shape[0]
        key = (Q, G, qs, fcap, y.device, y.dtype)
        if self.shape == key:
            return
        k, T, K1w, W2 = eng.k, eng.T, (eng.k + 1) * eng.key_words, eng.Lc2 - eng.k
        Qp, i32, i64, ft = qs * G, torch.int32, torch.int64, eng.ft
        e = lambda shape, dt: eng.empty(shape, dt, y)  # noqa: E731
        self.codes_slice, self.codes_all = e((qs * T,), i32), e((Qp * T,), i32)
        self.keys, self.keys_in = e((Qp, K1w), i64), e((Qp, K1w), i64)
        self.nvalid, self.nown = e((Q,), i32), e((Q,), i32)
        self.top_i, self.top_d, self.top_all = e((qs, k), i32), e((qs, k), ft), e((Qp, k), i32)
        self.s2, self.s2_in = e((Qp, W2), ft), e((Qp, W2), ft)
        self.flagged = e((Q + 1,), i32)
        self.top_d_all = e((Qp, k), ft)
        self.flist, self.xrows_i, self.xrows_d = e((2 + fcap,), i32), e((fcap, eng.Lc1), i32), e((fcap, eng.Lc1), ft)
        # the step's only host read-back: {flagged left to the host, flagged in total}, copied to pinned memory by the
        # batch's own stream so that collect() issues no GPU work on any other stream
        self.head_dev = e((2,), i32)
        self.head_host = torch.empty((2,), dtype=i32, pin_memory=True) if y.is_cuda else torch.empty((2,), dtype=i32)
        self.ids64 = e((Qp, k), i64)
        es = 4 if ft == torch.float32 else 8
        self.nb_d, self.nb_i = qs * k * es, qs * k * 4
        per = (self.nb_d + self.nb_i + 15) // 16 * 16             # this rank's results: distances, then ids, padded
        self.pack, self.pack_all = e((per,), torch.uint8), e((G, per), torch.uint8)
        self.out_d_slice = self.pack[: self.nb_d].view(ft).view(qs, k)
        self.out_i_slice = self.pack[self.nb_d: self.nb_d + self.nb_i].view(i32).view(qs, k)
        self.out_i, self.out_d = e((Qp, k), i32), e((Qp, k), ft)
        self.shape = key


class ShardedQuery:
    """query(y) = collect(submit(y)).  With submit()/collect() up to `lanes` batches are in flight.

    exchange: "alltoall" (default), "allgather" (fallback: every rank receives everything and keeps its slice), or None =
    ANN_SHARD_EXCHANGE from the environment, else probe all_to_all_single at start-up and agree across ranks."""

    def __init__(self, ix_or_engine, dist=None, group=None, exchange=None, lanes=2, exact_all=None, fcap=32,
                 reserve_cus=None):
        self.eng = ix_or_engine if hasattr(ix_or_engine, "sh_stage1") else HipEngine(ix_or_engine)
        # ANN_SHARD_FORCE_DIST=1: keep the collectives even with ONE rank (rehearses the RCCL calls on a single-GPU box)
        self.dist = dist if (dist is not None and dist.is_initialized() and
                             (dist.get_world_size(group) > 1 or os.environ.get("ANN_SHARD_FORCE_DIST") == "1")) else None
        self.group = group
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        backend = self.dist.get_backend(group) if self.dist else None
        self._via_cpu = backend == "gloo"   # gloo moves host tensors: device tensors are staged through the host
        self.exact_all = bool(os.environ.get("ANN_HIP_EXACT")) if exact_all is None else exact_all
        if self.eng.k > self.eng.P1:
            self.exact_all = True           # the selection cannot be proven when k exceeds the sorted prefix (Q1)
        self.exchange = self._agree_exchange(exchange or os.environ.get("ANN_SHARD_EXCHANGE"))
        self.fcap = max(1, int(fcap))
        # compute units the gathers leave to everything else (small kernels, RCCL); ANN_SHARD_RESERVE_CUS overrides
        self.reserve_cus = int(os.environ.get("ANN_SHARD_RESERVE_CUS", 0 if reserve_cus is None else reserve_cus))
        self._gather_streams = {}           # CUs left free -> the gathers' stream
        self._lanes = [_Lane(None) for _ in range(max(1, lanes))]
        self.depth = len(self._lanes)       # lanes in use (autotune() may lower it)
        self.split = 0 if os.environ.get("ANN_SHARD_NO_SPLIT") == "1" else 1   # issue order of the stages (see _stage())
        self.tuned, self.pieces, self.slots = None, 1, 0
        self._next, self._tickets = 0, {}
        self.last_exact = 0

    # ------------------------------------------------------------------ collectives
    def _agree_exchange(self, want):
        if not self.dist:
            return "none"
        if want in ("alltoall", "allgather"):
            return want
        ok = 1
        dev = "cpu" if (self._via_cpu or not torch.cuda.is_available()) else torch.device("cuda", torch.cuda.current_device())
        a = torch.arange(self.world, dtype=torch.int64, device=dev) + 100 * self.rank
        b = torch.empty_like(a)
        try:  # a tiny all-to-all on the backend's native tensors.  ONLY "this backend has no such op" -- raised before
            # anything is enqueued, on every rank alike -- selects the fallback; any other failure (out of memory, a
            # communicator error) is rank-local: the peers are inside the all-to-all, so this rank must die, not go on
            # to a different collective.
            self.dist.all_to_all_single(b, a, group=self.group)
        except NotImplementedError:
            ok = 0
        except RuntimeError as err:
            msg = str(err).lower()
            if not ("not supported" in msg or "unsupported" in msg or "does not support" in msg or "not implemented" in msg):
                raise
            ok = 0
        if ok and b.cpu().tolist() != [100 * g + self.rank for g in range(self.world)]:
            raise RuntimeError("all_to_all_single delivered wrong data on rank %d" % self.rank)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)   # every rank takes the same branch
        return "alltoall" if int(flag.cpu()[0]) == 1 else "allgather"

    def _host(self, t):
        return t.cpu() if (self._via_cpu and t.is_cuda) else t

    def _gather_cat(self, out, inp):
        """out[g*rows:(g+1)*rows] = rank g's inp (concatenation along dim 0; out is contiguous)."""
        if not self.dist:
            out.copy_(inp.reshape(out.shape))
            return
        if self._via_cpu and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, inp.cpu().contiguous(), group=self.group)
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def _to_owner(self, out, inp):
        """inp = [G*qs, w]: rows [g*qs,(g+1)*qs) are for owner g.  out = [G*qs, w]: rows [g*qs,(g+1)*qs) = what rank g
        computed for MY queries."""
        if not self.dist:
            out.copy_(inp)
            return
        G, r = self.world, self.rank
        qs = inp.shape[0] // G
        src = self._host(inp)
        dst = torch.empty(out.shape, dtype=out.dtype) if src is not inp else out
        if self.exchange == "alltoall":
            self.dist.all_to_all_single(dst, src, group=self.group)
        else:
            everything = torch.empty((G * inp.shape[0],) + tuple(inp.shape[1:]), dtype=inp.dtype, device=src.device)
            self.dist.all_gather_into_tensor(everything, src, group=self.group)
            dst.copy_(everything.view(G, G, qs, -1)[:, r].reshape(dst.shape))
        if dst is not out:
            out.copy_(dst)

    def _all_min(self, t):
        if not self.dist:
            return t
        if self._via_cpu and t.is_cuda:
            c = t.cpu()
            self.dist.all_reduce(c, op=self.dist.ReduceOp.MIN, group=self.group)
            t.copy_(c)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t

    # ------------------------------------------------------------------ one step
    # A batch goes through six stages, each ending in exactly ONE collective:
    #   0  hash of the owned query slice              -> all-gather of the codes
    #   1  stage-1 gather over the owned rows         -> all-to-all of the candidate keys
    #   2  owner: merge + proof                       -> all-gather of the top-k ids
    #   3  flagged queries: exact rows                -> MIN all-reduce (fixed size)
    #   4  their network; stage-2 distances           -> all-to-all of the partial rows
    #   5  owner: min + network                       -> all-gather of the results
    # All collectives of a process group execute in ISSUE order on the communicator's one stream, so the order in which
    # the host issues the stages of the in-flight batches decides what can overlap:
    #   split = 0   a whole batch at once (stages 0..5 of batch i, then batch i+1)
    #   split = 1   two halves: submit(i+1) issues stages 0-1 of batch i+1 BEFORE stages 2-5 of batch i -- otherwise the codes
    #               all-gather of batch i+1 sits behind batch i's exchanges (which all wait for gather i) and gather i+1
    #               cannot start before batch i is nearly done
    #   split = 2   STAGE-MAJOR (software pipeline): every submit() advances every in-flight batch by ONE stage --
    #               stage 0 of the new batch first, then stages 5, 4, 3, 2 of the older ones, then stage 1 (gather + key
    #               exchange) of the batch hashed one submit earlier, so that a gather never waits for a collective issued
    #               in the same step.  Needs >= 7 lanes (a batch is in flight for six submits).  On ONE GPU with RCCL
    #               kernels in the loop (tools/emulate_rank.py --rccl) it measures like split = 1 (1.65-1.75 ms per 8-way
    #               rank step): there a collective that becomes ready in the middle of a gather completes only when the
    #               gather drains -- 1.0-1.2 ms of a 1.25 ms gather, also when the gather runs as a persistent grid that
    #               leaves half of the wave slots free (annhip_index_set_gather_slots), so it is not a placement problem --
    #               and the rest of its batch's chain (~0.35 ms) runs before the next gather can start.  Whether RCCL
    #               behaves like that between real ranks only a multi-GPU run can tell: the order is one of the settings
    #               autotune() measures in place.
    def _stage(self, L, st):
        e, G, Q, qs, q_lo, y, alias = self.eng, self.world, L.Q, L.qs, L.q_lo, L.y, L.alias
        with e.use(L.stream):
            if st == 0:
                L.ensure(e, y, G, qs, self.fcap)
                e.sh_codes(y, q_lo, q_lo + qs, L.codes_slice)
                self._gather_cat(L.codes_all, L.codes_slice)
            elif st == 1:
                if self.exact_all:
                    L.top_all.fill_(ID_FLAG - (1 << 32))                 # every query takes the exact path
                    L.top_i.fill_(ID_FLAG - (1 << 32))
                else:
                    gs = self._gather_streams[self.reserve_cus] if L.stream is not None else None
                    if gs is not None:                                   # gathers of all batches: back to back
                        if L.ev_in is None:
                            L.ev_in, L.ev_out, L.event = e.new_event(), e.new_event(), e.new_event()
                        L.ev_in.record(L.stream)
                        gs.wait_event(L.ev_in)
                    e.sh_stage1(y, alias, L.codes_all, L.keys, L.nvalid, L.nown, stream=gs)
                    if gs is not None:
                        L.ev_out.record(gs)
                        L.stream.wait_event(L.ev_out)
                    if self.split != 1:        # (two-half order: the key exchange belongs to the second half)
                        self._to_owner(L.keys_in, L.keys)
            elif st == 2:
                if not self.exact_all:
                    if self.split == 1:
                        self._to_owner(L.keys_in, L.keys)
                    e.sh_merge_finalize(G, Q, q_lo, qs, L.keys_in, L.nvalid, L.top_i, L.top_d)
                    self._gather_cat(L.top_all, L.top_i)
            elif st == 3:      # flagged queries: exact stage 1 on the device, one fixed-size MIN all-reduce in the middle
                e.sh_exact1_begin(y, alias, L.codes_all, L.top_all, self.fcap, L.flist, L.xrows_i, L.xrows_d)
                self._all_min(L.xrows_d)
            elif st == 4:
                e.sh_exact1_end(Q, q_lo, qs, self.fcap, L.flist, L.xrows_i, L.xrows_d, L.top_all, L.top_d_all, L.top_i, L.top_d)
                e.sh_stage2(y, alias, L.top_all, L.s2, L.flagged)
                self._to_owner(L.s2_in, L.s2)
            elif st == 5:
                e.sh_final(G, Q, q_lo, qs, L.top_i, L.top_d, L.s2_in, L.out_i_slice, L.out_d_slice)
                self._gather_cat(L.pack_all, L.pack.view(1, -1))
                # (the packed rows are unpacked in collect(): straight into the 64-bit ids / the returned distances)
                L.head_host[0:1].copy_(L.flagged[0:1], non_blocking=True)
                L.head_host[1:2].copy_(L.flist[1:2], non_blocking=True)
                if L.stream is not None:
                    if L.event is None:
                        L.event = e.new_event()
                    L.event.record(L.stream)
        L.next = st + 1

    def _advance(self, L, upto):
        while L.next <= upto:
            self._stage(L, L.next)

    def _tick(self, new=None):
        """Stage-major order: one stage for every batch in flight (see above)."""
        if new is not None:
            self._stage(new, 0)
        older = sorted((P for P in self._lanes if P.busy and P is not new and P.next <= 5), key=lambda P: P.seq)
        for P in older:                       # oldest first: stages 5, 4, 3, 2 ...
            if P.next >= 2:
                self._stage(P, P.next)
        for P in older:                       # ... and last the gather + key exchange of the batch hashed a step ago
            if P.next == 1:
                self._stage(P, 1)

    def submit(self, y, alias=False):
        """Enqueue one batch (y: [Q,d], identical on every rank).  Returns a ticket for collect()."""
        e, G, r = self.eng, self.world, self.rank
        L = self._lanes[self._next % self.depth]
        if L.busy:
            raise RuntimeError("every lane is in flight: collect() the oldest ticket first")
        if L.stream is None and y.is_cuda:
            L.stream = e.new_stream(y.device, high_priority=True)
        if y.is_cuda and self.reserve_cus not in self._gather_streams:
            self._gather_streams[self.reserve_cus] = e.new_stream(y.device, reserve_cus=self.reserve_cus)
        Q = y.shape[0]
        L.y, L.alias, L.Q, L.qs = y, alias, Q, (Q + G - 1) // G
        L.q_lo = r * L.qs
        L.next, L.seq = 0, self._next
        if L.stream is not None:
            L.stream.wait_stream(torch.cuda.current_stream(y.device))   # y was produced on the caller's stream
        L.busy = True
        if self.split >= 2:
            self._tick(L)
        else:
            self._advance(L, 1)
            if not self.split:                                           # the whole batch at once
                self._advance(L, 5)
            for P in self._lanes:                                        # the batch submitted before this one
                if P is not L and P.busy and P.next <= 5:
                    self._advance(P, 5)
        t = self._next
        self._tickets[t] = L
        self._next += 1
        return t

    def collect(self, ticket):
        """Wait for the batch, repair flagged queries if there are any, return (ids int64 [Q,k], sq. distances [Q,k])."""
        e = self.eng
        L = self._tickets.pop(ticket)
        while L.next <= 5:                    # no later batch was submitted (or fewer than the pipeline is deep): flush
            if self.split >= 2:
                self._tick(None)
            else:
                self._advance(L, 5)
        if L.event is not None:
            L.event.synchronize()
        nf = int(L.head_host[0])              # flagged queries the device-driven exact path had no room for; the same
        self.last_exact = int(L.head_host[1])  # on every rank.  [1] = all flagged queries of the batch
        Q, G, qs = L.Q, self.world, L.qs
        # fresh result tensors (the lane is re-used), allocated OUTSIDE the lane's stream context: they belong to the
        # caller's stream, which waits for the lane below -- memory the caller frees can then never be handed out again
        # while this lane's stream still writes it
        ids = torch.empty((G * qs, e.k), dtype=torch.int64, device=L.pack_all.device)
        dd = torch.empty((G * qs, e.k), dtype=e.ft, device=L.pack_all.device)
        if L.stream is not None:
            L.stream.wait_stream(torch.cuda.current_stream(L.y.device))   # ... and not before the caller's earlier work
        with e.use(L.stream):
            packed_d = L.pack_all[:, : L.nb_d].view(e.ft)                               # [G, qs*k]
            packed_i = L.pack_all[:, L.nb_d: L.nb_d + L.nb_i].view(torch.int32)
            if nf:                                                   # rare: the repair patches rows of out_i / out_d
                L.out_d.view(G, qs * e.k).copy_(packed_d)
                L.out_i.view(G, qs * e.k).copy_(packed_i)
                self._repair(L, nf)
                ids.copy_(L.out_i)
                dd.copy_(L.out_d)
            else:
                ids.view(G, qs * e.k).copy_(packed_i)
                dd.view(G, qs * e.k).copy_(packed_d)
            ids.bitwise_and_(0xFFFFFFFF)                             # u32 bit patterns -> the ABI's 64-bit ids
            ids, dd = ids[:Q], dd[:Q]
            if L.stream is not None:
                torch.cuda.current_stream(L.y.device).wait_stream(L.stream)
        L.busy, L.y = False, None
        return ids, dd

    def _repair(self, L, nf):
        """Exact path for the flagged queries, on every rank alike: full distance rows of the first Lc1 slots, MIN
        all-reduce, the literal network; then their stage-2 rows the same way; the result rows are patched."""
        e, y, alias = self.eng, L.y, L.alias
        fl = torch.sort(_u32(L.flagged[1:1 + nf])).values.to(torch.int32)   # appended by atomics: sort => same order everywhere
        top_d_all = L.top_d_all
        ids1, dd1 = e.stage1_rows(y, alias, L.codes_all, fl)
        self._all_min(dd1)
        e.exact_select(1, ids1, dd1, fl, L.top_all, top_d_all)
        ids2, dd2 = e.stage2_rows_list(y, alias, fl, L.top_all, top_d_all)
        self._all_min(dd2)
        e.exact_select(2, ids2, dd2, fl, L.out_i, L.out_d)

    def query(self, y, alias=False):
        """y: [Q,d] (identical on every rank).  Returns (ids int64 [Q,k], squared distances [Q,k])."""
        return self.collect(self.submit(y, alias))

    def close(self):
        """Release the CU-masked gather streams (annhip_stream_create_reserving); the object is unusable afterwards."""
        if any(L.busy for L in self._lanes):
            raise RuntimeError("close() with batches in flight")
        for cus, st in list(self._gather_streams.items()):
            if cus > 0 and isinstance(st, torch.cuda.ExternalStream):
                st.synchronize()
                self.eng.lib.annhip_stream_destroy(st.cuda_stream)
        self._gather_streams = {}
        self._lanes = []

    # ------------------------------------------------------------------ scheduling knobs, measured in place
    # How the small kernels and the RCCL kernels of one batch get compute units beside the saturating gather of the next
    # one depends on the machine state (ranks, RCCL channel count, batch size): nothing of it changes a result bit, so it
    # is measured, not guessed.
    # Candidates: (lanes in use, two-half issue, CUs the gathers leave free, launches per gather).
    # Candidates: (lanes in use, two-half issue, CUs the gathers leave free, launches per gather).  The first one is the
    # PINNED default: what every job runs unless a measurement inside its time budget finds something faster.  The
    # CU-masked candidates (hipExtStreamCreateWithCUMask) are opt-in: they only ever won the two-lane case.
    PINNED = (3, True, 0, 1)
    STAGEWISE = (7, 2, 0, 1)   # stage-major issue order, hash one step ahead (needs 7 lanes)
    TUNE_CANDIDATES = (PINNED, STAGEWISE, (3, False, 0, 1), (2, False, 0, 1), (3, True, 0, 4), (3, True, 0, 1, 3), (1, False, 0, 1))
    TUNE_CANDIDATES_MASKED = ((3, True, 8, 1), (3, False, 8, 1), (2, False, 8, 1))

    def pump(self, ys, alias=False):
        """Pipelined loop over the batches ys with `depth` batches in flight; returns the list of results."""
        out, pend = [], []
        for y in ys:
            pend.append(self.submit(y, alias))
            if len(pend) >= self.depth:
                out.append(self.collect(pend.pop(0)))
        while pend:
            out.append(self.collect(pend.pop(0)))
        return out

    def configure(self, depth, split, reserve_cus=0, pieces=1, slots=0):
        """slots > 0: the stage-1 gather as a persistent grid holding that many waves per SIMD (annhip_index_set_gather_slots)."""
        if any(L.busy for L in self._lanes):
            raise RuntimeError("configure() with batches in flight")
        self.depth, self.split, self.reserve_cus = max(1, min(int(depth), len(self._lanes))), int(split), int(reserve_cus)
        self.pieces, self.slots = max(1, int(pieces)), max(0, int(slots))
        self.eng.set_gather_pieces(self.pieces)
        if hasattr(self.eng, "set_gather_slots"):
            self.eng.set_gather_slots(self.slots)

    def _agreed_max(self, value, like):
        """MAX over the ranks of a host float (one tiny all-reduce; every rank gets the same number)."""
        if not self.dist:
            return float(value)
        dev = "cpu" if (self._via_cpu or not like.is_cuda) else like.device
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.cpu()[0])

    def autotune(self, y, alias=False, batches=8, candidates=None, budget_s=20.0):
        """Start from the PINNED schedule; then, while the wall-clock budget lasts, time `batches` pipelined batches of y
        under each further candidate (after 3 untimed ones) and keep the fastest.  Every decision -- a candidate's time, and
        whether there is budget left for the next one -- is the MAX over the ranks (all-reduce), so every rank times
        the same candidates and ends with the same setting, which the collectives' issue order requires; a candidate
        that cannot be set up on some rank (CU-masked stream refused) counts as infinitely slow on all of them.
        Returns {"depth", "split", "reserve_cus", "pieces", "ms_per_batch", "table", ...}; also kept in self.tuned."""
        import time
        cands = [c for c in (candidates or self.TUNE_CANDIDATES) if c[0] <= len(self._lanes)]
        if not cands:
            cands = [(len(self._lanes), True, 0, 1)]
        sync = (lambda: torch.cuda.synchronize(y.device)) if y.is_cuda else (lambda: None)
        t_start = time.perf_counter()
        self.configure(len(self._lanes), True, 0, 1)
        self.pump([y] * (len(self._lanes) + 1), alias)        # every lane's buffers and streams exist before anything is timed
        table, stopped = [], None
        for i, c in enumerate(cands):
            if i > 0 and self._agreed_max(time.perf_counter() - t_start, y) > budget_s:
                stopped = "budget of %.0f s used up after %d of %d candidates" % (budget_s, i, len(cands))
                break
            ok = 1.0
            if c[2] > 0 and y.is_cuda and c[2] not in self._gather_streams:
                st = self.eng.new_stream(y.device, reserve_cus=c[2])
                if isinstance(st, torch.cuda.ExternalStream):
                    self._gather_streams[c[2]] = st
                else:
                    ok = 0.0
            if self._agreed_max(1.0 - ok, y) > 0:             # some rank cannot run this candidate: nobody does
                table.append(float("inf"))
                continue
            self.configure(*c)
            self.pump([y] * 3, alias)
            sync()
            if self.dist:
                self.dist.barrier(group=self.group)
            t0 = time.perf_counter()
            self.pump([y] * batches, alias)
            sync()
            table.append(self._agreed_max((time.perf_counter() - t0) * 1e3 / batches, y))
        best = min(range(len(table)), key=lambda j: table[j])
        self.configure(*cands[best])
        self.tuned = {"depth": self.depth, "split": self.split, "reserve_cus": self.reserve_cus, "pieces": self.pieces,
                      "slots": self.slots, "ms_per_batch": round(table[best], 4), "pinned_ms_per_batch": round(table[0], 4),
                      "tune_s": round(time.perf_counter() - t_start, 2),
                      "table": [{"depth": c[0], "split": c[1], "reserve_cus": c[2], "pieces": c[3],
                                 "slots": c[4] if len(c) > 4 else 0,
                                 "ms": (round(v, 4) if v != float("inf") else None)} for c, v in zip(cands, table)]}
        if stopped:
            self.tuned["stopped"] = stopped
        return self.tuned


class ReplicaQuery:
    """Query-sharded query(): every rank holds ALL rows and the whole index and answers a contiguous slice of each batch.

    SURVEY 8(e) names this split as the alternative to row sharding; 20 GB of cfg4's points fit every 288-GB GPU.  Results
    depend on the whole batch (query x reads hash codes of other queries, Q2), so one exchange remains: every rank hashes
    its slice, ONE all-gather gives everybody the codes of all queries (4*T bytes per query), then annhip_query_slice runs
    the single-GPU path -- the whole-index gather at its full roofline fraction, the fused stage 2, the device-driven exact
    path -- on the slice.  gather=True all-gathers the results (k*(8+s) bytes per query) so that every rank returns the
    whole batch; otherwise a rank returns its own slice.  Bit-identical to the single-GPU answer.
    Up to `lanes` batches are in flight (submit / collect / pump), each on its own stream and workspace."""

    def __init__(self, ix, dist=None, group=None, lanes=2, gather=True):
        self.ix, self.lib, self.group, self.gather = ix, ix.lib, group, gather
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self._via_cpu = bool(self.dist) and self.dist.get_backend(group) == "gloo"
        self.ft = torch.float32 if ix.prec == "f32" else torch.float64
        self.depth = max(1, int(lanes))
        self._lanes = [dict(stream=None, ws=None, busy=False, shape=None) for _ in range(self.depth)]
        self._next, self._tickets = 0, {}

    def _all_gather(self, out, inp):
        if not self.dist:
            out.copy_(inp.reshape(out.shape))
        elif self._via_cpu and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, inp.cpu().contiguous(), group=self.group)
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def submit(self, y, alias=False):
        assert y.is_cuda and y.is_contiguous() and y.dtype == self.ft and y.shape[1] == self.ix.d
        if alias and self.world > 1:
            raise ValueError("an aliased batch (y == points) cannot be query-sharded")
        L = self._lanes[self._next % self.depth]
        if L["busy"]:
            raise RuntimeError("every lane is in flight: collect() the oldest ticket first")
        ix, G, r, k, T = self.ix, self.world, self.rank, self.ix.k, self.ix.tries
        Q = y.shape[0]
        qs = (Q + G - 1) // G
        q_lo = min(Q, r * qs)
        nq = min(qs, Q - q_lo)
        if L["stream"] is None:
            L["stream"], L["ws"] = torch.cuda.Stream(device=y.device), ix.workspace()
        if L["shape"] != (Q, G):
            e = lambda shape, dt: torch.empty(shape, dtype=dt, device=y.device)  # noqa: E731
            es = 4 if self.ft == torch.float32 else 8
            L["codes_slice"], L["codes_all"] = e((qs * T,), torch.int32), e((qs * G * T,), torch.int32)
            L["nb_i"], L["nb_d"] = qs * k * 8, qs * k * es
            per = (L["nb_i"] + L["nb_d"] + 15) // 16 * 16          # this rank's results: ids, then distances, padded
            L["pack"], L["pack_all"] = e((per,), torch.uint8), e((G, per), torch.uint8)
            L["ids"] = L["pack"][: L["nb_i"]].view(torch.int64).view(qs, k)
            L["dd"] = L["pack"][L["nb_i"]: L["nb_i"] + L["nb_d"]].view(self.ft).view(qs, k)
            L["shape"] = (Q, G)
        L.update(y=y, Q=Q, qs=qs, nq=nq)
        st = L["stream"]
        st.wait_stream(torch.cuda.current_stream(y.device))
        with torch.cuda.stream(st):
            self.lib.annhip_sh_codes(ix.h, st.cuda_stream, Q, y.data_ptr(), q_lo, q_lo + qs, L["codes_slice"].data_ptr())
            self._all_gather(L["codes_all"], L["codes_slice"])
            if nq:
                self.lib.annhip_query_slice(ix.h, L["ws"], st.cuda_stream, Q, q_lo, nq, y[q_lo:].data_ptr(),
                                            L["codes_all"].data_ptr(), int(alias), L["ids"].data_ptr(), L["dd"].data_ptr())
            if self.gather:
                self._all_gather(L["pack_all"], L["pack"].view(1, -1))
        L["busy"] = True
        t = self._next
        self._tickets[t] = L
        self._next += 1
        return t

    def collect(self, ticket):
        """(ids int64 [Q,k], squared distances [Q,k]) of the whole batch (gather=True) or of this rank's slice."""
        L = self._tickets.pop(ticket)
        k, Q, qs, G = self.ix.k, L["Q"], L["qs"], self.world
        cur = torch.cuda.current_stream(L["y"].device)
        if self.gather:
            ids = torch.empty((G * qs, k), dtype=torch.int64, device=L["y"].device)
            dd = torch.empty((G * qs, k), dtype=self.ft, device=L["y"].device)
            L["stream"].wait_stream(cur)
            with torch.cuda.stream(L["stream"]):
                ids.view(G, qs * k).copy_(L["pack_all"][:, : L["nb_i"]].view(torch.int64))
                dd.view(G, qs * k).copy_(L["pack_all"][:, L["nb_i"]: L["nb_i"] + L["nb_d"]].view(self.ft))
            ids, dd = ids[:Q], dd[:Q]
        else:
            ids, dd = torch.empty((L["nq"], k), dtype=torch.int64, device=L["y"].device), \
                torch.empty((L["nq"], k), dtype=self.ft, device=L["y"].device)
            L["stream"].wait_stream(cur)
            with torch.cuda.stream(L["stream"]):
                ids.copy_(L["ids"][: L["nq"]])
                dd.copy_(L["dd"][: L["nq"]])
        cur.wait_stream(L["stream"])
        L["busy"], L["y"] = False, None
        return ids, dd

    def query(self, y, alias=False):
        return self.collect(self.submit(y, alias))

    def pump(self, ys, alias=False):
        out, pend = [], []
        for y in ys:
            pend.append(self.submit(y, alias))
            if len(pend) >= self.depth:
                out.append(self.collect(pend.pop(0)))
        while pend:
            out.append(self.collect(pend.pop(0)))
        return out


def same_everywhere(dist, values, what="values", group=None, device="cpu"):
    """All ranks must hold the same 64-bit `values` (checksums of the index, of a batch ...): MIN and MAX all-reduce;
    raises SystemExit(3) ON EVERY RANK when they differ -- a rank that went on with other inputs would return wrong
    answers silently or leave its peers in mismatched collectives."""
    v = [int(x) & 0xFFFFFFFFFFFFFFFF for x in values]
    if dist is None or not dist.is_initialized():
        return v
    # split into 32-bit halves: exact in int64 whatever the backend's integer reduction does with the sign bit
    parts = [h for x in v for h in (x >> 32, x & 0xFFFFFFFF)]
    lo = torch.tensor(parts, dtype=torch.int64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo.cpu(), hi.cpu()):
        import sys
        sys.stderr.write("approximatenn_amd: rank %d: %s differ between the ranks (mine: %s)\n"
                         % (dist.get_rank(group), what, ["%016x" % x for x in v]))
        sys.stderr.flush()
        raise SystemExit(3)
    return v


def precomp_sharded(points, k, tries=10, rots_before=6, rot_len_before=1, rots_after=1, rot_len_after=1, dist=None,
                    group=None, want_dists=False, _begin_hook=None):
    """precomp() with its distance passes spread over the ranks (include/ann_hip.h: annhip_precomp_begin ... _finish).

    points: torch device tensor [n,d], ALL rows, identical on every rank (the build needs every row on every GPU; the
    query path afterwards keeps only a row shard, Index.reshard).  Every rank must have seeded libc random() identically:
    the rotations are drawn from it, in the reference's order.  Returns an Index holding the complete index, bit-identical
    to the single-GPU build -- and to the reference's -- whatever the world size.  Three kinds of collectives: an
    all-gather of the hash codes per try (4 B / point), ONE MIN all-reduce pair over the merged candidate rows (every
    entry has exactly one writer: bucket b is scored by rank b mod world), an all-gather of the graph rows.
    _begin_hook: called right before annhip_precomp_begin (tests with several ranks in one process seed random() there)."""
    import ctypes as C

    from . import _lib
    from .api import Index
    prec = "f32" if points.dtype == torch.float32 else "f64"
    lib = _lib.load(prec)
    assert points.is_cuda and points.is_contiguous()
    on = dist is not None and dist.is_initialized() and (dist.get_world_size(group) > 1 or
                                                         os.environ.get("ANN_SHARD_FORCE_DIST") == "1")
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    via_cpu = on and dist.get_backend(group) == "gloo"
    n, d = points.shape
    if _begin_hook is not None:
        _begin_hook()
    h = lib.annhip_precomp_begin(n, k, d, points.data_ptr(), 1, tries, rots_before, rot_len_before, rots_after,
                                 rot_len_after, rank, world)
    # annhip_precomp_begin has made every draw this build takes from the caller's random() stream.  From here on torch
    # and RCCL set things up (allocations, the first collectives' connections, synchronisations) and the HIP runtime
    # has been seen drawing from random() in such moments: the stream is parked until the build is done, so that the
    # callers' next draws (bench.py: the query batches) are the same on every rank and in every run.
    with _lib.park_random():
        return _precomp_sharded_rest(lib, h, points, k, tries, dist, group, on, world, rank, via_cpu, want_dists, prec)


def _precomp_sharded_rest(lib, h, points, k, tries, dist, group, on, world, rank, via_cpu, want_dists, prec):
    import ctypes as C

    from .api import Index
    n, d = points.shape
    info = (C.c_size_t * 6)()
    lib.annhip_precomp_info(h, C.byref(info))
    Wn = int(info[1])
    dev, i32 = points.device, torch.int32

    def all_gather(out, inp):
        if not on:
            out.copy_(inp.reshape(out.shape))
        elif via_cpu:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu(), group=group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=group)

    def all_min(t):
        if not on:
            return
        if via_cpu:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MIN, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)

    rows_per = (n + world - 1) // world
    lo = min(n, rank * rows_per)
    hi = min(n, lo + rows_per)
    mi = torch.empty((n, Wn), dtype=i32, device=dev)
    md = torch.empty((n, Wn), dtype=points.dtype, device=dev)
    if on:
        lib.annhip_precomp_init_merged(h, mi.data_ptr(), md.data_ptr())
    codes_slice = torch.zeros((rows_per,), dtype=i32, device=dev)
    codes_all = torch.empty((world * rows_per,), dtype=i32, device=dev)
    for t in range(tries):
        lib.annhip_precomp_hash(h, t, lo, hi, codes_slice.data_ptr())
        all_gather(codes_all, codes_slice)
        lib.annhip_precomp_try(h, t, codes_all.data_ptr(), mi.data_ptr(), md.data_ptr())
    torch.cuda.synchronize(dev)
    all_min(mi)
    all_min(md)
    torch.cuda.synchronize(dev)
    lib.annhip_precomp_merge(h, mi.data_ptr(), md.data_ptr())
    del mi, md
    g_slice = torch.zeros((rows_per, k), dtype=i32, device=dev)
    gd_slice = torch.zeros((rows_per, k), dtype=points.dtype, device=dev)
    lib.annhip_precomp_graph(h, lo, hi, g_slice.data_ptr(), gd_slice.data_ptr())
    g_all = torch.empty((world * rows_per, k), dtype=i32, device=dev)
    all_gather(g_all, g_slice)
    gd_all = None
    if want_dists:
        gd_all = torch.empty((world * rows_per, k), dtype=points.dtype, device=dev)
        all_gather(gd_all, gd_slice)
    torch.cuda.synchronize(dev)
    ix = Index(prec, lib.annhip_precomp_finish(h, g_all.data_ptr()), keep=(points,))
    ix.graph_dists = gd_all[:n] if want_dists else None
    return ix
