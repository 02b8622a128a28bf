"""A CPU stand-in for approximatenn_amd.sharded.HipEngine, built on the oracle -- TEST INFRASTRUCTURE.

It implements the staged per-shard steps (include/ann_hip.h: annhip_sh_* and the exact-path calls) in numpy + oracle
calls so that the multi-rank orchestration in approximatenn_amd/sharded.py (owner protocol, exchanges, repair) can run
under gloo without a GPU.  It restates the per-shard kernels' contracts, not their code.
"""
import contextlib

import numpy as np
import torch

from oracle import oracle_py as O


def lg(x):
    r = 0
    while x >> 1:
        x >>= 1
        r += 1
    return r


def need_len(L, k):
    if L < 16:
        return L
    P = 1 << lg(L)
    return min(L, max(P, k) + 1)


class CpuShardEngine:
    def __init__(self, save, points, lo, hi, prec="f32"):
        self.prec, self.save, self.points, self.lo, self.hi = prec, save, points, lo, hi
        self.orc = O.CpuBackend(prec, "oracle")
        self.hs = O.HostSave(save, prec)
        self.npft = np.float32 if prec == "f32" else np.float64
        self.ft = torch.float32 if prec == "f32" else torch.float64
        self.n, self.k, self.T, self.ds = int(save["n"]), int(save["k"]), int(save["tries"]), int(save["d_short"])
        self.pm = [int(v) for v in save["par_maxes"]]
        self.off = np.concatenate([[0], np.cumsum([p * (self.ds + 1) for p in self.pm])]).astype(int)
        self.L1 = int(self.off[-1])
        self.P1 = 1 << lg(self.L1)
        self.Lc1 = need_len(self.L1, self.k)
        self.L2 = self.k * (self.k + 1)
        self.Lc2 = need_len(self.L2, self.k)

    # ---- helpers
    def _row_ids(self, x, Q, codes, upto):
        ids = np.empty(upto, dtype=np.int64)
        j = 0
        for t in range(self.T):
            pm, code = self.pm[t], int(codes[t * Q + x]) & 0xFFFFFFFF   # Q2 read layout
            tab = self.save["which_par"][t]
            for yy in range(self.ds + 1):
                b = code ^ ((1 << (yy - 1)) if yy else 0)
                take = min(pm, upto - j)
                if take <= 0:
                    return ids
                ids[j:j + take] = tab[b, :take]
                j += take
        return ids

    def _dist(self, a, pid):
        df = (a - self.points[pid]).astype(self.npft)
        return self.orc.tree_sum(df * df)

    def _ok(self, pid, x, alias):
        return pid < self.n and not (alias and pid == x)

    def _own(self, pid):
        return self.lo <= pid < self.hi

    # ---- HipEngine interface: lanes and buffers
    key_words = property(lambda self: 1 if self.prec == "f32" else 2)

    def new_stream(self, device, high_priority=False, reserve_cus=0):
        return None

    def use(self, stream):
        return contextlib.nullcontext()

    def set_gather_pieces(self, pieces):      # a launch-shape knob of the device engine: nothing to do here
        pass

    def empty(self, shape, dtype, like):
        return torch.empty(shape, dtype=dtype)

    def _bits(self, dv):
        return int(np.array(dv, dtype=self.npft).view(np.uint32 if self.prec == "f32" else np.uint64))

    # ---- owner protocol (annhip_sh_*)
    def sh_codes(self, y, q_lo, q_hi, out):
        Q = y.shape[0]
        c = self.orc.query_codes(self.hs, y.numpy()).astype(np.uint32)        # [q*T+t] for the whole batch
        hi = min(q_hi, Q)
        if hi > q_lo:
            out.numpy().view(np.uint32)[: (hi - q_lo) * self.T] = c[q_lo * self.T: hi * self.T]

    def sh_stage1(self, y, alias, codes, keys, nvalid, nown, stream=None):
        Q, K1, kw = y.shape[0], self.k + 1, self.key_words
        cd, ci, nv = self.stage1_local(y, alias, codes)
        cdn, cin = cd.numpy(), ci.numpy().view(np.uint32)
        kk = keys.numpy().view(np.uint64)
        for x in range(Q):
            for t in range(K1):
                b = self._bits(cdn[x, t])
                if kw == 1:
                    kk[x, t] = (b << 32) | int(cin[x, t])
                else:
                    kk[x, 2 * t], kk[x, 2 * t + 1] = b, int(cin[x, t])
        nvalid.numpy()[:] = nv.numpy()
        nown.numpy()[:] = 0

    def _unpack(self, row, t):
        if self.key_words == 1:
            v = int(row[t])
            return v >> 32, v & 0xFFFFFFFF
        return int(row[2 * t]), int(row[2 * t + 1])

    def sh_merge_finalize(self, G, Q, q_lo, qs, keys_in, nvalid, top_i, top_d):
        k, K1 = self.k, self.k + 1
        kin = keys_in.numpy().view(np.uint64).reshape(G, qs, -1)
        nvn = nvalid.numpy().view(np.uint32)
        ti, td = top_i.numpy().view(np.uint32), top_d.numpy()
        bt = np.uint32 if self.prec == "f32" else np.uint64
        for xl in range(qs):
            if q_lo + xl >= Q:
                ti[xl], td[xl] = 0xFFFFFFFF, np.inf
                continue
            cand = []
            for g in range(G):
                for t in range(K1):
                    b, i = self._unpack(kin[g, xl], t)
                    if i != 0xFFFFFFFF:
                        cand.append((b, i))
            cand = sorted(cand)[:K1]
            m = len(cand)
            flag = k > self.P1 or m < k
            if not flag:
                dk = np.array([cand[k - 1][0]], dtype=bt).view(self.npft)[0]
                flag = not np.isfinite(dk) or any(cand[t][0] == cand[t + 1][0] for t in range(m - 1)) \
                    or (self.L1 > self.P1 and nvn[q_lo + xl] >= self.P1)
            for t in range(min(k, m)):
                ti[xl, t] = cand[t][1]
                td[xl, t] = np.array([cand[t][0]], dtype=bt).view(self.npft)[0]
            if flag:
                ti[xl, 0] = 0xFFFFFFFE

    def sh_exact1_begin(self, y, alias, codes, top_all, fcap, flist, rows_i, rows_d):
        Q = y.shape[0]
        ta, fl = top_all.numpy().view(np.uint32), flist.numpy().view(np.uint32)
        flagged = [x for x in range(Q) if ta[x, 0] == 0xFFFFFFFE]
        fl[0], fl[1] = min(len(flagged), fcap), len(flagged)
        fl[2:2 + fl[0]] = flagged[:fcap]
        if fl[0]:
            ids, dd = self.stage1_rows(y, alias, codes, torch.from_numpy(fl[2:2 + fl[0]].astype(np.int32)))
            rows_i.numpy()[: fl[0]] = ids.numpy()
            rows_d.numpy()[: fl[0]] = dd.numpy()

    def sh_exact1_end(self, Q, q_lo, qs, fcap, flist, rows_i, rows_d, top_all, top_d_all, top_i, top_d):
        fl = flist.numpy().view(np.uint32)
        nl = int(fl[0])
        if not nl:
            return
        qidx = torch.from_numpy(fl[2:2 + nl].astype(np.int32))
        self.exact_select(1, rows_i[:nl], rows_d[:nl], qidx, top_all, top_d_all)
        for x in fl[2:2 + nl]:
            if q_lo <= x < q_lo + qs:
                top_i.numpy()[x - q_lo] = top_all.numpy()[x]
                top_d.numpy()[x - q_lo] = top_d_all.numpy()[x]

    def _stage2_id(self, ti_row, j):
        k, n = self.k, self.n
        parent, z = int(ti_row[j // k - 1]), j % k
        return int(self.save["graph"][parent, z]) if parent < n else (int(self.save["graph"][0, z]) | n)

    def sh_stage2(self, y, alias, top_all, dist_out, flagged):
        Q, k = y.shape[0], self.k
        yn, ta = y.numpy(), top_all.numpy().view(np.uint32)
        do, fl = dist_out.numpy(), flagged.numpy().view(np.uint32)
        fl[0] = 0
        for x in range(Q):
            if ta[x, 0] == 0xFFFFFFFE:
                fl[1 + fl[0]] = x
                fl[0] += 1
                continue
            for j in range(k, self.Lc2):
                pid = self._stage2_id(ta[x], j)
                do[x, j - k] = self._dist(yn[x], pid) if (self._ok(pid, x, alias) and self._own(pid)) else np.inf
        if fl[0] > 1:   # the kernel appends with atomics in no particular order
            fl[1:1 + fl[0]] = fl[1:1 + fl[0]][::-1].copy()

    def sh_final(self, G, Q, q_lo, qs, top_i, top_d, dist_in, out_i, out_d):
        k = self.k
        ti, td = top_i.numpy().view(np.uint32), top_d.numpy()
        din = dist_in.numpy().reshape(G, qs, -1)
        oi, od = out_i.numpy().view(np.uint32), out_d.numpy()
        for xl in range(qs):
            if q_lo + xl >= Q or ti[xl, 0] == 0xFFFFFFFE:
                oi[xl], od[xl] = (0xFFFFFFFF if q_lo + xl >= Q else 0xFFFFFFFE), np.inf
                continue
            rid = np.empty(self.Lc2, dtype=np.uint64)
            rdd = np.empty(self.Lc2, dtype=self.npft)
            rid[:k], rdd[:k] = ti[xl], td[xl]
            for j in range(k, self.Lc2):
                rid[j] = self._stage2_id(ti[xl], j) & 0xFFFFFFFF
                rdd[j] = din[:, xl, j - k].min()
            L = self.L2
            rid = np.concatenate([rid, (1 << 40) + np.arange(L - self.Lc2, dtype=np.uint64)])
            rdd = np.concatenate([rdd, np.full(L - self.Lc2, np.inf, dtype=self.npft)])
            s_ids, s_d = self.orc.topk_stage(rid, rdd)
            oi[xl], od[xl] = s_ids[:k].astype(np.uint32), s_d[:k]

    def stage2_rows_list(self, y, alias, qidx, top_i, top_d):
        ids, dd = self.stage2_rows(y, alias, top_i, top_d)
        sel = torch.from_numpy(qidx.numpy().view(np.uint32).astype(np.int64))
        return ids[sel].contiguous(), dd[sel].contiguous()

    # ---- per-shard building blocks
    def codes(self, y):
        c = self.orc.query_codes(self.hs, y.numpy())
        return torch.from_numpy(c.astype(np.uint32).view(np.int32).copy())

    def stage1_local(self, y, alias, codes):
        Q, K1 = y.shape[0], self.k + 1
        yn, cn = y.numpy(), codes.numpy()
        cd = np.full((Q, K1), np.inf, dtype=self.npft)
        ci = np.full((Q, K1), 0xFFFFFFFF, dtype=np.uint32)
        nv = np.zeros(Q, dtype=np.uint32)
        for x in range(Q):
            ids = self._row_ids(x, Q, cn, self.P1)
            keys = set()
            for pid in ids:
                pid = int(pid)
                if self._ok(pid, x, alias):
                    nv[x] += 1
                    if self._own(pid):
                        dv = self._dist(yn[x], pid)
                        keys.add((int(np.array(dv).view(np.uint32 if self.prec == "f32" else np.uint64)), pid, dv))
            for i, (_, pid, dv) in enumerate(sorted(keys)[:K1]):
                cd[x, i], ci[x, i] = dv, pid
        return torch.from_numpy(cd), torch.from_numpy(ci.view(np.int32)), torch.from_numpy(nv.view(np.int32))

    def stage1_rows(self, y, alias, codes, qidx):
        Q = y.shape[0]
        yn, cn = y.numpy(), codes.numpy()
        nq = qidx.shape[0]
        ids = np.zeros((nq, self.Lc1), dtype=np.uint32)
        dd = np.full((nq, self.Lc1), np.inf, dtype=self.npft)
        for r, x in enumerate(qidx.numpy()):
            row = self._row_ids(int(x), Q, cn, self.Lc1)
            ids[r] = row.astype(np.uint32)
            for j, pid in enumerate(row):
                pid = int(pid)
                if self._ok(pid, int(x), alias) and self._own(pid):
                    dd[r, j] = self._dist(yn[int(x)], pid)
        return torch.from_numpy(ids.view(np.int32)), torch.from_numpy(dd)

    def stage2_rows(self, y, alias, top_i, top_d):
        Q, k, n = y.shape[0], self.k, self.n
        yn = y.numpy()
        ti, td = top_i.numpy().view(np.uint32), top_d.numpy()
        graph = self.save["graph"]
        ids = np.zeros((Q, self.Lc2), dtype=np.uint32)
        dd = np.full((Q, self.Lc2), np.inf, dtype=self.npft)
        for x in range(Q):
            for j in range(self.Lc2):
                if j < k:
                    ids[x, j], dd[x, j] = ti[x, j], td[x, j]
                    continue
                parent, z = int(ti[x, j // k - 1]), j % k
                pid = int(graph[parent, z]) if parent < n else (int(graph[0, z]) | n)
                ids[x, j] = pid & 0xFFFFFFFF
                if self._ok(pid, x, alias) and self._own(pid):
                    dd[x, j] = self._dist(yn[x], pid)
        return torch.from_numpy(ids.view(np.int32)), torch.from_numpy(dd)

    def exact_select(self, stage, ids, dd, qidx, out_i, out_d):
        L = self.L1 if stage == 1 else self.L2
        idn, ddn = ids.numpy().view(np.uint32), dd.numpy()
        oi, od = out_i.numpy().view(np.uint32), out_d.numpy()
        for r in range(idn.shape[0]):
            stored = idn.shape[1]
            rid = np.concatenate([idn[r].astype(np.uint64), (1 << 40) + np.arange(L - stored, dtype=np.uint64)])
            rdd = np.concatenate([ddn[r], np.full(L - stored, np.inf, dtype=self.npft)])
            s_ids, s_d = self.orc.topk_stage(rid, rdd)
            x = int(qidx.numpy()[r]) if qidx is not None else r
            oi[x], od[x] = s_ids[:self.k].astype(np.uint32), s_d[:self.k]
