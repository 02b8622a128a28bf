"""A CPU stand-in for approximatenn_amd.sharded.HipEngine, built on the oracle -- TEST INFRASTRUCTURE.

It implements the staged per-shard steps (include/ann_hip.h) in numpy + oracle calls so that the multi-rank
orchestration in approximatenn_amd/sharded.py (exchanges, merge, fallback) can run under gloo without a GPU.
It restates the per-shard kernels' contracts, not their code.
"""
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

    # ---- HipEngine interface
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

    def finalize(self, cd, ci, nv):
        k, K1 = self.k, self.k + 1
        cdn, cin, nvn = cd.numpy(), ci.numpy().view(np.uint32), nv.numpy().view(np.uint32)
        Q = cdn.shape[0]
        top_i = np.zeros((Q, k), dtype=np.uint32)
        top_d = np.zeros((Q, k), dtype=self.npft)
        flagged = []
        for x in range(Q):
            m = int(np.sum(cin[x] != 0xFFFFFFFF))
            flag = k > self.P1 or m < k
            if not flag:
                flag = not np.isfinite(cdn[x, k - 1]) or any(cdn[x, t] == cdn[x, t + 1] for t in range(m - 1)) \
                    or (self.L1 > self.P1 and nvn[x] >= self.P1)
            if flag:
                flagged.append(x)
            else:
                top_i[x], top_d[x] = cin[x, :k], cdn[x, :k]
        return (torch.from_numpy(top_i.view(np.int32)), torch.from_numpy(top_d),
                torch.from_numpy(np.array(flagged, dtype=np.int32)))

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
