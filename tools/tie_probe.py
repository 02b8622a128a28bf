"""Latency of one flagged row through exact_select with and without the tie path (run under rocprofv3 --kernel-trace
--stats: the two template instantiations show up as separate kernels)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import approximatenn_amd as A  # noqa: E402

L, k, n = int(sys.argv[1]) if len(sys.argv) > 1 else 6006, 10, 1 << 20
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
ft = np.float32 if prec == "f32" else np.float64
lib = A._lib.load(prec)
P = 1 << (L.bit_length() - 1)
ln = min(L, max(P, k) + 1)
rng = np.random.default_rng(1)
ids = np.full((1, ln), n, np.uint32)
dist = np.full((1, ln), np.inf, ft)
pool = rng.choice(n, size=1400, replace=False)
dv = (rng.integers(1 << 10, 1 << 20, size=pool.size).astype(ft)) / ft(64)
dv[5] = dv[9] = dv.min() + 1
slots = rng.choice(P, size=1400, replace=False)
ids[0, slots] = pool
dist[0, slots] = dv
keys = sorted(set((float(dist[0, j]), int(ids[0, j])) for j in range(P) if np.isfinite(dist[0, j])))[:k + 1]
cand_d = np.array([[a for a, b in keys]], ft)
cand_i = np.array([[b for a, b in keys]], np.uint32)
out = {}
for use in (0, 1, 2):  # the network / tie path with the given list / tie path with the list derived from the row
    oi = torch.zeros((1, k), dtype=torch.int32, device="cuda")
    od = torch.zeros((1, k), dtype=torch.float32 if prec == "f32" else torch.float64, device="cuda")
    cnt = torch.zeros(16, dtype=torch.int64, device="cuda")
    cd, ci = torch.from_numpy(cand_d).cuda(), torch.from_numpy(cand_i.view(np.int32)).cuda()
    for it in range(20):
        ti, td = torch.from_numpy(ids.copy()).cuda(), torch.from_numpy(dist.copy()).cuda()
        torch.cuda.synchronize()
        lib.annhip_test_sort_rows(L, k, 1, ti.data_ptr(), td.data_ptr(), cd.data_ptr() if use == 1 else None,
                                  ci.data_ptr() if use == 1 else None, oi.data_ptr(), od.data_ptr(), cnt.data_ptr(), 1 if use == 2 else 0)
    torch.cuda.synchronize()
    out[use] = (oi.cpu().numpy().copy(), od.cpu().numpy().copy(), int(cnt[0].item()))
    if use and os.environ.get("ANN_TIE_TS"):
        ts = cnt.cpu().numpy()[1:9]
        print("phase timestamps (us since entry): list %.1f scan %.1f counts %.1f sort1 %.1f rdups %.1f sort2 %.1f out %.1f"
              % tuple((ts[i] - ts[0]) / 100.0 for i in range(1, 8)))
assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
assert np.array_equal(out[0][0], out[2][0]) and np.array_equal(out[0][1], out[2][1])
print("L=%d %s: same result; rows answered by the tie path: %d of 20 (list given), %d of 20 (list derived)" % (L, prec, out[1][2], out[2][2]))
