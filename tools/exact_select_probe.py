"""Latency of the exact path's network kernel (annhip_exact_select, stage-1 rows of a cfg3-shaped index) for 1..32 rows,
register form (default) vs LDS form (ANN_HIP_REGNET=0).  python tools/exact_select_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import approximatenn_amd as A

n, d, k, T = 2_000_000, 128, 10, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
pts = torch.randn((n, d), device=dev, generator=g)
ctypes.CDLL("libc.so.6").srandom(1)
ix = A.Index.precomp(pts, k, T)
print("L1 %d P1 %d Lc1 %d" % (ix.L1, ix.P1, ix.Lc1))
lib = ix.lib
for mode in ("1", "0"):
    os.environ["ANN_HIP_REGNET"] = mode
    A._lib.reload_env()
    for rows in (1, 2, 4, 8, 16, 32):
        ids = torch.randint(0, n, (rows, ix.Lc1), device=dev, dtype=torch.int32)
        dd = torch.rand((rows, ix.Lc1), device=dev)
        oi = torch.empty((rows, k), device=dev, dtype=torch.int32); od = torch.empty((rows, k), device=dev)
        ts = []
        for rep in range(6):
            i2, d2 = ids.clone(), dd.clone()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            lib.annhip_exact_select(ix.h, 1, rows, i2.data_ptr(), d2.data_ptr(), None, oi.data_ptr(), od.data_ptr())
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        print("regnet=%s rows=%2d: %.1f us (min of 6; first %.1f)" % (mode, rows, min(ts), ts[0]))
