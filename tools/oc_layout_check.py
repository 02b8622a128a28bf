import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import approximatenn_amd as A
from oracle import oracle_py as O
for d in [int(v) for v in os.environ.get("OC_DIMS", "96,160,48").split(",")]:
    n, Q, k, T = 5000, 300, 10, 5
    orc = O.CpuBackend("f32", "oracle")
    O.srandom(5 + d); orc.rand_norm_reset()
    pts = orc.gen_rand(n * d).reshape(n, d); y = orc.gen_rand(Q * d).reshape(Q, d)
    O.srandom(7); o_ids, o_d, o_save = orc.precomp(pts, k, T)
    O.srandom(7); ids, dd, save = A.precomp(pts, k, T)
    ok1 = np.array_equal(ids, o_ids) and np.array_equal(dd.view(np.uint32), o_d.view(np.uint32))
    want, got = orc.query(o_save, pts, y), A.query(save, pts, y)
    ok2 = np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    print("d=%d OC_C=%s precomp %s query %s" % (d, os.environ.get("ANN_HIP_OC_C"), ok1, ok2))
    A._lib.load("f32").annhip_cache_clear(); save.free()
