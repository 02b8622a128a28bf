import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import approximatenn_amd as A
from oracle import oracle_py as O
n, d, k, T, Q = [int(v) for v in sys.argv[1:6]]
orc = O.CpuBackend("f32", "oracle")
O.srandom(2024); orc.rand_norm_reset()
pts = orc.gen_rand(n * d).reshape(n, d); y = orc.gen_rand(Q * d).reshape(Q, d)
O.srandom(7); o_ids, o_d, o_save = orc.precomp(pts, k, T)
for mode in ("select", "exact"):
    if mode == "exact": os.environ["ANN_HIP_EXACT"] = "1"
    else: os.environ.pop("ANN_HIP_EXACT", None)
    O.srandom(7); ids, dists, save = A.precomp(pts, k, T)
    s = save.to_dict()
    print(mode, "ds", s["d_short"], o_save["d_short"], "pm", list(map(int, s["par_maxes"])), list(map(int, o_save["par_maxes"])))
    print("  means eq", np.array_equal(s["row_means"].view(np.uint32), o_save["row_means"].view(np.uint32)),
          "bases eq", np.array_equal(s["bases"].view(np.uint32), o_save["bases"].view(np.uint32)))
    for t in range(T):
        if s["which_par"][t].shape != o_save["which_par"][t].shape: print("  which_par shape differs", t)
        elif not np.array_equal(s["which_par"][t], o_save["which_par"][t]): print("  which_par differs try", t, int(np.sum(s["which_par"][t] != o_save["which_par"][t])))
    bad = np.where((ids != o_ids).any(axis=1))[0]
    print("  graph rows differing:", len(bad), bad[:10], " dist bits differing rows:", int(np.sum((dists.view(np.uint32) != o_d.view(np.uint32)).any(axis=1))))
    for r in bad[:3]:
        print("   row", r, "gpu", ids[r], dists[r]); print("        ", "cpu", o_ids[r], o_d[r])
    A._lib.load("f32").annhip_cache_clear(); save.free()
