#!/bin/bash
# tools/ab_d80.sh NAME... -- on the GPU box: the d = 80 bench (N=4M, Q=10k, randn data) for each A/B build ("base" = in-tree)
for name in "$@"; do
  if [ "$name" = base ]; then unset ANN_HIP_LIBDIR; else export ANN_HIP_LIBDIR=$PWD/approximatenn_amd/csrc/ab/$name; fi
  python3 bench.py --points 4000000 --dim ${AB_DIM:-80} --steps 8 --warmup 2 --no-cpu-baseline --no-host-api --no-overlap-extra --data randn 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$name d=${AB_DIM:-80}: %.3f ms/step, stage1 %.4f ms, %.0f GB/s = %.1f%% of peak' % (d['ms_per_step'], r['kernel_ms'], r['achieved'], 100*r['frac']))"
done
