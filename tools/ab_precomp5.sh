#!/bin/bash
# tools/ab_precomp5.sh NAME... -- cfg5-size precomp (f64, N=10M d=256 k=100) for each A/B build; prints precomp seconds
for name in "$@"; do
  if [ "$name" = base ]; then unset ANN_HIP_LIBDIR; else export ANN_HIP_LIBDIR=$PWD/approximatenn_amd/csrc/ab/$name; fi
  python3 bench.py --dtype f64 --points 10000000 --dim 256 --knn 100 --queries 10000 --data randn --steps 1 --warmup 0 --no-cpu-baseline --no-host-api --no-overlap-extra > gpurun_out/ab5_$name.log 2>&1
  echo "$name $(grep -o 'precomp_s[^,]*' gpurun_out/ab5_$name.log)"
done
