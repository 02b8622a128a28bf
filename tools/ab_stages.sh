#!/bin/bash
# tools/ab_stages.sh NAME... -- like ab_run.sh, but prints the per-stage milliseconds of the serial step (HIP events inside the library)
for name in "$@"; do
  if [ "$name" = base ]; then unset ANN_HIP_LIBDIR; else export ANN_HIP_LIBDIR=$PWD/approximatenn_amd/csrc/ab/$name; fi
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-overlap-extra --data randn > gpurun_out/ab_$name.bench.log 2>&1
  python3 - "$name" gpurun_out/ab_$name.bench.log <<'PY'
import json, sys
name, path = sys.argv[1:3]
line = [l for l in open(path) if l.startswith("{")]
if not line:
    print(name, "bench FAILED"); sys.exit(0)
d = json.loads(line[0])
print("%-10s %.4f ms/step  %s" % (name, d["ms_per_step"], " ".join("%s %.4f" % kv for kv in d["config"]["stage_ms_per_step"].items())), flush=True)
PY
done
