#!/bin/bash
# tools/ab_run.sh NAME... -- on the GPU box: the default bench (cfg3, randn data, no extras) and the G=8 rank emulation for
# each A/B build under approximatenn_amd/csrc/ab/ ("base" = the in-tree library).  Prints one line per variant.
for name in "$@"; do
  if [ "$name" = base ]; then unset ANN_HIP_LIBDIR; else export ANN_HIP_LIBDIR=$PWD/approximatenn_amd/csrc/ab/$name; fi
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-api --data randn > gpurun_out/ab_$name.bench.log 2>&1
  python3 - "$name" gpurun_out/ab_$name.bench.log <<'PY'
import json, sys
name, path = sys.argv[1:3]
line = [l for l in open(path) if l.startswith("{")]
if not line:
    print(name, "bench FAILED"); sys.exit(0)
d = json.loads(line[0])
print("%-10s G=1: %.3f ms/step  stage1 %.4f ms (%.1f%% of peak)  overlap %.3f ms  precomp %.2f s" % (
    name, d["ms_per_step"], d["roofline"]["kernel_ms"], 100 * d["roofline"]["frac"], d.get("overlap", {}).get("ms_per_step", 0),
    d["config"]["precomp_s"]))
PY
  if [ -z "$AB_NO_EMUL" ]; then
    python3 tools/emulate_rank.py --worlds 8 > gpurun_out/ab_$name.emul.log 2>&1
    grep -E "^G=8|stage1 \(own|total" gpurun_out/ab_$name.emul.log | tr '\n' ' '; echo
  fi
done
