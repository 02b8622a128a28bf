#!/bin/bash
# tools/ab_env.sh "VAR=val ..." "VAR=val ..." ... -- on the GPU box: the default bench (cfg3, randn data, no extras) once per
# environment setting, interleaved twice (box drift shows as the difference between the two rounds).  One line each.
for round in 1 2; do
for setting in "$@"; do
  env $setting python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-api --data randn > gpurun_out/ab_env.log 2>&1
  python3 - "$setting" gpurun_out/ab_env.log <<'PY'
import json, sys
name, path = sys.argv[1:3]
line = [l for l in open(path) if l.startswith("{")]
if not line:
    print(name, "bench FAILED"); sys.exit(0)
d = json.loads(line[0])
st = d["config"]["stage_ms_per_step"]
print("%-36s %.4f ms/step  stage1 %.4f ms (%.1f%%)  codes %.3f fin %.3f s2 %.3f  overlap %.3f ms" % (
    name or "(default)", d["ms_per_step"], d["roofline"]["kernel_ms"], 100 * d["roofline"]["frac"], st["codes"], st["finalize_fallback"],
    st["stage2_rows"], d.get("overlap", {}).get("ms_per_step", 0)))
PY
done
done
