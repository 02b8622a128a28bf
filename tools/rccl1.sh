# one rank over real RCCL, full cfg3 size: A/B of lanes, of the FRONT/BACK split and of CUs kept free of the gathers.
#   bash tools/rccl1.sh "ANN_SHARD_TUNE=2,1,0" "ANN_SHARD_TUNE=2,1,8" "" ...
# columns: settings, queries/s, ms/step, stage-1 kernel ms, host ms/step
run() {  # $1 = "VAR=value ..." (ANN_SHARD_TUNE=depth,split,reserve pins the schedule)
  env ANN_SHARD_FORCE_DIST=1 $1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29642 bench.py --gpus 1 --steps 20 --warmup 3 --data randn --no-strong-extra --cpu-seconds 1 2>/dev/null | grep "^{" > gpurun_out/rccl1.json
  python3 - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/rccl1.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["host_submit_ms_per_step"], flush=True)
PY
}
for s in "$@"; do run "$s"; done
