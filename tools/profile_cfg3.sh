#!/bin/bash
# tools/profile_cfg3.sh OUT.md -- on the GPU box: the three rocprofv3 passes behind profiles/*_rocprof_summary.md and
# profiles/traffic.json for the default bench workload (kernel trace; FETCH_SIZE; WRITE_SIZE -- counters in their own
# runs, never combined with a trace), condensed by tools/summarize_profile.py.  The program comes directly after `--`.
set -e
out=${1:-gpurun_out/rocprof_summary.md}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-api --no-overlap-extra"
rm -rf gpurun_out/prof_kt gpurun_out/prof_f gpurun_out/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 $args > gpurun_out/prof_kt.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_f -- python3 $args > gpurun_out/prof_f.log 2>&1
echo "FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_w -- python3 $args > gpurun_out/prof_w.log 2>&1
echo "WRITE_SIZE done"
python3 tools/summarize_profile.py $(find gpurun_out/prof_kt -name "*kernel_trace.csv") $(find gpurun_out/prof_f -name "*counter_collection.csv") \
   $(find gpurun_out/prof_w -name "*counter_collection.csv") 2560000 $out "round 3: $args (cfg3 N=10M d=128 k=10 Q=10k f32, reference-stream data), MI355X"
grep "^{" gpurun_out/prof_kt.log > gpurun_out/prof_kt_bench.json
rm -rf gpurun_out/prof_kt gpurun_out/prof_f gpurun_out/prof_w
