#!/bin/bash
# tools/r03_artifacts.sh -- on the GPU box: the measurements behind profiles/r03_* (one gpurun call, ~12 minutes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python3 bench.py > gpurun_out/r03/cfg3_bench.json.log 2> gpurun_out/r03/cfg3_bench.err; echo "bench done"
tests/harness/time_results_f32 -n 10000000 -d 128 -k 10 -y 10000 -o 12 -S 12345 -F -P 3 -C 64 > gpurun_out/r03/cfg3_time_results.log 2>&1; echo "time_results done"
tests/harness/time_results_f32 -n 10000000 -d 128 -k 10 -y 10000 -o 8 -S 12345 -V 2 -C 16 > gpurun_out/r03/cfg3_time_results_2_virtual_shards.log 2>&1
tests/harness/time_results_f32 -n 10000000 -d 128 -k 10 -y 10000 -o 8 -S 12345 -V 8 -C 16 > gpurun_out/r03/cfg3_time_results_8_virtual_shards.log 2>&1; echo "virtual shards done"
ANN_HIP_DEVICES=1 ANN_HIP_FORCE_RCCL=1 tests/harness/time_results_f32 -n 10000000 -d 128 -k 10 -y 10000 -o 8 -S 12345 -C 16 > gpurun_out/r03/cfg3_time_results_one_device_rccl.log 2>&1; echo "rccl host done"
bash tools/profile_cfg3.sh gpurun_out/r03/cfg3_rocprof_summary.md > gpurun_out/r03/profile.log 2>&1; cp gpurun_out/prof_kt_bench.json gpurun_out/r03/cfg3_profiled_run_bench.json 2>/dev/null; echo "rocprof done"
python3 tools/emulate_rank.py --rccl --worlds 1,2,4,8 --lanes 7 --schedules "3,1,0,1;7,2,0,1;3,0,0,1;3,1,0,1,3" > gpurun_out/r03/emulated_rank_rccl.txt 2>&1
python3 tools/emulate_rank.py --worlds 1,2,4,8 --lanes 7 --schedules "3,1,0,1;7,2,0,1" > gpurun_out/r03/emulated_rank_loopback.txt 2>&1; echo "emulation done"
( for d in 64 80 96 128 160 256; do AB_DIM=$d bash tools/ab_d80.sh base; done ) > gpurun_out/r03/row_length_sweep.txt 2>&1; echo "sweep done"
./tools/readbw > gpurun_out/r03/readbw_ceilings.log 2>&1
ANN_SHARD_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29642 bench.py --gpus 1 2>gpurun_out/r03/one_rank_rccl.err | grep "^{" > gpurun_out/r03/bench_one_rank_rccl.log; echo "one rank rccl done"
python3 tools/run_configs.py --out gpurun_out/r03/configs --only cfg1,cfg2,cfg4,cfg5 > gpurun_out/r03/configs.jsonl 2> gpurun_out/r03/configs.err; echo "configs done"
