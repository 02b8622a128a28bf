# kernel timeline of the one-rank-over-RCCL run under given settings:  bash tools/rccl1_trace.sh TAG "ENV=1 ENV2=2"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for kv in $1; do export $kv; done
export ANN_SHARD_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29643
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rccl1_kt_$tag -- python3 bench.py --gpus 1 --steps 8 --warmup 2 --data randn --no-strong-extra --cpu-seconds 1 > gpurun_out/rccl1_kt_$tag.log 2>&1
python3 tools/lanes_timeline.py $(find gpurun_out/rccl1_kt_$tag -name "*kernel_trace.csv") 200 > gpurun_out/rccl1_kt_${tag}_timeline.txt
rm -rf gpurun_out/rccl1_kt_$tag
