cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="bench.py --points 4000000 --dim 80 --steps 6 --warmup 2 --no-cpu-baseline --no-host-api --no-overlap-extra --data randn"
python3 $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('d80 N=4M', d['value'], d['ms_per_step'], d['roofline'])"
rm -rf gpurun_out/p80
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p80/f -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/p80/s -- python3 $args > /dev/null 2>&1
python3 tools/pmc_table.py stage1_select $(find gpurun_out/p80 -name "*counter_collection.csv")
rm -rf gpurun_out/p80
args="bench.py --points 4000000 --dim 64 --steps 6 --warmup 2 --no-cpu-baseline --no-host-api --no-overlap-extra --data randn"
python3 $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('d64 N=4M', d['value'], d['ms_per_step'], d['roofline'])"
