#!/bin/bash
# tools/sq_counters.sh OUT.md -- on the GPU box: SQ occupancy / wait / LDS counters of the stage-1 kernel for the default
# bench workload (torch.randn data), two rocprofv3 --pmc passes (counters in their own runs), condensed by tools/pmc_table.py.
set -e
out=${1:-gpurun_out/stage1_sq_counters.md}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-overlap-extra --data randn"
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_a -- python3 $args > gpurun_out/pmc_a.log 2>&1
echo "pass A done"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_b -- python3 $args > gpurun_out/pmc_b.log 2>&1
echo "pass B done"
python3 tools/pmc_table.py stage1_select_kernel $(find gpurun_out/pmc_a -name "*counter_collection.csv") $(find gpurun_out/pmc_b -name "*counter_collection.csv") > $out
python3 tools/pmc_table.py exact_select_kernel $(find gpurun_out/pmc_a -name "*counter_collection.csv") $(find gpurun_out/pmc_b -name "*counter_collection.csv") >> $out
python3 tools/pmc_table.py codes_lpq_kernel $(find gpurun_out/pmc_a -name "*counter_collection.csv") $(find gpurun_out/pmc_b -name "*counter_collection.csv") >> $out
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
