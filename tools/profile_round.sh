#!/bin/bash
# Round-end evidence: default bench line (with cpu_baseline), rocprofv3 kernel stats of the same workload, two PMC passes.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
TAG=$1
ROOT=$PWD
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python tools/bench_line.py gpurun_out/${TAG}_bench.json bench
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG} -o t -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-rollout --no-config4 > $ROOT/gpurun_out/prof_${TAG}.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_f_${TAG} -o f -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-rollout --no-config4 > $ROOT/gpurun_out/pmc_f_${TAG}.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_w_${TAG} -o w -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-rollout --no-config4 > $ROOT/gpurun_out/pmc_w_${TAG}.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pmc_m_${TAG} -o m -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-rollout --no-config4 > $ROOT/gpurun_out/pmc_m_${TAG}.log 2>&1
cd $ROOT
find gpurun_out/pmc_m_${TAG} -name "*counter_collection.csv"
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1
find gpurun_out/pmc_f_${TAG} gpurun_out/pmc_w_${TAG} -name "*counter_collection.csv"
