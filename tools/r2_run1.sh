#!/bin/bash
# round 2, first GPU call: new parity tests + lane / micro-batch experiments
set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -u -m pytest tests/test_gpu_golden_rollout.py tests/test_gpu_k4_parity.py tests/test_gpu_configs.py -x -v --timeout 300 --timeout-method thread -s > gpurun_out/r2a/tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r2a/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_lane0.json 2> gpurun_out/r2a/bench_lane0.err || exit 1
FNET_WGRAD_LANE=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_lane1.json 2> gpurun_out/r2a/bench_lane1.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile > gpurun_out/r2a/bench_lane0b.json 2>> gpurun_out/r2a/bench_lane0.err || exit 1
FNET_WGRAD_LANE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile > gpurun_out/r2a/bench_lane1b.json 2>> gpurun_out/r2a/bench_lane1.err || exit 1
timeout -k 10 300 python tools/exp_concurrent.py 8 > gpurun_out/r2a/concurrent.json 2> gpurun_out/r2a/concurrent.err || exit 1
tail -n 3 gpurun_out/r2a/bench_lane0b.json gpurun_out/r2a/bench_lane1b.json gpurun_out/r2a/concurrent.json | cut -c1-400
