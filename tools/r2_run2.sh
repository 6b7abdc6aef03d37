#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 1000 python -u -m pytest tests/test_gpu_golden_rollout.py tests/test_gpu_k4_parity.py tests/test_gpu_configs.py -v --timeout 300 --timeout-method thread -s > gpurun_out/r2b/tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r2b/tests.log
FNET_WGRAD_LANE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --dump-ops gpurun_out/r2b/ops.csv > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err || exit 1
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/r2b/tests.log | tail -30
