#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2c
timeout -k 10 1000 python -u -m pytest tests/test_gpu_golden_rollout.py tests/test_gpu_k4_parity.py -v --timeout 300 --timeout-method thread -s > gpurun_out/r2c/tests_new.log 2>&1
echo "new tests rc=$?"
FNET_WGRAD_LANE=1 timeout -k 10 1100 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/r2c/tests_all_lane1.log 2>&1
echo "all tests (lane 1) rc=$?"
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/r2c/tests_new.log | tail -20
tail -5 gpurun_out/r2c/tests_all_lane1.log
