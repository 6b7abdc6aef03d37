#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2e
timeout -k 10 600 python -u -m pytest tests/test_gpu_physics_loss.py tests/test_gpu_metrics.py tests/test_hbm_loader.py tests/test_gpu_trainer.py -q --timeout 300 --timeout-method thread > gpurun_out/r2e/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r2e/tests.log
FNET_WGRAD_LANE=1 timeout -k 10 500 python bench.py > gpurun_out/r2e/bench_lane1.json 2> gpurun_out/r2e/bench_lane1.err || { tail -5 gpurun_out/r2e/bench_lane1.err; exit 1; }
FNET_WGRAD_LANE=1 timeout -k 10 300 python bench.py --static-batch --no-cpu-baseline --no-rollout --no-profile > gpurun_out/r2e/bench_static.json 2> gpurun_out/r2e/bench_static.err || exit 1
cut -c1-600 gpurun_out/r2e/bench_lane1.json; echo; python tools/bench_line.py gpurun_out/r2e/bench_static.json static
