#!/bin/bash
# r2: side-lane bias-gradient / loss sums, aliased residual gradient; wgrad block count below one per CU
set -e
mkdir -p gpurun_out/r2g
timeout -k 10 300 python -u -m pytest tests/test_gpu_lockstep.py tests/test_gpu_golden_rollout.py -x -q -s --timeout 120 --timeout-method thread > gpurun_out/r2g/tests.log 2>&1
for wb in 128 192 256; do
  FNET_WG_BLOCKS=$wb timeout -k 10 200 python bench.py --static-batch --no-cpu-baseline --no-rollout --no-profile --steps 12 > gpurun_out/r2g/wg_${wb}.json 2> gpurun_out/r2g/wg_${wb}.err
done
FNET_KEEP_G=1 timeout -k 10 200 python bench.py --static-batch --no-cpu-baseline --no-rollout --no-profile --steps 12 > gpurun_out/r2g/keepg.json 2> gpurun_out/r2g/keepg.err
