#!/bin/bash
set -e
mkdir -p gpurun_out/r2ew
export FNET_EW_LAYERS=128x64,64x128

timeout -k 10 600 python -u -m pytest tests/test_gpu_lockstep.py tests/test_gpu_api.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r2ew/tests.log 2>&1
