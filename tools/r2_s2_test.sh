#!/bin/bash
set -e
mkdir -p gpurun_out/r2s
timeout -k 10 400 python -u -m pytest tests/test_gpu_conv.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r2s/conv.log 2>&1
timeout -k 10 400 python -u -m pytest tests/test_gpu_lockstep.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r2s/lockstep.log 2>&1
