#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2d
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -v --timeout 120 --timeout-method thread > gpurun_out/r2d/noalloc.log 2>&1
echo "noalloc rc=$?"
timeout -k 10 1100 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/r2d/tests_all.log 2>&1
echo "all gpu tests rc=$?"
tail -4 gpurun_out/r2d/noalloc.log
tail -6 gpurun_out/r2d/tests_all.log
