#!/bin/bash
# tuning sweep with the weight-gradient lane on: blocks per weight-gradient launch / reduce launch
mkdir -p gpurun_out/r2f
for wb in 256 384 512 768; do
  for rb in 512 1024; do
    FNET_WG_BLOCKS=$wb FNET_WG_RED_BLOCKS=$rb timeout -k 10 200 python bench.py --static-batch --no-cpu-baseline --no-rollout --no-profile --steps 12 > gpurun_out/r2f/wg_${wb}_${rb}.json 2>/dev/null || exit 1
    echo "wg_blocks=$wb red_blocks=$rb $(python tools/bench_line.py gpurun_out/r2f/wg_${wb}_${rb}.json | cut -c1-40)"
  done
done
