#!/bin/bash
# A/B of library variants / environments inside one box: r2_ab_lib.sh OUTDIR "ENV=.. ENV=.." ...  (each run twice, interleaved; default bench step with
# the per-step batch gather, 16 steps)
set -e
out=gpurun_out/$1; shift
mkdir -p $out
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-rollout --no-profile --steps 16 > $out/v${i}_r${rep}.json 2> $out/v${i}_r${rep}.err
    echo "$e rep $rep: $(python -c "import json,sys; d=[json.loads(l) for l in open('$out/v${i}_r${rep}.json') if l.startswith('{')][-1]; print(d['ms_per_step'], d['breakdown_ms_per_step'].get('WGRAD_MFMA') if 'breakdown_ms_per_step' in d else '')")"
  done
done
