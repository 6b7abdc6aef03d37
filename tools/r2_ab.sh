#!/bin/bash
# A/B inside one box (clocks differ from box to box by ~2 %): usage r2_ab.sh OUTDIR "ENV=A" "ENV=B" ...  (each run twice, interleaved)
set -e
out=gpurun_out/$1; shift
mkdir -p $out
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 200 python bench.py --static-batch --no-cpu-baseline --no-rollout --no-profile --steps 16 > $out/v${i}_r${rep}.json 2> $out/v${i}_r${rep}.err
    echo "$e rep $rep: $(python -c "import json,sys; d=[json.loads(l) for l in open('$out/v${i}_r${rep}.json') if l.startswith('{')][-1]; print(d['ms_per_step'])")"
  done
done
