#!/bin/bash
# per-op breakdown (bench.py's serialised per-op timing) of two environments inside one box: r2_ab_prof.sh OUTDIR "ENV=A" "ENV=B"
set -e
out=gpurun_out/$1; shift
mkdir -p $out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 300 python bench.py --static-batch --no-cpu-baseline --no-rollout --steps 10 > $out/p${i}.json 2> $out/p${i}.err
done
