#!/bin/bash
# A/B of bench.py flag sets inside one box: r2_ab_flags.sh OUTDIR "flags A" "flags B" ...  (each twice, interleaved)
set -e
out=gpurun_out/$1; shift
mkdir -p $out
for rep in 1 2; do
  i=0
  for f in "$@"; do
    i=$((i+1))
    timeout -k 10 200 python bench.py $f --no-cpu-baseline --no-rollout --no-profile --steps 16 > $out/v${i}_r${rep}.json 2> $out/v${i}_r${rep}.err
    echo "[$f] rep $rep: $(python -c "import json,sys; d=[json.loads(l) for l in open('$out/v${i}_r${rep}.json') if l.startswith('{')][-1]; print(d['ms_per_step'])")"
  done
done
