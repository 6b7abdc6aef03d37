#!/bin/bash
# Same-box A/B of library variants / environments (boxes of the pool spread +-3 %, so only runs inside ONE gpurun call compare):
#   tools/ab_env.sh OUTDIR "ENV=.. ENV=.." "ENV=.." ...     each variant REPS times (default 2), interleaved; the default bench step, 16 steps
# A variant library comes from tools/build_variant.sh (FNET_LIB=...), or from a copy of an older build.
set -e
out=gpurun_out/$1; shift
mkdir -p $out
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-rollout --no-config4 ${AB_FLAGS:---no-profile} --steps 16 > $out/v${i}_r${rep}.json 2> $out/v${i}_r${rep}.err
    echo "[$e] rep $rep: $(python -c "import json,sys; d=[json.loads(l) for l in open('$out/v${i}_r${rep}.json') if l.startswith('{')][-1]; b=d.get('breakdown_ms_per_step',{}); print(d['ms_per_step'], {k:b[k] for k in ('CONV3X3_MFMA','WGRAD_MFMA','NORM_ACT_BWD','NORM_ACT_FWD') if k in b})")"
  done
done
