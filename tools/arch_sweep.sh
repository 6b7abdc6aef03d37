mkdir -p gpurun_out/arch
i=0
for f in "--arch large" "--arch small" "--arch small-medium" "--k 2 --res 256 --depth 4 --batch 32" "--k 2 --res 512 --depth 5 --batch 8" "--mode rollout --arch medium --batch 1 --res 128 --steps 300" "--mode rollout --arch small --steps 100"; do
  i=$((i+1))
  timeout -k 10 250 python bench.py --steps 6 $f --no-cpu-baseline --no-rollout --no-config4 --no-profile > gpurun_out/arch/v$i.json 2> gpurun_out/arch/v$i.err || { echo "FAILED: $f"; tail -5 gpurun_out/arch/v$i.err; exit 1; }
  python -c "
import json; d=json.loads([l for l in open('gpurun_out/arch/v$i.json') if l.startswith('{')][-1]); print('$f', d['value'], d['unit'], d['ms_per_step'])"
done
