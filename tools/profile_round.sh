#!/bin/bash
# Round-end evidence in one GPU call: the default bench line (with cpu_baseline, the 600-frame rollout leg and the config-4 legs), the
# rocprofv3 kernel statistics + trace of the same training workload, three PMC passes (FETCH_SIZE, WRITE_SIZE, the MFMA counters; each
# in a run of its own, the program directly after `--`), and their summaries, all under gpurun_out/<tag>_profiles/ -- copy that directory's
# files into profiles/ afterwards.     usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
TAG=$1
ROOT=$PWD
OUT=$ROOT/gpurun_out/${TAG}_profiles
mkdir -p $OUT
CMD="--steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-rollout --no-config4"
timeout -k 10 500 python bench.py > $OUT/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
python tools/bench_line.py $OUT/${TAG}_bench.json.log bench
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG} -o t -- python $ROOT/bench.py $CMD > $ROOT/gpurun_out/prof_${TAG}.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_f_${TAG} -o f -- python $ROOT/bench.py $CMD > $ROOT/gpurun_out/pmc_f_${TAG}.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_w_${TAG} -o w -- python $ROOT/bench.py $CMD > $ROOT/gpurun_out/pmc_w_${TAG}.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pmc_m_${TAG} -o m -- python $ROOT/bench.py $CMD > $ROOT/gpurun_out/pmc_m_${TAG}.log 2>&1
cd $ROOT
STATS=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1)
TRACE=$(find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" | head -1)
python - "$STATS" "$OUT/${TAG}_train_k4_b64_kernel_stats.csv" <<'PY'
import csv, re, sys
src, dst = sys.argv[1], sys.argv[2]
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]
rows = list(csv.reader(open(src)))
with open(dst, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-rollout --no-config4 (5 steps of K4 B64 128^2 bf16)\n")
    w = csv.writer(f)
    for r in rows:
        if r and r[0] != "Name":
            r[0] = short(r[0])
        w.writerow(r)
PY
python tools/pmc_traffic.py $(find gpurun_out/pmc_f_${TAG} -name "*counter_collection.csv") $(find gpurun_out/pmc_w_${TAG} -name "*counter_collection.csv") $OUT/${TAG}_pmc_traffic.json "python bench.py $CMD" > $OUT/${TAG}_pmc_traffic.txt
python tools/pmc_mfma.py $(find gpurun_out/pmc_m_${TAG} -name "*counter_collection.csv") $OUT/${TAG}_pmc_mfma.json "$STATS" "python bench.py $CMD" > $OUT/${TAG}_pmc_mfma.txt
python tools/lane_overlap.py "$TRACE" > $OUT/${TAG}_lane_overlap.txt
head -12 $OUT/${TAG}_pmc_traffic.txt; cat $OUT/${TAG}_pmc_mfma.txt; head -24 $OUT/${TAG}_lane_overlap.txt
