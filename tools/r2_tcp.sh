#!/bin/bash
set -e
mkdir -p gpurun_out/r2tcp
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r2tcp/counters.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
grep -o "TCP_[A-Z_0-9a-z]*" gpurun_out/r2tcp/counters.txt | sort -u > gpurun_out/r2tcp/tcp_names.txt || true
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2tcp/p1 -o a -- python $GRAFT_REPO_ROOT/tools/conv_bench.py run > $GRAFT_REPO_ROOT/gpurun_out/r2tcp/p1.log 2>&1
