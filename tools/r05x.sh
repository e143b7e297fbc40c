#!/bin/bash
# kernel trace of the four-launch loop on a clustered window (sigma 100 px, 2 M events)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05x
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05x/trace -o skew -- python3 tools/bench_skew_solver.py --events 2000000 --sigma 100 --no-uniform --modes pipeline --iters 200 > gpurun_out/r05x/run.log 2>&1
find gpurun_out/r05x/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05x/kernel_stats.csv
head -12 gpurun_out/r05x/kernel_stats.csv | cut -c1-200
tail -2 gpurun_out/r05x/run.log
find gpurun_out/r05x/trace -type f ! -name "*stats.csv" -delete
