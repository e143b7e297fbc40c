#!/bin/bash
# kernel trace of the reference YAML's window end to end (fractional events: full plan build + resident 2-DoF loop)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05y
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05y/trace -o e2e -- python3 tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > gpurun_out/r05y/run.log 2>&1
cp gpurun_out/r05y/trace/e2e_kernel_stats.csv gpurun_out/r05y/kernel_stats.csv
cp gpurun_out/r05y/trace/e2e_kernel_trace.csv gpurun_out/r05y/kernel_trace.csv
tail -2 gpurun_out/r05y/run.log | cut -c1-600
