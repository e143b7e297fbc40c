#!/bin/bash
# VALU / LDS / wait counters of the step's kernels (fwd + bwd of the objective, tools/profile_step.py), three --pmc passes with
# --kernel-trace only; summary -> gpurun_out/<tag>/pmc_util.txt      usage (through gpurun, repo root): tools/collect_pmc_util.sh <tag>
set -u
TAG=${1:-pmcutil}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/p1" -- python3 "$ROOT/tools/profile_step.py" --iters 6 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/p2" -- python3 "$ROOT/tools/profile_step.py" --iters 6 > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d "$OUT/p3" -- python3 "$ROOT/tools/profile_step.py" --iters 6 > "$OUT/p3.log" 2>&1
cd "$ROOT"
python tools/pmc_summary.py "$OUT/p1" "$OUT/p2" "$OUT/p3" > "$OUT/pmc_util.txt" 2>&1
find "$OUT" -name "*.csv" -size +256k -delete
cat "$OUT/pmc_util.txt" | head -80
