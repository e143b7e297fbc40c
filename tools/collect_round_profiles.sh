#!/bin/bash
# Everything the round's profiles/ entries come from, on the GPU box (run through gpurun from the repo root):
#   tools/collect_round_profiles.sh <tag>        e.g. r02
# 1. bench.py (default command)                                  -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the SAME command        -> gpurun_out/<tag>/kernel_stats_bench.csv
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, of tools/profile_step.py (fwd + bwd of the objective)
#                                                                -> gpurun_out/<tag>/pmc_{fetch,write}/ -> pmc.json
# (PMC passes carry --kernel-trace only; no sys / hip / memory-copy tracing beside counters.)
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
ROOT=$PWD
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof_bench.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/tools/profile_step.py" --iters 10 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/tools/profile_step.py" --iters 10 > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
mkdir -p "$OUT/pmc_both" && cp -r "$OUT/pmc_fetch" "$OUT/pmc_both/" && cp -r "$OUT/pmc_write" "$OUT/pmc_both/"
python tools/make_pmc_json.py "$OUT/pmc_both" "$OUT/pmc.json" "${COMMIT:-unknown}" > "$OUT/pmc_summary.txt" 2>&1
find "$OUT/prof_bench" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_bench.csv"
rm -rf "$OUT/pmc_both"
# keep the merge small: drop the per-dispatch traces
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
head -12 "$OUT/kernel_stats_bench.csv" | cut -c1-200
cat "$OUT/pmc_summary.txt" | head -40
