#!/bin/bash
# Everything the round's profiles/ entries come from, on the GPU box (run through gpurun from the repo root):
#   tools/collect_round_profiles.sh <tag>        e.g. r03
# 1. (last, on the counters of step 3) bench.py: default command, --config 3 / 4 / 5, --fractional, --weighted, --steps 20, and
#    tools/bench_autograd.py (the Python autograd routes)              -> gpurun_out/<tag>/bench*.json, autograd_path.json
# 2. rocprofv3 --kernel-trace --stats of the SAME default command     -> gpurun_out/<tag>/kernel_stats_bench.csv
# 3. rocprofv3 --pmc, four separate passes (FETCH_SIZE | WRITE_SIZE | instruction counts | LDS / wait counters) of
#    tools/profile_step.py --mode all (dense fwd + bwd under both contrasts, the batched patch-grid pass, the solver loop as
#    four launches and as the resident kernel, the 2-DoF sweep; built halo and run-time windows)
#                                                                      -> gpurun_out/<tag>/pmc.json, pmc_summary.txt
# (PMC passes carry --kernel-trace only; no sys / hip / memory-copy tracing beside counters.)
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof_bench.err"
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU")
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc/pass$i" -- python3 "$ROOT/tools/profile_step.py" --mode all --iters 6 --out-dir "$OUT/pmc" > "$OUT/pmc_pass$i.log" 2>&1
  i=$((i+1))
done
cd "$ROOT"
python tools/make_pmc_json.py "$OUT/pmc" "$OUT/pmc.json" "${COMMIT:-unknown}" > "$OUT/pmc_summary.txt" 2>&1
PMC_RC=$?
find "$OUT/prof_bench" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_bench.csv"
# the bench lines read profiles/pmc_latest.json / kernel_stats_latest.json: install this collection's (on the box) before they run
# -- but only a collection that holds what they read.  A failed or empty rocprofv3 pass still yields a JSON stamped with the current
# blob hash and an empty kernel table; installed, it would replace the committed evidence by "no counters" (ADVICE r03).
python tools/make_kernel_stats_json.py "$OUT/kernel_stats_bench.csv" "$OUT/kernel_stats.json" "${COMMIT:-unknown}" > /dev/null
KS_RC=$?
python - "$OUT/pmc.json" "$OUT/kernel_stats.json" "$PMC_RC" "$KS_RC" <<'PY'
import json, sys
pmc_path, ks_path, pmc_rc, ks_rc = sys.argv[1:5]
ok = pmc_rc == "0" and ks_rc == "0"
try:
    k = json.load(open(pmc_path))["kernels"]
    acc = next((v for n, v in k.items() if "iwe_slab_accumulate_kernel" in n), None)
    ok = ok and acc is not None and acc.get("hbm_bytes_per_launch") and acc.get("SQ_INSTS_VALU")
    ks = json.load(open(ks_path))["kernels"]
    ok = ok and any("iwe_slab_accumulate_kernel" in n for n in ks)
except Exception as e:  # missing / malformed file
    print("collect_round_profiles: unreadable collection:", e, file=sys.stderr)
    ok = False
sys.exit(0 if ok else 1)
PY
if [ $? -ne 0 ]; then
  echo "collect_round_profiles: the counter / kernel-stats collection is incomplete; profiles/*_latest.json left as they were" >&2
  exit 1
fi
cp "$OUT/pmc.json" profiles/pmc_latest.json
cp "$OUT/kernel_stats.json" profiles/kernel_stats_latest.json
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --config 3 > "$OUT/bench_config3.json" 2> "$OUT/bench_config3.err"
python bench.py --fractional --no-cpu-baseline > "$OUT/bench_fractional.json" 2> "$OUT/bench_fractional.err"
python bench.py --weighted --no-cpu-baseline > "$OUT/bench_weighted.json" 2> "$OUT/bench_weighted.err"
python tools/bench_autograd.py > "$OUT/autograd_path.json" 2> "$OUT/autograd_path.err"
python bench.py --config 4 > "$OUT/bench_config4.json" 2> "$OUT/bench_config4.err"
python bench.py --config 5 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"   # the driver's form
# keep the merge small: drop the per-dispatch traces
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
head -12 "$OUT/kernel_stats_bench.csv" | cut -c1-200
cat "$OUT/pmc_summary.txt" | head -40
