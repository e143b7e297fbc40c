# Round 6, the round's profiles in one GPU call (run through gpurun from the repo root):   bash tools/r06z.sh <commit>
#   1. tools/collect_round_profiles.sh r06z: rocprofv3 kernel stats of the default bench command, the four PMC passes, then the bench lines
#      (default, --config 3 / 4 / 5, --fractional, --weighted, --steps 20) on the installed counters
#   2. the lean plan build: wall clock per window at 10 M / 2 M / 100 k events (tools/bench_plan_build.py), in-kernel phase stamps
#      (tools/stamp_plan_lean.py on the diagnostic twin of the library), rocprofv3 kernel stats of the 10 M build
#   3. small-flow lines of configs 2 and 4 (+-2, +-6 px, run-time windows)
#   4. the GPU test suite and smoke()
set -x
cd /root/repo
COMMIT=$1 bash tools/collect_round_profiles.sh r06z > gpurun_out/r06z_collect.log 2>&1; tail -3 gpurun_out/r06z_collect.log
O=gpurun_out/r06z
mkdir -p $O
python tools/bench_plan_build.py > $O/plan_build.json 2> $O/plan_build.err
python tools/bench_plan_build.py --events 2000000 > $O/plan_build_2m.json 2>> $O/plan_build.err
python tools/bench_plan_build.py --events 100000 --height 260 --width 346 > $O/plan_build_100k.json 2>> $O/plan_build.err
cat $O/plan_build*.json
for S in f64 raw; do EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_lean_stamps.so python tools/stamp_plan_lean.py --source $S 2>&1 | grep -v amdgpu.ids > $O/plan_lean_stamps_$S.txt; done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_plan -- python3 /root/repo/tools/bench_plan_build.py > /dev/null 2> /root/repo/$O/prof_plan.err)
find $O/prof_plan -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/plan_build_kernel_stats.csv; rm -rf $O/prof_plan
for F in 2 6; do
  python bench.py --flow-max $F --halo auto --no-cpu-baseline --no-extras > $O/bench_config2_flow$F.json 2> $O/e2_$F.err
  python bench.py --config 4 --flow-max $F --halo auto > $O/bench_config4_flow$F.json 2> $O/e4_$F.err
done
timeout 3000 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > $O/run_cmax_ref_346x260.json 2> $O/err1.txt
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 > $O/run_cmax_own_600.json 2> $O/err3.txt
cat $O/run_cmax*.json | cut -c1-900
