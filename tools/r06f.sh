set -x
cd /root/repo
mkdir -p gpurun_out/r06f
O=gpurun_out/r06f
python tools/experiments/dbg_lean_hot.py 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_ingest.py -q -m gpu > $O/tests_ingest.log 2>&1; tail -6 $O/tests_ingest.log
for S in f64 raw; do EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_lean_stamps.so python tools/stamp_plan_lean.py --source $S > $O/stamps_$S.txt 2>&1; cat $O/stamps_$S.txt | grep -v amdgpu.ids; done
python tools/bench_plan_build.py > $O/plan_build.json 2> $O/plan_build.err; cat $O/plan_build.json
