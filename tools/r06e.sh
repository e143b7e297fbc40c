set -x
cd /root/repo
mkdir -p gpurun_out/r06e
O=gpurun_out/r06e
timeout 1200 python -m pytest tests/test_ingest.py -q -m gpu -x > $O/tests_ingest.log 2>&1; tail -4 $O/tests_ingest.log
for S in f64 raw; do EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_lean_stamps.so python tools/stamp_plan_lean.py --source $S > $O/stamps_$S.txt 2>&1; cat $O/stamps_$S.txt | grep -v amdgpu.ids; done
python tools/bench_plan_build.py > $O/plan_build.json 2> $O/plan_build.err; cat $O/plan_build.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -- python3 /root/repo/tools/bench_plan_build.py > /dev/null 2> /root/repo/$O/prof.err
cd /root/repo
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/plan_build_kernel_stats.csv
rm -rf $O/prof
grep lean $O/plan_build_kernel_stats.csv | cut -c1-60,200-400
