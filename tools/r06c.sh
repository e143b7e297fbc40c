set -x
cd /root/repo
mkdir -p gpurun_out/r06c
O=gpurun_out/r06c
timeout 1200 python -m pytest tests/test_ingest.py -q -m gpu -x > $O/tests_ingest.log 2>&1; tail -8 $O/tests_ingest.log
python tools/bench_plan_build.py > $O/plan_build.json 2> $O/plan_build.err; cat $O/plan_build.json; tail -3 $O/plan_build.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -- python3 /root/repo/tools/bench_plan_build.py > /dev/null 2> /root/repo/$O/prof.err
cd /root/repo
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/plan_build_kernel_stats.csv
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete
head -20 $O/plan_build_kernel_stats.csv | cut -c1-180
