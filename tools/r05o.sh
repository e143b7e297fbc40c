set -x
mkdir -p gpurun_out/r05o
cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r05o/tests.log 2>&1; tail -15 gpurun_out/r05o/tests.log
timeout 900 python bench.py --weighted --no-cpu-baseline --no-extras > gpurun_out/r05o/bench_weighted.json 2> gpurun_out/r05o/bench_weighted.err; tail -c 300 gpurun_out/r05o/bench_weighted.json
