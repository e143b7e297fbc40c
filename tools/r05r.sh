set -x
mkdir -p gpurun_out/r05r
cd /root/repo
timeout 1800 python -m pytest tests -q -m gpu -x -k "weight or iwe or golden or polarity or derived or fuzz" > gpurun_out/r05r/tests.log 2>&1; tail -8 gpurun_out/r05r/tests.log
timeout 900 python bench.py --weighted --no-cpu-baseline --no-extras > gpurun_out/r05r/bench_weighted.json 2> gpurun_out/r05r/bench_weighted.err; tail -c 1500 gpurun_out/r05r/bench_weighted.json; tail -3 gpurun_out/r05r/bench_weighted.err
