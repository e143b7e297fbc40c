set -x
mkdir -p gpurun_out/r05g
cd /root/repo
timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05g/bench.json 2> gpurun_out/r05g/bench.err; tail -c 400 gpurun_out/r05g/bench.json
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r05g/tests.log 2>&1; tail -30 gpurun_out/r05g/tests.log
