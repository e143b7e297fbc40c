set -x
mkdir -p gpurun_out/r05h
cd /root/repo
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r05h/bench.json 2> gpurun_out/r05h/bench.err; tail -c 300 gpurun_out/r05h/bench.json; tail -3 gpurun_out/r05h/bench.err
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_config5.py -q -m gpu -x > gpurun_out/r05h/tests.log 2>&1; tail -5 gpurun_out/r05h/tests.log
