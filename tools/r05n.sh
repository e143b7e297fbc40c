set -x
mkdir -p gpurun_out/r05n
cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "gradient_magnitude or gradmag or config3 or gm" > gpurun_out/r05n/tests_gm.log 2>&1; tail -8 gpurun_out/r05n/tests_gm.log
timeout 900 python bench.py --config 3 > gpurun_out/r05n/bench_config3.json 2> gpurun_out/r05n/bench_config3.err; tail -c 900 gpurun_out/r05n/bench_config3.json
