set -x
mkdir -p gpurun_out/r05i
cd /root/repo
timeout 2400 python -m pytest tests/test_solver.py -q -m gpu > gpurun_out/r05i/test_solver.log 2>&1; tail -40 gpurun_out/r05i/test_solver.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "blur" > gpurun_out/r05i/test_blur.log 2>&1; tail -5 gpurun_out/r05i/test_blur.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r05i/bench.json 2> gpurun_out/r05i/bench.err; tail -c 200 gpurun_out/r05i/bench.json
