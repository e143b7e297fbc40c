set -x
mkdir -p gpurun_out/r05f
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_value" > gpurun_out/r05f/test_value.log 2>&1; tail -30 gpurun_out/r05f/test_value.log
timeout 1500 python -m pytest tests/test_solver.py -q -m gpu -k "resident or translation or trajectory or window_pipeline" > gpurun_out/r05f/test_solver.log 2>&1; tail -30 gpurun_out/r05f/test_solver.log
timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05f/bench_fused.json 2> gpurun_out/r05f/bench_fused.err; tail -c 1500 gpurun_out/r05f/bench_fused.json
EBOS_VALUE_FUSED=0 timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05f/bench_three.json 2> gpurun_out/r05f/bench_three.err; tail -c 600 gpurun_out/r05f/bench_three.json
