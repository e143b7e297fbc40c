set -x
mkdir -p gpurun_out/r05e
cd /root/repo
timeout 1500 python -m pytest tests/test_solver.py -q -m gpu > gpurun_out/r05e/test_solver.log 2>&1; tail -40 gpurun_out/r05e/test_solver.log
