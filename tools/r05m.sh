set -x
mkdir -p gpurun_out/r05m
cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05m/smoke.log 2>&1; tail -8 gpurun_out/r05m/smoke.log
timeout 900 python tools/bench_skew_solver.py --out gpurun_out/r05m/skew_solver.json > gpurun_out/r05m/skew_solver.log 2>&1; grep -v amdgpu gpurun_out/r05m/skew_solver.log
timeout 1200 python -m pytest tests/test_solver.py -q -m gpu -k "resident or spill or pipeline" > gpurun_out/r05m/tests.log 2>&1; tail -4 gpurun_out/r05m/tests.log
