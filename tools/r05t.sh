set -x
mkdir -p gpurun_out/r05t
cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r05t/tests.log 2>&1; tail -4 gpurun_out/r05t/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05t/smoke.txt 2>&1; tail -3 gpurun_out/r05t/smoke.txt
