set -x
mkdir -p gpurun_out/r05l
cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r05l/tests.log 2>&1; tail -12 gpurun_out/r05l/tests.log
