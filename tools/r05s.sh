set -x
mkdir -p gpurun_out/r05s
cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r05s/tests.log 2>&1; tail -6 gpurun_out/r05s/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05s/bench.json 2> gpurun_out/r05s/bench.err; tail -c 300 gpurun_out/r05s/bench.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05s/smoke.txt 2>&1; tail -3 gpurun_out/r05s/smoke.txt
