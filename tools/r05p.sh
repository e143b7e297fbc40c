set -x
mkdir -p gpurun_out/r05p
cd /root/repo
timeout 900 python -m pytest tests/test_solver.py -q -m gpu -x -k "gradient_magnitude or resident" -s > gpurun_out/r05p/tests_gm.log 2>&1; tail -40 gpurun_out/r05p/tests_gm.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05p/bench.json 2> gpurun_out/r05p/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('solver_iteration'))[:5000])
PY
