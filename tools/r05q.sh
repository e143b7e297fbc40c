set -x
mkdir -p gpurun_out/r05q
cd /root/repo
timeout 1500 python -m pytest tests/test_solver.py -q -m gpu -x -k "gradient_magnitude or resident or pyramid or translation or trajectory" > gpurun_out/r05q/tests.log 2>&1; tail -5 gpurun_out/r05q/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05q/bench.json 2> gpurun_out/r05q/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05q/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
si=d.get('solver_iteration')
for k in ('2M_events','10M_events','100k_events_346x260'):
    v=si[k]
    print(k, {m:(x.get('us_per_iteration') if 'us_per_iteration' in x else {a:b.get('us_per_iteration') for a,b in x.items()}) for m,x in v.items() if isinstance(x,dict)})
PY
