#!/bin/bash
# Same-box A/B of ONE library under different environments: tools/ab_env.sh "NAME=VALUE" "NAME=VALUE" ...  (2 alternations)
for i in 1 2; do for v in "$@"; do
  env $v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --min-seconds 0.3 --rotating-windows 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step', d['ms_per_step'], 'acc', d['roofline']['kernel_ms'], 'comb', d['combine_kernel_ms']['mean'], 'fwd+bwd', d['fwd_bwd_ms'], 'bwd', d['roofline_bwd']['kernel_ms'], 'contrast', d['contrast'], 'solver', d.get('solver_iteration',{}).get('us_per_iteration'))"
done; done
