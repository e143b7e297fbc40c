#!/bin/bash
# Upper bound of "K hypotheses per event read" for the 2-DoF sweep (BASELINE configs[4]): ablation builds that warp and accumulate
# every event K times into ONE image (results wrong on purpose): tools/build_variant.py k<K> -DEBOS_ABL_MULTIK=<K>, then, on the GPU box,
#   tools/ab_multik.sh            -> kernel time per launch and per hypothesis-equivalent, K = 1 .. 4
for i in 1 2; do for k in 1 2 3 4; do
  EBOS_HIP_LIBRARY=ab/lib_k$k.so python bench.py --config 5 --min-seconds 0.2 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=$k
print('K', k, 'kernel_ms per launch', d['roofline']['kernel_ms'], 'per hypothesis-equivalent', round(d['roofline']['kernel_ms']/k*1e3,2), 'us; sweep ms per launch', round(d['ms_per_step']/512,5), '-> per hypothesis-equivalent', round(d['ms_per_step']/512/k*1e3,2), 'us')"
done; done
