#!/bin/bash
# Same-box A/B of libebos_hip.so builds on the Python-API objective: tools/ab_autograd.sh j0 j1   (libraries ab/lib_<name>.so)
for i in 1 2 3; do for v in "$@"; do
  EBOS_HIP_LIBRARY=ab/lib_$v.so python tools/bench_autograd.py 2>/dev/null | head -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'variance_and_grad_dense', d['variance_and_grad_dense']['us_per_iteration'], 'contrast_dense.backward', d['contrast_dense_backward']['us_per_iteration'], 'single-threaded engine', d['contrast_dense_backward_single_threaded_engine']['us_per_iteration'])"
done; done
