#!/bin/bash
# Same-box A/B of libebos_hip.so builds on the solver iteration: tools/ab_solver.sh g0 g1   (libraries ab/lib_<name>.so)
for i in 1 2; do for v in "$@"; do
  EBOS_HIP_LIBRARY=ab/lib_$v.so python tools/bench_solver.py --events 2000000 2>/dev/null | grep -i "fused=True" | head -2 | sed "s/^/$v /"
done; done
