#!/bin/bash
# The resident solver kernel's regression loop on the GPU box: its tests, the per-iteration times, optionally the in-kernel stamps.
#   tools/ab_resident.sh [stamps]
timeout 900 python -m pytest tests/test_solver.py -m gpu -x -q -s -k "resident or run_modes or trajectory or window_pipeline" 2>&1 | grep -v "^$" | tail -8
best() {  # the minimum of five runs of 300 iterations (box-to-box and run-to-run noise is ~0.3 us: below that nothing can be told)
  for i in 1 2 3 4 5; do python tools/profile_solver.py "$@" --halo auto --mode resident 2>&1 | tail -1 | grep -o "[0-9.]* us/iteration" | cut -d" " -f1; done | sort -n | head -1
}
for args in "--events 2000000" "--size 260 346 --events 100000" "--events 10000000" "--events 200000"; do
  echo "NEW  $args: $(best $args) us/iteration (min of 5)"
done
if [ -f event_based_bos_amd/lib/libebos_prev.so ]; then   # A/B on the same box: the library of the previous build
  for args in "--events 2000000" "--size 260 346 --events 100000"; do
    echo "PREV $args: $(EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_prev.so best $args) us/iteration (min of 5)"
  done
fi
python tools/profile_solver.py --events 2000000 --halo auto --mode pipeline 2>&1 | tail -1 | sed -E 's/\(status 0\), //; s/sample_grid.*gradient 0.0: //' 
if [ "${1:-}" = "stamps" ]; then
  export EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_stamps.so
  python tools/stamp_resident.py 2>&1 | tail -23
  python tools/stamp_resident.py --size 260 346 --events 100000 --patch 20 20 2>&1 | tail -23
fi
