#!/bin/bash
# Timing builds of the resident solver kernels (cmax_resident_core.h, EBOS_ABL; the patch-grid units of the 45 x 80 and 32 x 32 tiles): each leaves pieces of the iteration out (results wrong
# on purpose) -- what a piece costs WHERE IT STANDS is the difference to the whole.  In-kernel stamps cannot tell: a stamp orders the
# code around it, and the phases of a workgroup overlap across waves.
#   tools/ablate_resident.sh build "0 1 2 ..."   (here: cross-compiles lib/libebos_abl<mask>.so)
#   tools/ablate_resident.sh run "0 1 2 ..."     (on the GPU box)
cd "$(dirname "$0")/.."
MASKS=${2:-"0 1 2 4 8 16 32 64 128 256 512 1024 2048 4096"}
if [ "$1" = "build" ]; then
  python -m event_based_bos_amd.build > /dev/null
  mkdir -p /tmp/ebos_abl
  OBJS=$(ls event_based_bos_amd/lib/obj/*.o | grep -v "cmax_resident_45x80\.o" | grep -v "cmax_resident_32x32\.o")
  CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -fno-gpu-rdc -Wno-unused-function -Iinclude -Ievent_based_bos_amd/csrc -mllvm -sink-insts-to-avoid-spills=1"
  for M in $MASKS; do
    ( $CC -DEBOS_ABL=$M -x hip -c event_based_bos_amd/csrc/cmax_resident_45x80.hip -o /tmp/ebos_abl/a$M.o && \
      $CC -DEBOS_ABL=$M -x hip -c event_based_bos_amd/csrc/cmax_resident_32x32.hip -o /tmp/ebos_abl/b$M.o && \
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o event_based_bos_amd/lib/libebos_abl$M.so /tmp/ebos_abl/a$M.o /tmp/ebos_abl/b$M.o $OBJS ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait
  ls event_based_bos_amd/lib/ | grep abl | tr '\n' ' '
else
  for M in $MASKS; do
    for ARGS in "--events 2000000" "--size 260 346 --events 100000"; do
      echo -n "mask $M $ARGS: "
      EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_abl$M.so python tools/profile_solver.py $ARGS --halo auto --mode resident --lr 0 2>&1 | tail -1 | grep -o "[0-9.]* us/iteration\|status -[0-9]*"
    done
  done
fi
# (blur passes of the resident kernel: masks 16384 = no forward interior loop, 32768 = no adjoint interior loop, 65536 = no border
#  fix-ups; measured on the 2-DoF 32 x 32 unit at 346 x 260 / 100 k events, blur 3: 21.3 us whole, 20.6 / 20.3 / 18.0 without one
#  piece, 15.7 without all three -- the border fix-ups of 36 of 99 tiles were more than half of what the blur cost)
