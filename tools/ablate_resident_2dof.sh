#!/bin/bash
# Timing builds of the resident 2-DoF kernel (32 x 32 unit; cmax_resident_core.h, EBOS_ABL) at the reference YAML's size
# (346 x 260, 100 k events, integer and fractional coordinates, with and without blur 3) -- as tools/ablate_resident.sh.
#   tools/ablate_resident_2dof.sh build "0 16384 ..."   (here)      tools/ablate_resident_2dof.sh run "0 16384 ..."   (GPU box)
# masks of the blur: 16384 forward interior loop, 32768 adjoint interior loop, 65536 forward border fix-up, 131072 adjoint border fix-up,
# 524288 the clearing of what lies outside the image (45 x 80 at 346 x 260, blur 3: 25.2 us whole; 23.4 / 23.1 / - / - / 24.6 without a piece)
cd "$(dirname "$0")/.."
MASKS=${2:-"0 16384 32768 65536 131072 245760"}
UNIT=${UNIT:-32x32}   # the resident 2-DoF unit to rebuild (32x32 | 32x64 | 45x80); run: TILE="45 80" picks the plan's tile
if [ "$1" = "build" ]; then
  python -m event_based_bos_amd.build > /dev/null
  mkdir -p /tmp/ebos_abl
  OBJS=$(ls event_based_bos_amd/lib/obj/*.o | grep -v "cmax_resident_${UNIT}_2dof\.o")
  CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -fno-gpu-rdc -Wno-unused-function -Iinclude -Ievent_based_bos_amd/csrc -mllvm -sink-insts-to-avoid-spills=1"
  for M in $MASKS; do
    ( $CC -DEBOS_ABL=$M -x hip -c event_based_bos_amd/csrc/cmax_resident_${UNIT}_2dof.hip -o /tmp/ebos_abl/c$M.o && \
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o event_based_bos_amd/lib/libebos_abl$M.so /tmp/ebos_abl/c$M.o $OBJS ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait
  ls event_based_bos_amd/lib/ | grep abl | tr '\n' ' '
else
  for M in $MASKS; do
    echo -n "mask $M: "
    EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_abl$M.so TILE="${TILE:-}" python tools/bench_frac_2dof.py 2>&1 | grep "260x346" | grep -o "[a-z]* coordinates, blur [0-9.]*: {'resident': [0-9.]*" | tr '\n' ';'
    echo
  done
fi
