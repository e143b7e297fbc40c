#!/bin/bash
# Diagnostic twin of the product library: plan_lean.hip compiled with -DEBOS_LEAN_STAMPS -> event_based_bos_amd/lib/libebos_lean_stamps.so
set -e
cd "$(dirname "$0")/.."
python -m event_based_bos_amd.build > /dev/null
mkdir -p /tmp/ebos_stamps
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -fno-gpu-rdc -Wno-unused-function -Iinclude -Ievent_based_bos_amd/csrc -DEBOS_LEAN_STAMPS"
/opt/rocm/bin/hipcc $FLAGS -x hip -c event_based_bos_amd/csrc/plan_lean.hip -o /tmp/ebos_stamps/plan_lean.o
OBJS=$(ls event_based_bos_amd/lib/obj/*.o | grep -v "plan_lean\.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o event_based_bos_amd/lib/libebos_lean_stamps.so /tmp/ebos_stamps/plan_lean.o $OBJS
