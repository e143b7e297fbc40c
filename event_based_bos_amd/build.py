"""Build libebos_hip.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m event_based_bos_amd.build [--force] [--keep-temps]

The shared library lands in event_based_bos_amd/lib/ (git-ignored, but it travels with the
gpurun snapshot).  hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libebos_hip.so")

SOURCES = ["errors.cpp", "warp_kernels.hip", "splat_kernels.hip", "event_plan.hip", "plan_lean.hip", "iwe_fused.hip", "iwe_tiled.hip",
           "cost_kernels.hip", "flow_upsample.hip", "image_filters.hip", "solver_kernels.hip", "cmax_resident.hip",
           "iwe_tiled_64x64x32.hip", "iwe_tiled_45x80x32.hip", "iwe_tiled_32x64x32.hip", "iwe_tiled_32x32x32.hip", "iwe_tiled_64x64x16.hip",
           "iwe_tiled_45x80x16.hip", "iwe_tiled_32x32x16.hip", "iwe_tiled_32x32x8.hip",
           "cmax_resident_45x80.hip", "cmax_resident_32x32.hip", "cmax_resident_32x64.hip",
           "cmax_resident_45x80_2dof.hip", "cmax_resident_32x32_2dof.hip", "cmax_resident_32x64_2dof.hip"]

# -munsafe-fp-atomics: hardware global_atomic_add_f32/f64 and ds_add_f32 instead of CAS loops.
HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fPIC",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libebos_hip.so cannot be built (ROCm toolchain required)")
    return exe


def _needs_rebuild(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# per-file flags.  cmax_resident_<tile>.hip (kernel: cmax_resident_core.h): one kernel with a dozen phases inside an iteration loop -- LICM hoists every phase's
# thread-index arithmetic out of that loop, where it stays live across all phases, is spilled, and comes back through scratch
# loads each followed by a full vmcnt(0) wait (5.5 us for four pixels per thread in the epilogue).  MachineSink's
# sink-insts-to-avoid-spills puts those computations back next to their uses: 41 -> 7 spilled VGPRs.
_RESIDENT_FLAGS = ["-mllvm", "-sink-insts-to-avoid-spills=1"]
PER_FILE_FLAGS = {f"cmax_resident_{t}{p}.hip": _RESIDENT_FLAGS for t in ("45x80", "32x32", "32x64") for p in ("", "_2dof")}


def _compile(src: str, extra: List[str]) -> str:
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(INCLUDE, "ebos_hip.h"), __file__]
    if _needs_rebuild(obj, [path] + headers):
        cmd = [hipcc()] + HIPCC_FLAGS + PER_FILE_FLAGS.get(src, []) + extra + ["-x", "hip", "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, keep_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    extra = ["-save-temps=obj"] if keep_temps else []
    extra += os.environ.get("EBOS_EXTRA_FLAGS", "").split()  # e.g. -DEBOS_STAMPS for the diagnostic build
    with cf.ThreadPoolExecutor(max_workers=min(int(os.environ.get("EBOS_BUILD_JOBS", "7")), len(SOURCES))) as ex:
        # (the slow units first: a tile configuration's kernels take 20 - 30 s, most other units a few)
        order = sorted(SOURCES, key=lambda s: 0 if s.startswith("iwe_tiled_") else (1 if s.startswith("cmax_resident_") else 2))
        built = dict(zip(order, ex.map(lambda s: _compile(s, extra), order)))
        objs = [built[s] for s in SOURCES]
    if force or _needs_rebuild(LIB_PATH, objs):
        cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[ebos build] {LIB_PATH} ({os.path.getsize(LIB_PATH) // 1024} KiB)")
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
