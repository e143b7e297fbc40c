"""ctypes binding of libebos_hip.so (the C ABI declared in include/ebos_hip.h).

There is deliberately NO CPU fallback anywhere in this package: if the shared library or a GPU is
missing, every compute entry point raises ``HipUnavailableError`` -- loudly, at the call site.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libebos_hip.so")

# ebos_status / enums of include/ebos_hip.h
EBOS_OK = 0
REF_FIRST, REF_LAST, REF_FRACTION, REF_TIMEBASE = 0, 1, 2, 3
SPLAT_BILINEAR, SPLAT_COUNT, SPLAT_POLARITY = 0, 1, 2
GAUSS_REFLECT_SCIPY, GAUSS_REFLECT_TORCH = 0, 1
PROFILE_SLAB_ACCUMULATE, PROFILE_TILED_BWD, PROFILE_SLAB_COMBINE, PROFILE_GRADMAG_FUSED = 0, 1, 2, 3
ABI_VERSION = 2



# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable says otherwise) in creation order, and
# two streams on one queue run their kernels one after the other: resident solver launches of different windows then take turns instead
# of running side by side (solver.WindowPipeline: four 346 x 260 windows in flight take 3.5 ms each on queues of their own, 6.3 ms on
# shared ones).  The variable is read when the runtime initialises: where the process has not said anything and has not touched the GPU
# yet, ask for 16 queues (eight windows in flight, beside the ingest and the default stream: 3.1 ms each against 4.1 with four); ``hw_queues()`` is what the pipeline may count on.
_HW_QUEUES = 4


def _runtime_initialised() -> bool:
    """Has ANY HIP user of this process brought the runtime up already?  The ROCm runtime opens /dev/kfd when it initialises (which
    is also when it reads GPU_MAX_HW_QUEUES); torch's own ``torch.cuda.is_initialized()`` is its lazy-init flag and stays False after
    ``torch.cuda.is_available()`` or a ctypes HIP call has initialised the runtime with the default four queues.
    Unknown (no /proc) counts as initialised: the pipeline then plans for four queues, which is slower but never aliases streams."""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return True
            except OSError:
                continue
        return False
    except OSError:
        return True


def configure_queues(n: int = 16) -> int:
    """Ask the HIP runtime for ``n`` hardware queues (GPU_MAX_HW_QUEUES) if that can still take effect -- the variable is not set and
    the runtime is not up yet -- and return what ``hw_queues()`` reports from now on.  Runs at import with n = 16 unless
    EBOS_NO_QUEUE_ENV=1 (then the process environment is left alone and the pipeline plans for the default four queues)."""
    global _HW_QUEUES
    if "GPU_MAX_HW_QUEUES" in os.environ:
        try:
            _HW_QUEUES = max(1, int(os.environ["GPU_MAX_HW_QUEUES"]))
        except ValueError:
            _HW_QUEUES = 4
    elif not _runtime_initialised() and not torch.cuda.is_initialized():
        os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
        _HW_QUEUES = int(n)
    else:
        _HW_QUEUES = 4
    return _HW_QUEUES


if os.environ.get("EBOS_NO_QUEUE_ENV", "0") in ("", "0"):
    configure_queues(16)
elif "GPU_MAX_HW_QUEUES" in os.environ:
    configure_queues(0)  # (reads the variable; sets nothing)


def hw_queues() -> int:
    """Hardware queues this process's HIP streams are spread over (GPU_MAX_HW_QUEUES as the runtime saw or will see it; 4 when the
    runtime was already up before this package could ask for more)."""
    return _HW_QUEUES


class HipUnavailableError(RuntimeError):
    """The HIP extension (or a GPU to run it on) is missing.  Nothing falls back to the CPU."""


_P, _I, _L, _D, _Z, _F = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t, C.c_float

# name -> (restype, argtypes).  Mirrors include/ebos_hip.h one to one (tests/test_abi.py checks it).
_WARP = [_P, _P, _P, _I, _D, _I, _L, _L, _I, _I, _I, _P, _P, _P]
_WARP_BWD = [_P, _P, _I, _D, _I, _P, _L, _L, _I, _I, _I, _P, _P]
_W2 = [_P, _P, _P, _I, _D, _I, _P, _L, _P, _P]
_W2_BWD = [_P, _P, _I, _D, _I, _P, _P, _L, _P, _P]
_SPLAT = [_P, _P, _D, _I, _D, _L, _L, _I, _I, _I, _I, _P, _P]
_SPLAT_BWD = [_P, _P, _D, _D, _P, _L, _L, _I, _I, _I, _I, _P, _P, _P]
_SOA = [_P, _P, _I, _D, _I, _L, _P, _P, _P, _P, _P]
_VAR = [_P, _I, _I, _I, _I, _P, _P, _P, _Z, _P]
_VAR_GRAD = [_P, _I, _I, _I, _I, _P, _P, _P, _P]
_GM = [_P, _I, _I, _I, _I, _P, _P, _Z, _P]
_GM_GRAD = [_P, _I, _I, _I, _I, _P, _P, _P]
_GAUSS = [_P, _P, _L, _L, _L, _P, _I, _I, _P]
SIGNATURES = {
    "ebos_profile_start": (_I, [_I]),
    "ebos_profile_stop": (_I, [C.POINTER(C.c_float), _I]),
    "ebos_profile_start_kernel": (_I, [_I, _I]),
    "ebos_version": (_I, []),
    "ebos_last_error": (C.c_char_p, []),
    "ebos_build_info": (C.c_char_p, []),
    "ebos_time_range_f32": (_I, [_P, _L, _L, _P, _P]),
    "ebos_time_range_f64": (_I, [_P, _L, _L, _P, _P]),
    "ebos_warp_dense_f32": (_I, _WARP),
    "ebos_warp_dense_f64": (_I, _WARP),
    "ebos_warp_dense_bwd_f32": (_I, _WARP_BWD),
    "ebos_warp_dense_bwd_f64": (_I, _WARP_BWD),
    "ebos_warp_2dof_f32": (_I, _W2),
    "ebos_warp_2dof_f64": (_I, _W2),
    "ebos_warp_2dof_bwd_f32": (_I, _W2_BWD),
    "ebos_warp_2dof_bwd_f64": (_I, _W2_BWD),
    "ebos_splat_f32": (_I, _SPLAT),
    "ebos_splat_f64": (_I, _SPLAT),
    "ebos_splat_bwd_f32": (_I, _SPLAT_BWD),
    "ebos_splat_bwd_f64": (_I, _SPLAT_BWD),
    "ebos_events_to_soa_f32": (_I, _SOA),
    "ebos_events_to_soa_f64": (_I, _SOA),
    "ebos_raw_time_range": (_I, [_P, _I, _L, _D, _P, _P, _P]),
    "ebos_raw_events_to_soa": (_I, [_P, _P, _P, _I, _P, _D, _P, _I, _D, _I, _L, _P, _P, _P, _P, _P]),
    "ebos_bin_scratch_bytes": (_Z, [_L]),
    "ebos_bin_scratch_bytes_events": (_Z, [_L, _I, _I, _I, _I]),
    "ebos_bin_events_f32": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "ebos_plan_lean_scratch_bytes": (_Z, [_L, _I, _I, _I, _I]),
    "ebos_plan_lean": (_I, [_I, _P, _P, _P, _P, _D, _L, _I, _D, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P, _P, _P, _Z, _P]),
    "ebos_plan_compact_f32": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _L, _P]),
    "ebos_plan_compact_frac_f32": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    "ebos_iwe_dense_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P]),
    "ebos_iwe_dense_tiled_f32": (_I, [_P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ebos_tiled_config": (_I, [C.POINTER(C.c_int), _I]),
    "ebos_iwe_dense_bwd_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P]),
    "ebos_slab_config": (_I, [C.POINTER(C.c_int), _I]),
    "ebos_halo_auto": (_I, [_I, _D]),
    "ebos_iwe_slab_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "ebos_iwe_dense_slab_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _I, _I, _P, _P, _P, _P]),
    "ebos_iwe_2dof_slab_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _I, _I, _P, _P, _P, _P]),
    "ebos_iwe_2dof_slab_batch_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _I, _I, _P, _P, _P, _P, _P]),
    "ebos_plan_parts": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ebos_plan_facts": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ebos_iwe_2dof_tiled_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _Z, _P]),
    "ebos_iwe_dense_tiled_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _Z, _P, _P]),
    "ebos_variance_dense_job_f32": (_I, [_P, _P, _P, _P, _P, _P]),
    "ebos_gradient_magnitude_dense_job_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "ebos_variance_dense_job_signed_f32": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "ebos_gradient_magnitude_fused_partials": (_L, [_I, _I]),
    "ebos_gradient_magnitude_fused_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "ebos_iwe_2dof_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P]),
    "ebos_iwe_2dof_bwd_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    "ebos_cost_scratch_bytes": (_Z, [_I]),
    "ebos_image_variance_f32": (_I, _VAR),
    "ebos_image_variance_f64": (_I, _VAR),
    "ebos_image_variance_grad_f32": (_I, _VAR_GRAD),
    "ebos_image_variance_grad_f64": (_I, _VAR_GRAD),
    "ebos_image_variance_affine_f32": (_I, [_P, _P, _I, _P, _P]),
    "ebos_gradient_magnitude_f32": (_I, _GM),
    "ebos_gradient_magnitude_f64": (_I, _GM),
    "ebos_gradient_magnitude_grad_f32": (_I, _GM_GRAD),
    "ebos_gradient_magnitude_grad_f64": (_I, _GM_GRAD),
    "ebos_upsample_patch_flow_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ebos_upsample_bwd_scratch_bytes": (_Z, [_I, _I]),
    "ebos_upsample_patch_flow_bwd_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "ebos_upsample_patch_flow_bwd_adam_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _D, _D, _D, _D, _I, _P,
                                                   _P, _F, _P, _I, _P, _I, _P, _P]),
    "ebos_patch_fused_supported": (_I, [_I, _I, _I, _I, _I]),
    "ebos_iwe_slab_batch_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _Z, _I, _I, _P, _P]),
    "ebos_iwe_patch_slab_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _I, _I,
                                     _P, _P, _P, _P]),
    "ebos_iwe_patch_slab_frac_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P,
                                          _I, _I, _P, _P, _P, _P]),
    "ebos_iwe_patch_tiled_bwd_frac_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P,
                                               _P, _P, _P, _Z, _P, _F, _F, _P, _P, _L, _L, _P, _P, _F, _F, _P]),
    "ebos_patch_grad_partials_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "ebos_iwe_patch_tiled_bwd_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P,
                                          _P, _P, _P, _Z, _P, _F, _F, _P, _P, _L, _L, _P, _P, _P]),
    "ebos_iwe_dense_tiled_bwd_blur_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _Z, _P,
                                                _P, _L, _L, _P, _P, _F, _F, _P]),
    "ebos_iwe_patch_tiled_bwd_blur_f32": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P,
                                               _P, _Z, _P, _F, _F, _P, _P, _L, _L, _P, _P, _F, _F, _P]),
    "ebos_blur3_variance_partials": (_L, [_I, _I]),
    "ebos_cmax_cost_scratch_bytes": (_Z, [_I, _I]),
    "ebos_blur3_variance_adjoint_f32": (_I, [_P, _I, _I, _I, _F, _F, _P, _P, _L, _P]),
    "ebos_cmax_2dof_solve_f32": (_I, [_P, _I, _P]),
    "ebos_patch_grad_combine_adam_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _D, _D, _D, _D, _I, _P, _P,
                                              _F, _P, _I, _P, _I, _P, _P]),
    "ebos_flow_regularisers_partials": (_I, []),
    "ebos_flow_regularisers_f32": (_I, [_P, _I, _I, _F, _F, _P, _P, _P, _L, _L, _P, _P, _P]),
    "ebos_iwe_slab_partials": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "ebos_cmax_adam_step_f32": (_I, [_P, _P, _P, _P, _I, _D, _D, _D, _D, _P, _P, _F, _P, _I, _P, _I, _P]),
    "ebos_cmax_patch_solve_f32": (_I, [_P, _I, _P]),
    "ebos_cmax_patch_solve_many_f32": (_I, [_P, _P, _I, _I]),
    "ebos_cmax_resident_mailbox_bytes": (_Z, [_I, _I, _I, _I]),
    "ebos_cmax_resident_supported": (_I, [_P]),
    "ebos_cmax_patch_solve_resident_f32": (_I, [_P, _I, _P, _Z, _D, _P]),
    "ebos_cmax_2dof_resident_supported": (_I, [_P]),
    "ebos_cmax_2dof_solve_resident_f32": (_I, [_P, _I, _P, _Z, _D, _P]),
    "ebos_cmax_resident_status": (_I, [_P, _P]),
    "ebos_cmax_resident_iterations": (_I, [_P, _P]),
    "ebos_gauss1d_f32": (_I, _GAUSS),
    "ebos_gauss1d_f64": (_I, _GAUSS),
    "ebos_gauss1d_bwd_f32": (_I, _GAUSS),
    "ebos_gauss1d_bwd_f64": (_I, _GAUSS),
}



class CmaxPatchProblem(C.Structure):
    """``ebos_cmax_patch_problem`` of include/ebos_hip.h (same field order)."""
    _fields_ = ([(k, _P) for k in ("xs", "ys", "dts", "grp_offsets", "cpix", "cdt", "key_offsets")] + [("n", _L)] +
                [(k, _I) for k in ("H", "W", "tile_h", "tile_w", "halo", "pad_h", "pad_w", "omit_boundary", "splits")] +
                [("part_table", _P)] +
                [(k, _I) for k in ("gh", "gw", "patch_h", "patch_w", "slide_h", "slide_w")] +
                [(k, _F) for k in ("w_variance", "w_flow_norm", "w_image_gradient", "w_gradient_magnitude")] +
                [(k, _D) for k in ("lr", "beta1", "beta2", "eps")] +
                [(k, _P) for k in ("theta", "d_theta", "exp_avg", "exp_avg_sq", "step")] + [("steps_done", _I)] +
                [(k, _P) for k in ("dense", "d_dense", "d_reg", "iwe", "variance", "d_iwe", "cost_scratch")] +
                [("cost_scratch_bytes", _Z)] +
                [(k, _P) for k in ("moments", "upstream", "reg_partials", "upsample_scratch", "workspace")] +
                [("workspace_bytes", _Z), ("losses", _P), ("losses_cap", _I), ("theta_mask", _P), ("grad_partials", _P),
                 ("grad_partials_bytes", _Z), ("blur_k0", _F), ("blur_k1", _F), ("blur_image", _P), ("cfx", _P), ("cfy", _P)])


class Cmax2dofProblem(C.Structure):
    """``ebos_cmax_2dof_problem`` of include/ebos_hip.h (same field order)."""
    _fields_ = ([(k, _P) for k in ("xs", "ys", "dts", "grp_offsets", "cpix", "cdt", "cfx", "cfy", "key_offsets")] + [("n", _L)] +
                [(k, _I) for k in ("H", "W", "tile_h", "tile_w", "halo", "pad_h", "pad_w", "omit_boundary", "splits")] +
                [("part_table", _P), ("w_variance", _F), ("blur_k0", _F), ("blur_k1", _F)] +
                [(k, _D) for k in ("lr", "beta1", "beta2", "eps")] +
                [(k, _P) for k in ("theta", "d_theta", "exp_avg", "exp_avg_sq", "step")] + [("steps_done", _I)] +
                [(k, _P) for k in ("iwe", "blur_image", "variance", "moments", "upstream", "cost_scratch")] +
                [("cost_scratch_bytes", _Z), ("workspace", _P), ("workspace_bytes", _Z), ("losses", _P), ("losses_cap", _I)])


class DenseJob(C.Structure):
    """``ebos_dense_job`` of include/ebos_hip.h (same field order)."""
    _fields_ = ([(k, _P) for k in ("xs", "ys", "dts", "grp_offsets", "cpix", "cdt", "key_offsets")] + [("n", _L)] +
                [(k, _I) for k in ("H", "W", "tile_h", "tile_w", "halo", "splits", "pad_h", "pad_w", "omit_boundary")] +
                [("workspace", _P), ("workspace_bytes", _Z), ("part_table", _P), ("iwe", _P), ("moments", _P)])


class SlabWindow(C.Structure):
    """``ebos_slab_window`` of include/ebos_hip.h (same field order): one window of ``ebos_iwe_slab_batch_f32``."""
    _fields_ = [(k, _P) for k in ("grp_offsets", "cpix", "cdt", "key_offsets", "part_table", "flow", "workspace", "iwe",
                                  "out_variance", "moments")]


_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libebos_hip.so (no GPU needed for loading) and attach the prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("EBOS_HIP_LIBRARY", LIB_PATH)
    if not os.path.exists(p):
        raise HipUnavailableError(
            f"libebos_hip.so not found at {p}. Build it with `python -m event_based_bos_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise HipUnavailableError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipUnavailableError(f"{p} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ebos_version() != ABI_VERSION:
        raise HipUnavailableError(f"{p}: ABI version {lib.ebos_version()} != expected {ABI_VERSION}; rebuild")
    if path is None:
        _lib = lib
    return lib


_gpu_checked = False


def require_gpu() -> C.CDLL:
    """Library + a visible GPU, or HipUnavailableError.  (Called on every operator launch: the positive answer of
    ``torch.cuda.is_available()`` -- ~3 us of environment look-ups per call -- is remembered.)"""
    global _gpu_checked
    if _gpu_checked and _lib is not None:
        return _lib
    lib = load_library()
    if not torch.cuda.is_available():
        raise HipUnavailableError(
            "no GPU visible to PyTorch-ROCm: event_based_bos_amd runs its warp/IWE/cost path on an MI355X "
            "through libebos_hip.so only; there is no CPU fallback.")
    _gpu_checked = True
    return lib


# The raw getters are unversioned torch internals: resolved ONCE, with the public API as the fall-back of a torch build
# that renames them -- the fall-back builds a Stream object per call (~5-10 us), it is never wrong.
_raw_device = getattr(torch._C, "_cuda_getDevice", None) or (lambda: torch.cuda.current_device())
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def current_device_index() -> int:
    return _raw_device()


def stream_ptr(device: Optional[torch.device] = None) -> int:
    """hipStream_t of torch's current stream on ``device`` (default: the current device) as an integer.  The raw getter:
    ``torch.cuda.current_stream()`` builds a Stream object per call (~5-10 us of the ~20 us an operator launch costs)."""
    idx = _raw_device() if device is None or device.index is None else device.index
    return _raw_stream(idx)


class on_device(object):
    """``with on_device(dev):`` -- like ``torch.cuda.device(dev)`` but free when ``dev`` is already current (the usual
    case; the torch context manager costs ~10 us per operator launch either way)."""
    __slots__ = ("idx", "ctx")

    def __init__(self, device):
        self.idx = device.index if isinstance(device, torch.device) else device
        self.ctx = None

    def __enter__(self):
        if self.idx is not None and _raw_device() != self.idx:
            self.ctx = torch.cuda.device(self.idx)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False


def check(rc: int, what: str) -> None:
    if rc != EBOS_OK:
        msg = load_library().ebos_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with ebos_status {rc}: {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def suffix(dtype: torch.dtype) -> str:
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise TypeError(f"event_based_bos_amd kernels exist for float32 and float64 tensors, got {dtype}")


def slab_configs():
    lib = load_library()
    n = lib.ebos_slab_config(None, 0)
    buf = (C.c_int * (3 * n))()
    lib.ebos_slab_config(buf, n)
    return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]


def tiled_configs():
    lib = load_library()
    n = lib.ebos_tiled_config(None, 0)
    buf = (C.c_int * (3 * n))()
    lib.ebos_tiled_config(buf, n)
    return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]
