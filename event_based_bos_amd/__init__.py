"""event_based_bos_amd -- MI355X-native contrast-maximisation inner loop for event-based BOS.

One hot path, built from scratch for gfx950 / CDNA4 behind the plugin surface of
tub-rip/event_based_bos:

    warp events (dense flow | 2-DoF)  ->  bilinear-splat image of warped events  ->  contrast cost (+ gradients)

    Warp                 drop-in for src/warp.py
    EventImageConverter  drop-in for src/event_image_converter.py
    costs                drop-in for src/costs (+ image_variance, gradient_magnitude)
    EventPlan            device-resident SoA event window + the fused warp/IWE kernels
    SlabBatch            several independent windows per launch (ebos_iwe_slab_batch_f32)
    solver               contrast-maximisation solver behind the reference's solver registry
    data_loader          raw-column event store (the CCS raw_events layout) feeding EventPlan.build_raw

All arithmetic of the path runs in hand-written HIP kernels reached through the C ABI of
libebos_hip.so (include/ebos_hip.h).  There is no CPU fallback: without the library or a GPU the
operators raise ``HipUnavailableError``.
"""
from ._hip import HipUnavailableError, load_library  # noqa: F401
from .warp import MotionModelKeyError, Warp  # noqa: F401
from .event_image_converter import EventImageConverter  # noqa: F401
from .event_plan import EventPlan, SlabBatch  # noqa: F401
from . import costs, data_loader, fusion, ops, solver, types, utils  # noqa: F401

__version__ = "0.1.0"
