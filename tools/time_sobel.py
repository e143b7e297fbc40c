"""Kernel-only timing of the fused Sobel pass (run under rocprofv3 --kernel-trace --stats; the host loop is launch-bound)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_based_bos_amd import _hip
lib = _hip.require_gpu()
h, w = 720, 1280
img = torch.rand(h, w, device='cuda') * 20
d = torch.empty_like(img)
n = int(lib.ebos_gradient_magnitude_fused_partials(h, w))
part = torch.empty(n, dtype=torch.float64, device='cuda')
s = _hip.stream_ptr()
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(400):
    lib.ebos_gradient_magnitude_fused_f32(img.data_ptr(), h, w, dbg << 8, None, None, d.data_ptr(), part.data_ptr(), n, s)
torch.cuda.synchronize()
