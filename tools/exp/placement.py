#!/usr/bin/env python3
"""Does the physical placement of the 1.6 GB result / gradient tensors change the configs[1] kernels' time?
Allocates K candidate tensors and times the forward (writes y) and the dW pass (reads dy) on each.
usage: placement.py [K]; BCNN_HIP_LIB selects a variant library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n, c, h, w, f = 128, 3, 224, 224, 64
dev = "cuda:0"
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = (torch.rand((f, c, 3, 3), device=dev) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
bias = torch.rand(f, device=dev) * 0.1
dw = torch.zeros_like(wt); db = torch.zeros_like(bias)
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device=dev)
e0, e1 = L.bcnn_hip_event_create(), L.bcnn_hip_event_create()
def timeit(fn, reps=8):
    fn(); fn(); L.bcnn_hip_sync()
    L.bcnn_hip_event_record(e0)
    for _ in range(reps): fn()
    L.bcnn_hip_event_record(e1); L.bcnn_hip_event_sync(e1)
    return L.bcnn_hip_event_elapsed_ms(e0, e1) / reps
bufs = [torch.empty((n, f, h, w), device=dev) for _ in range(K)]
for i, y in enumerate(bufs):
    y.uniform_(-0.01, 0.01)
    tf = timeit(lambda: ops.conv_forward(x, wt, bias, y, 3, 1, 1, 1, 0))
    modes = []
    for dyn, grid in [(1, int(g)) for g in os.environ.get('GRIDS', '256,384,512,768,1024').split(',')]:   # experiment library only
        os.environ["BCNN_HIP_WINDOW_DYN"] = str(dyn); os.environ["BCNN_HIP_WINDOW_GRID"] = str(grid)
        modes.append("%s%d %.3f" % ("dyn" if dyn else "static", grid, timeit(lambda: ops.conv_forward(x, wt, bias, y, 3, 1, 1, 1, 0))))
    os.environ.pop("BCNN_HIP_WINDOW_DYN"); os.environ.pop("BCNN_HIP_WINDOW_GRID")
    print("   ", " | ".join(modes))
    tz = timeit(lambda: y.zero_())
    tb = timeit(lambda: ops.conv_backward(x, wt, y, y, None, dw, db, 3, 1, 1, 1, 0, ws))
    ts = timeit(lambda: L.bcnn_hip_grad_bias(db.data_ptr(), y.data_ptr(), n, f, h * w))
    print("buf %2d %#x: forward %.3f ms  memset %.3f  dW %.3f  channel-sum(read) %.3f" % (i, y.data_ptr(), tf, tz, tb, ts), flush=True)
