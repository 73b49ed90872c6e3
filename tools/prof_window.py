#!/usr/bin/env python3
"""Time the configs[1]-shaped convolution (3x3 / s1, few input channels) forward and dW separately through the C-ABI
(per-class HIP-event timers of the library). usage: prof_window.py [N C H W F [iters]]; BCNN_HIP_LIB selects a variant."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
a = [int(v) for v in sys.argv[1:]]
n, c, h, w, f = (a + [128, 3, 224, 224, 64][len(a):])[:5]
iters = a[5] if len(a) > 5 else 20
dev = "cuda:0"
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = (torch.rand((f, c, 3, 3), device=dev) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
bias = torch.rand(f, device=dev) * 0.1
y = torch.empty((n, f, h, w), device=dev)
dy = (torch.rand((n, f, h, w), device=dev) * 2 - 1) * 1e-2
dw = torch.zeros_like(wt); db = torch.zeros_like(bias)
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device=dev)
def run():
    ops.conv_forward(x, wt, bias, y, 3, 1, 1, 1, 0)
    ops.conv_backward(x, wt, y, dy, None, dw, db, 3, 1, 1, 1, 0, ws)
for _ in range(int(os.environ.get('WARM', '100'))): run()   # the memory / fabric clocks need ~40 ms of load after an idle phase
L.bcnn_hip_sync()
L.bcnn_hip_profile_reset(); L.bcnn_hip_profile_enable(1)
for _ in range(iters): run()
L.bcnn_hip_sync(); L.bcnn_hip_profile_enable(0)
out = []
for cls in range(L.bcnn_hip_profile_num_classes()):
    ms, cnt, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
    L.bcnn_hip_profile_read(cls, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
    if cnt.value:
        out.append("%s %.4f ms %.0f GB/s" % (L.bcnn_hip_profile_class_name(cls).decode(), ms.value / cnt.value,
                                              by.value / ms.value / 1e6))
print(os.path.basename(os.environ.get("BCNN_HIP_LIB", "product")), " | ".join(out))
