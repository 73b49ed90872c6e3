#!/usr/bin/env python3
"""Time one conv layer's forward / dW+dX through the C-ABI (per-class HIP-event timers).
usage: [PROF_BN=1] prof_layer.py N C H W F K S P [iters]   (PROF_BN: forward is conv + batch-norm + ReLU, the ResNet form)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
n, c, h, w, f, k, s, p = (int(v) for v in sys.argv[1:9])
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 5
dev = "cuda:0"
oh, ow = ops.conv_out_hw(h, w, k, s, p)
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = (torch.rand((f, c, k, k), device=dev) * 2 - 1) * (3.0 / (c * k * k)) ** 0.5
bias = torch.rand(f, device=dev) * 0.1
y = torch.empty((n, f, oh, ow), device=dev)
dy = (torch.rand((n, f, oh, ow), device=dev) * 2 - 1) * 1e-2
dx = torch.empty_like(x); dw = torch.zeros_like(wt); db = torch.zeros_like(bias)
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, k, s, p, 1)), device=dev)
bn = None
if os.environ.get("PROF_BN") == "1":
    Z = lambda: torch.zeros(f, device=dev)
    bn = dict(run_mean=Z(), run_var=Z() + 1, scales=Z() + 1, saved_mean=Z(), saved_var=Z(), workspace=torch.empty_like(y))
torch.cuda.synchronize()
def run():
    ops.conv_forward(x, wt, bias, y, k, s, p, 1, 2, bn=bn)
    ops.conv_backward(x, wt, y, dy, dx, dw, db, k, s, p, 1, 0, ws)
run(); L.bcnn_hip_sync()
L.bcnn_hip_profile_reset(); L.bcnn_hip_profile_enable(1)
for _ in range(iters): run()
L.bcnn_hip_sync(); L.bcnn_hip_profile_enable(0)
for cls in range(L.bcnn_hip_profile_num_classes()):
    ms, cnt, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
    L.bcnn_hip_profile_read(cls, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
    if cnt.value:
        print("%-10s %8.3f ms/launch  %7.2f TFLOP/s  %7.1f GB/s" % (L.bcnn_hip_profile_class_name(cls).decode(),
              ms.value / cnt.value, fl.value / ms.value / 1e9, by.value / ms.value / 1e6))
