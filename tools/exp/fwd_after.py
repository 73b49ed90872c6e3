#!/usr/bin/env python3
"""configs[1] forward timed alone (HIP events around it) when the kernel BEFORE it was: another forward, the dW pass, a
read-only sweep of a 1.6 GB tensor, a fill of a 1.6 GB tensor. Why the forward is slower inside the fwd / dW alternation
of the bench than in a loop of its own. BCNN_HIP_LIB selects a variant library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
n, c, h, w, f = 128, 3, 224, 224, 64
dev = "cuda:0"
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = (torch.rand((f, c, 3, 3), device=dev) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
bias = torch.rand(f, device=dev) * 0.1
dw = torch.zeros_like(wt); db = torch.zeros_like(bias)
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device=dev)
y = torch.empty((n, f, h, w), device=dev)
dy = torch.rand((n, f, h, w), device=dev) * 1e-2
e = [L.bcnn_hip_event_create() for _ in range(4)]
fwd = lambda: ops.conv_forward(x, wt, bias, y, 3, 1, 1, 1, 0)
befores = {
    "forward": fwd,
    "dW": lambda: ops.conv_backward(x, wt, y, dy, None, dw, db, 3, 1, 1, 1, 0, ws),
    "read 1.6 GB": lambda: L.bcnn_hip_grad_bias(db.data_ptr(), dy.data_ptr(), n, f, h * w),
    "fill 1.6 GB": lambda: dy.fill_(0.001),
    "read x": lambda: L.bcnn_hip_grad_bias(db.data_ptr(), x.data_ptr(), n, c, h * w),
}
for name, before in befores.items():
    tb = tf = 0.0
    reps = 12
    for r in range(reps + 2):
        L.bcnn_hip_event_record(e[0]); before(); L.bcnn_hip_event_record(e[1]); fwd(); L.bcnn_hip_event_record(e[2])
        L.bcnn_hip_event_sync(e[2])
        if r >= 2:
            tb += L.bcnn_hip_event_elapsed_ms(e[0], e[1]); tf += L.bcnn_hip_event_elapsed_ms(e[1], e[2])
    print("%s after [%s]: forward %.3f ms (the kernel before: %.3f ms)" % (os.path.basename(os.environ.get("BCNN_HIP_LIB", "product")), name, tf / reps, tb / reps), flush=True)
