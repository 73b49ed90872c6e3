#!/usr/bin/env python3
"""One depthwise layer forward + backward through the C-ABI (for rocprofv3 / PMC runs).
usage: prof_dw.py N C H W K S P [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
n, c, h, w, k, s, p = (int(v) for v in sys.argv[1:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
dev = "cuda:0"
oh, ow = ops.conv_out_hw(h, w, k, s, p)
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = torch.rand((c, k, k), device=dev) - 0.5
bias = torch.rand(c, device=dev) * 0.1
y = torch.empty((n, c, oh, ow), device=dev)
dy = torch.rand((n, c, oh, ow), device=dev) * 0.01
dx = torch.zeros_like(x); dw = torch.zeros_like(wt); db = torch.zeros_like(bias)
torch.cuda.synchronize()
for _ in range(iters):
    ops.depthwise_forward(x, wt, bias, y, k, s, p, 2)
    ops.depthwise_backward(x, wt, y, dy, dx, dw, db, k, s, p, 2)
L.bcnn_hip_sync()
print("ok", oh, ow)
