#!/usr/bin/env python3
"""Small driver for rocprofv3: runs the configs[1] conv forward and backward a few times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import ops

n, c, h, w, f, k, s, p = 128, 3, 224, 224, 64, 3, 1, 1
dev = "cuda:0"
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = (torch.rand((f, c, k, k), device=dev) * 2 - 1) * 0.33
bias = torch.rand(f, device=dev) * 0.1
y = torch.empty((n, f, h, w), device=dev)
dy = (torch.rand((n, f, h, w), device=dev) * 2 - 1) * 1e-2
dw = torch.zeros_like(wt); db = torch.zeros_like(bias)
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, k, s, p, 1)), device=dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(iters):
    ops.conv_forward(x, wt, bias, y, k, s, p, 1, 0)
    ops.conv_backward(x, wt, y, dy, None, dw, db, k, s, p, 1, 0, ws)
torch.cuda.synchronize()
print("done")
