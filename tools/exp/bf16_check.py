#!/usr/bin/env python3
"""Accuracy of the fused Winograd forward / dX kernels against float64 (torch on the GPU), exact-fp32 form vs the split-bf16
form: run with BCNN_HIP_LIB=<experiment library> BCNN_HIP_WINOGRAD_BF16=0|2|3 bf16_check.py N C H W F"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bcnn_amd import ops
n, c, h, w, f = (int(v) for v in sys.argv[1:6])
g = torch.Generator(device="cpu").manual_seed(3)
x = (torch.rand((n, c, h, w), generator=g) * 2 - 1).cuda()
wt = ((torch.rand((f, c, 3, 3), generator=g) * 2 - 1) * (3.0 / (c * 9)) ** 0.5).cuda()
b = torch.zeros(f, device="cuda")
y = torch.empty((n, f, h, w), device="cuda")
ops.conv_forward(x, wt, b, y, 3, 1, 1, 1, 0)
dy = ((torch.rand((n, f, h, w), generator=g) * 2 - 1) * 1e-2).cuda()
dx = torch.empty_like(x); dw = torch.zeros_like(wt); db = torch.zeros_like(b)
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device="cuda")
ops.conv_backward(x, wt, y, dy.clone(), dx, dw, db, 3, 1, 1, 1, 0, ws)
torch.cuda.synchronize()
x64, w64, dy64 = x.double(), wt.double(), dy.double()
y64 = torch.nn.functional.conv2d(x64, w64, padding=1)
dx64 = torch.nn.grad.conv2d_input(x64.shape, w64, dy64, padding=1)
rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
print("BF16=%s  N C H W F = %s : forward %.2e  dX %.2e  (max|a-b| / max|b| vs float64)" % (
    os.environ.get("BCNN_HIP_WINOGRAD_BF16", "-"), " ".join(sys.argv[1:6]), rel(y, y64), rel(dx, dx64)))
