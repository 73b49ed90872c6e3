#!/usr/bin/env python3
"""Time the LDS-staged depthwise kernels on MobileNet-v1's layer shapes (N=256) through the C-ABI: forward with the
batch-norm statistics epilogue, and the backward that applies the following batch-norm node's backward on the fly.
Prints microseconds per call and the algorithmic TB/s (forward: x + y; backward: dz + y + x + dx).
usage: prof_dw.py [iters]; BCNN_HIP_LIB selects a variant library (tools/exp/variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
N = int(os.environ.get("DW_N", "256"))
LAYERS = [(32, 112, 1), (64, 112, 2), (128, 56, 1), (128, 56, 2), (256, 28, 1), (256, 28, 2), (512, 14, 1), (512, 14, 2),
          (1024, 7, 1)]
tf = tb = 0.0
MULT = {(512, 14, 1): 5}
for c, hw, s in LAYERS:
    oh = (hw + 2 - 3) // s + 1
    x = torch.rand((N, c, hw, hw), device=dev) * 2 - 1
    wt = torch.rand(c * 9, device=dev) - 0.5
    bias = torch.rand(c, device=dev) * 0.1
    y = torch.empty((N, c, oh, oh), device=dev)
    dz = torch.rand((N, c, oh, oh), device=dev) - 0.5
    dx = torch.empty_like(x)
    dw, db = torch.zeros_like(wt), torch.zeros_like(bias)
    mean, var = torch.rand(c, device=dev), torch.rand(c, device=dev) + 0.5
    sc, dm, dv = torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev), torch.rand(c, device=dev)
    stats = torch.empty(max(1, ops.depthwise_stats_size(N, c, hw, hw, 3, s, 1)), device=dev)
    def fwd():
        ops.depthwise_forward_stats(x, wt, bias, y, 3, s, 1, 2, stats)
    def bwd():
        ops.depthwise_backward_bn(x, wt, y, dz, dx, dw, db, 3, s, 1, 2, True, mean, var, sc, dm, dv)
    res = []
    for fn, nbytes in ((fwd, 4 * (x.numel() + y.numel())), (bwd, 4 * (2 * y.numel() + 2 * x.numel()))):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        L.bcnn_hip_sync()
        import time
        t0 = time.perf_counter()
        for _ in range(iters): fn()
        L.bcnn_hip_sync()
        us = (time.perf_counter() - t0) / iters * 1e6
        res.append((us, nbytes / us / 1e6))
    tf += res[0][0] * MULT.get((c, hw, s), 1); tb += res[1][0] * MULT.get((c, hw, s), 1)
    print("c%-4d %3dx%-3d s%d  fwd %7.1f us %5.2f TB/s   bwd %7.1f us %5.2f TB/s" % (c, hw, hw, s, res[0][0], res[0][1], res[1][0], res[1][1]))
print("%s: MobileNet sum (512x14 s1 x5): fwd %.0f us  bwd %.0f us" % (os.path.basename(os.environ.get("BCNN_HIP_LIB", "product")), tf, tb))
