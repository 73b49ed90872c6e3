#!/usr/bin/env python3
"""Stand-alone batch-norm forward / backward (TRAIN) on the activation shapes of the benchmarks, for rocprofv3
kernel traces. usage: prof_bn.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda:0"
shapes = [(128, 64, 112), (128, 64, 56), (128, 128, 28), (128, 256, 14), (128, 512, 7),
          (256, 32, 112), (256, 64, 56), (256, 128, 56), (256, 256, 28), (256, 512, 14), (256, 1024, 7)]
for (n, c, hw) in shapes:
    x = torch.rand((n, c, hw, hw), device=dev) * 2 - 1
    y = torch.empty_like(x); ws = torch.empty_like(x)
    rm = torch.zeros(c, device=dev); rv = torch.ones(c, device=dev); sc = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    sm = torch.zeros(c, device=dev); sv = torch.zeros(c, device=dev)
    dy = torch.rand_like(x); dx = torch.empty_like(x)
    dsc = torch.zeros(c, device=dev); db = torch.zeros(c, device=dev); dm = torch.zeros(c, device=dev); dv = torch.zeros(c, device=dev)
    for _ in range(iters):
        ops.batchnorm_forward(x, y, rm, rv, sc, b, sm, sv, ws, 1)
        ops.batchnorm_backward(dy, dx, sc, dsc, db, sm, sv, dm, dv, ws)
    L.bcnn_hip_sync()
    mb = n * c * hw * hw * 4 / 1e6
    print("shape", n, c, hw, "tensor MB %.1f" % mb, flush=True)
