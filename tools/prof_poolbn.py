#!/usr/bin/env python3
"""batch-norm apply + 3x3/s2 max-pooling (two kernels) against the pooling kernel that normalises on the fly, on the
ResNet-18 stem shape; also checks that values and indexes are identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import _lib
L = _lib.load()
dev = "cuda:0"
n, c, h, w = 128, 64, 112, 112
oh, ow = 56, 56
x = torch.randn((n, c, h, w), device=dev)
mean, var = torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5
sc, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
y = torch.empty_like(x)
p1, p2 = torch.empty((n, c, oh, ow), device=dev), torch.empty((n, c, oh, ow), device=dev)
i1, i2 = torch.empty((n, c, oh, ow), device=dev, dtype=torch.int32), torch.empty((n, c, oh, ow), device=dev, dtype=torch.int32)
P = lambda t: t.data_ptr()
def two():
    L.bcnn_hip_batchnorm_apply(P(x), P(y), P(sc), P(b), P(mean), P(var), n, c, h * w, 2)
    L.bcnn_hip_maxpool_forward(P(y), P(p1), P(i1), n, c, h, w, oh, ow, 3, 2)
def one():
    L.bcnn_hip_maxpool_forward_bn(P(x), P(p2), P(i2), n, c, h, w, oh, ow, 3, 2, P(sc), P(b), P(mean), P(var), 2)
for fn in (two, one):
    for _ in range(3): fn()
    L.bcnn_hip_sync(); t0 = time.perf_counter()
    for _ in range(20): fn()
    L.bcnn_hip_sync(); print(fn.__name__, "%.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
print("equal values", torch.equal(p1, p2), "equal indexes", torch.equal(i1, i2))
