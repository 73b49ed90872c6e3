#!/usr/bin/env python3
"""HBM ceilings on this box for the access patterns the hot path uses (write-only, read-only, copy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcnn_amd import _lib
L = _lib.load()
n = 128 * 64 * 224 * 224  # 1.644 GB of fp32, the configs[1] output size
a = torch.empty(n, device="cuda:0"); b = torch.empty(n, device="cuda:0")
db = torch.zeros(64, device="cuda:0")
e0, e1 = L.bcnn_hip_event_create(), L.bcnn_hip_event_create()
def timeit(fn, reps=10):
    fn(); L.bcnn_hip_sync()
    L.bcnn_hip_event_record(e0)
    for _ in range(reps): fn()
    L.bcnn_hip_event_record(e1); L.bcnn_hip_event_sync(e1)
    return L.bcnn_hip_event_elapsed_ms(e0, e1) / reps
t = timeit(lambda: L.bcnn_hip_fill_f32(a.data_ptr(), n, 1.5)); print("fill (16B stores)   %.3f ms  %.0f GB/s write" % (t, 4*n/t/1e6))
t = timeit(lambda: L.bcnn_hip_fill_f32(a.data_ptr(), n, 0.0)); print("memset             %.3f ms  %.0f GB/s write" % (t, 4*n/t/1e6))
t = timeit(lambda: L.bcnn_hip_memcpy_d2d(b.data_ptr(), a.data_ptr(), 4*n)); print("memcpy d2d         %.3f ms  %.0f GB/s r+w" % (t, 8*n/t/1e6))
t = timeit(lambda: L.bcnn_hip_grad_bias(db.data_ptr(), a.data_ptr(), 128, 64, 224*224)); print("channel sum (read) %.3f ms  %.0f GB/s read" % (t, 4*n/t/1e6))
t = timeit(lambda: L.bcnn_hip_axpy(n, 0.5, a.data_ptr(), b.data_ptr())); print("axpy (2r+1w)       %.3f ms  %.0f GB/s" % (t, 12*n/t/1e6))
