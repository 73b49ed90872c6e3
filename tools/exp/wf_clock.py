#!/usr/bin/env python3
"""Cycle stamps of one steady-state block of the fused Winograd kernel (variant library built with -DWF_ABL_CLOCK):
prologue, every chunk's (requests | MFMAs | transform + wait + barrier) and the epilogue, for one wave of each half.
usage: BCNN_HIP_LIB=tools/exp/lib_wf_clock.so BCNN_HIP_WINOGRAD_FUSED=1 wf_clock.py N C H W F"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
n, c, h, w, f = (int(v) for v in sys.argv[1:6])
dev = "cuda:0"
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = (torch.rand((f, c, 3, 3), device=dev) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
y = torch.empty((n, f, h, w), device=dev)
dy = torch.rand_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(wt); db = torch.zeros(f, device=dev)
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device=dev)
for _ in range(3):
    ops.conv_backward(x, wt, y, dy, dx, dw, db, 3, 1, 1, 1, 0, ws)
L.bcnn_hip_sync()
buf = (C.c_ulonglong * 384)()
lib = C.CDLL(os.environ["BCNN_HIP_LIB"])
lib.bcnn_hip_debug_read_wf_clock(buf)
for half, name in ((0, "wave 0 (transforms first)"), (4, "wave 4 (multiplies first)")):
    t = [buf[half * 48 + i] for i in range(40)]
    print(name)
    print("  wait for previous epilogue readers %6d | prologue (decode, DMA, patch loads, transform) %6d | wait + barrier %6d"
          % (t[1] - t[0], t[2] - t[1], t[3] - t[2]))
    for kc in range(8):
        a, b, cdone = t[4 + 3 * kc], t[5 + 3 * kc], t[6 + 3 * kc]
        nxt = t[4 + 3 * (kc + 1)] if kc < 7 else t[28]
        print("  chunk %d: head (transform / requests) %5d | 32 MFMAs %5d | tail (transform / loads, wait, barrier) %5d" % (kc, b - a, cdone - b, nxt - cdone))
    print("  epilogue: column half + S to LDS %5d | barrier %5d | row half, stores, stats %6d | block total %6d"
          % (t[29] - t[28], t[30] - t[29], t[31] - t[30], t[31] - t[0]))
    for kc in (3, 4):
        b = t[5 + 3 * kc]
        print("  chunk %d k-steps (8 MFMAs each), cycles since the chunk's head stamp: " % kc
              + " ".join("%5d" % (t[32 + 4 * (kc - 3) + ks] - t[4 + 3 * kc]) for ks in range(4))
              + "  (MFMAs start at %d)" % (b - t[4 + 3 * kc]))
base = buf[4 + 9]
print("all waves, chunk 3, cycles vs wave 0's head stamp: head | mfma start | k-step ends | mfma end | next head")
for w in range(8):
    t = [buf[w * 48 + i] for i in range(48)]
    print("  wave %d: %5d | %5d | %s | %5d | %5d" % (w, t[13] - base, t[14] - base, " ".join("%5d" % (t[32 + k] - base) for k in range(4)),
                                                  t[15] - base, t[16] - base))
print("produce step of chunk 3, cycles vs wave 0's head stamp: start | V written | patches requested | U DMA issued | (patches of this chunk had arrived)")
for w in range(8):
    t = [buf[w * 48 + i] for i in range(48)]
    print("  wave %d: %s" % (w, " ".join("%5d" % (t[40 + k] - base) for k in (0, 1, 2, 3, 4))))
