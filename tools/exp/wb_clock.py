#!/usr/bin/env python3
"""Cycle stamps of one steady-state block of the split-bf16 fused Winograd kernel (library built with -DWF_ABL_CLOCK):
usage: BCNN_HIP_LIB=tools/exp/lib_wf_clock.so BCNN_HIP_WINOGRAD_FUSED=1 BCNN_HIP_WINOGRAD_BF16=2 wb_clock.py N C H W F"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
n, c, h, w, f = (int(v) for v in sys.argv[1:6])
x = torch.rand((n, c, h, w), device="cuda") * 2 - 1
wt = (torch.rand((f, c, 3, 3), device="cuda") * 2 - 1) * (3.0 / (c * 9)) ** 0.5
y = torch.empty((n, f, h, w), device="cuda")
dy = torch.rand_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(wt); db = torch.zeros(f, device="cuda")
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device="cuda")
for _ in range(3):
    ops.conv_backward(x, wt, y, dy, dx, dw, db, 3, 1, 1, 1, 0, ws)
L.bcnn_hip_sync()
buf = (C.c_ulonglong * 384)()
C.CDLL(os.environ["BCNN_HIP_LIB"]).bcnn_hip_debug_read_wf_clock(buf)
base = buf[0]
print("cycles since the unit's start (wave 0); per chunk: phase A start | end | after barrier | MFMAs done")
for wv in range(8):
    t = [buf[wv * 48 + i] for i in range(48)]
    row = []
    for kc in range(4):
        row.append(" ".join("%6d" % (t[4 + 4 * kc + i] - base) for i in range(4)))
    print("  wave %d: start %5d || %s || epilogue %6d .. %6d" % (wv, t[0] - base, " | ".join(row), t[28] - base, t[31] - base))
