#!/usr/bin/env python3
"""configs[1] forward / dW alternation with an idle gap (a sleeping kernel) between the two: how much of the pair's time is
the write drain of the forward pass meeting the dW pass's reads? Each kernel timed by its own HIP events."""
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
fwd = lambda: ops.conv_forward(x, wt, bias, y, 3, 1, 1, 1, 0)
bwd = lambda: ops.conv_backward(x, wt, y, dy, None, dw, db, 3, 1, 1, 1, 0, ws)
ev = [[L.bcnn_hip_event_create() for _ in range(4)] for _ in range(24)]
tiny = torch.zeros(64, device=dev)
import time
for where in ["after forward"] * 8 + ["sleep"] + ["after forward"] * 4:
  if where == "sleep":
      L.bcnn_hip_sync(); time.sleep(2.0); print("host slept 2 s"); continue
  for cycles in (0,):
    def gap():
        if cycles == 1: tiny.add_(1.0)           # a launch boundary without any waiting
        elif cycles: torch.cuda._sleep(cycles)
    L.bcnn_hip_sync()
    t0 = L.bcnn_hip_event_create(); t1 = L.bcnn_hip_event_create()
    L.bcnn_hip_event_record(t0)
    for r in range(24):
        e = ev[r]
        L.bcnn_hip_event_record(e[0]); fwd(); L.bcnn_hip_event_record(e[1])
        if where != "after dW": gap()
        L.bcnn_hip_event_record(e[2]); bwd(); L.bcnn_hip_event_record(e[3])
        if where != "after forward": gap()
    L.bcnn_hip_event_record(t1)
    L.bcnn_hip_event_sync(t1)
    tf = sum(L.bcnn_hip_event_elapsed_ms(e[0], e[1]) for e in ev[4:]) / 20
    tg = sum(L.bcnn_hip_event_elapsed_ms(e[1], e[2]) for e in ev[4:]) / 20
    tb = sum(L.bcnn_hip_event_elapsed_ms(e[2], e[3]) for e in ev[4:]) / 20
    print("gap %-13s %.3f ms: forward %.3f ms, dW %.3f ms, sum %.3f | whole pair incl. gaps %.3f" % (where, tg if where != "after dW" else -1, tf, tb, tf + tb, L.bcnn_hip_event_elapsed_ms(t0, t1) / 24), flush=True)
