"""bcnn_hip_conv_prepack: the filter banks of several layers re-arranged in one launch at the start of a pass
(include/bcnn_hip.h) instead of once per bcnn_hip_conv_forward / _backward call. The copies are the ones those calls
would make themselves, so outputs must be bit-identical with and without it; a copy is used at most once and a later
prepack call discards the unused ones, so rewritten weights are never read through a stale copy."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Desc(C.Structure):
    _fields_ = [("w_d", C.c_void_p)] + [(k, C.c_int) for k in ("n", "c", "h", "w", "f", "k", "stride", "pad", "groups")]


# (n, c, hw, f, k, stride, pad): fused Winograd (64 -> 64 with 8192 tiles), its ragged-channel form, the LDS-DMA GEMM as
# 1x1, 3x3 / s2 (stride-parity classes in the data gradient), 3x3 / s1 with too few tiles for Winograd, and a layer with
# few input channels that takes no copy at all
LAYERS = [(32, 64, 32, 64, 3, 1, 1), (32, 72, 32, 96, 3, 1, 1), (4, 64, 14, 128, 1, 1, 0), (4, 64, 28, 128, 3, 2, 1),
          (2, 48, 12, 80, 3, 1, 1), (2, 3, 32, 64, 3, 1, 1)]


def _make(rs, layer):
    n, c, hw, f, k, st, pad = layer
    T = lambda *sh: torch.from_numpy(rs.uniform(-1, 1, sh).astype(np.float32)).to(DEV)
    oh = (hw + 2 * pad - k) // st + 1
    return dict(layer=layer, x=T(n, c, hw, hw), w=T(f, c, k, k) * 0.1, b=T(f) * 0.1, dy=T(n, f, oh, oh) * 0.01, oh=oh)


def _descs(items):
    arr = (Desc * len(items))()
    for d, it in zip(arr, items):
        n, c, hw, f, k, st, pad = it["layer"]
        d.w_d = it["w"].data_ptr()
        d.n, d.c, d.h, d.w, d.f, d.k, d.stride, d.pad, d.groups = n, c, hw, hw, f, k, st, pad, 1
    return arr


def _forward(ops, it):
    n, c, hw, f, k, st, pad = it["layer"]
    y = torch.empty((n, f, it["oh"], it["oh"]), device=DEV)
    ops.conv_forward(it["x"], it["w"], it["b"], y, k, st, pad, 1, 2)
    return y


def _backward(ops, it):
    n, c, hw, f, k, st, pad = it["layer"]
    y = _forward(ops, it)
    dy, dx = it["dy"].clone(), torch.zeros_like(it["x"])
    dw, db = torch.zeros_like(it["w"]), torch.zeros_like(it["b"])
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, hw, hw, f, k, st, pad, 1)), device=DEV)
    ops.conv_backward(it["x"], it["w"], y, dy, dx, dw, db, k, st, pad, 1, 2, ws)
    return dx, dw


def test_outputs_are_bit_identical_with_and_without_prepack():
    from bcnn_amd import _lib, ops
    L = _lib.load()
    rs = np.random.RandomState(3)
    items = [_make(rs, l) for l in LAYERS]
    ref_y = [_forward(ops, it) for it in items]
    ref_b = [_backward(ops, it) for it in items]
    arr = _descs(items)
    for rep in range(2):  # the second round reuses the buffers and the uploaded job tables
        L.bcnn_hip_conv_prepack(C.cast(arr, C.c_void_p), len(items), 0)
        for it, y0 in zip(items, ref_y):
            assert torch.equal(_forward(ops, it), y0), (rep, it["layer"])
        L.bcnn_hip_conv_prepack(C.cast(arr, C.c_void_p), len(items), 1)
        for it, (dx0, dw0) in zip(items, ref_b):
            dx, dw = _backward(ops, it)  # its forward call finds no forward copy (other form) and packs itself
            assert torch.equal(dx, dx0) and torch.equal(dw, dw0), (rep, it["layer"])
    L.bcnn_hip_conv_prepack_reset()
    for it, y0 in zip(items, ref_y):
        assert torch.equal(_forward(ops, it), y0)


def test_a_copy_is_used_once_and_a_new_batch_discards_unused_ones():
    from bcnn_amd import _lib, ops
    L = _lib.load()
    rs = np.random.RandomState(4)
    items = [_make(rs, l) for l in LAYERS[:4]]
    arr = _descs(items)
    L.bcnn_hip_conv_prepack(C.cast(arr, C.c_void_p), len(items), 0)
    y_old = [_forward(ops, it) for it in items]  # consumes every copy
    for it in items:
        it["w"].mul_(-0.5)
    torch.cuda.synchronize()
    y_new = [_forward(ops, it) for it in items]  # no copy left: packs the rewritten weights
    L.bcnn_hip_conv_prepack(C.cast(arr, C.c_void_p), len(items), 0)  # copies of the rewritten weights ...
    for it in items:
        it["w"].mul_(-2.0)  # ... which are rewritten again (back to the first values) before any use
    torch.cuda.synchronize()
    L.bcnn_hip_conv_prepack(C.cast(arr, C.c_void_p), len(items), 0)  # the unused copies are discarded, new ones made
    for it, y0, y1 in zip(items, y_old, y_new):
        y = _forward(ops, it)
        assert torch.equal(y, y0) and not torch.equal(y, y1), it["layer"]
    # a batch that does not name a layer leaves it without a copy
    L.bcnn_hip_conv_prepack(C.cast(arr, C.c_void_p), 1, 0)
    for it, y0 in zip(items, y_old):
        assert torch.equal(_forward(ops, it), y0)
    L.bcnn_hip_conv_prepack_reset()
