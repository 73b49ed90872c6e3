"""The K-split tail of the fused Winograd forward / dX kernel (conv_winograd_fused.hip: the blocks a last, partly filled
round of the persistent workgroups would hold are cut along the input channels and dealt out over all CUs; a fix-up kernel
adds the pieces in channel order, stores them and takes the batch-norm statistics). Shapes whose block counts leave such a
tail (288 and 576 blocks on 256 CUs; a ragged channel count; a partial last tile block): raw forward with batch-norm
statistics and dX against torch's float64 convolution at 1e-5 (the parity bar is 1e-4; 3e-5 where an F(4x4, 3x3) kernel took
the shape -- read off the dispatch trace of include/bcnn_hip.h, not re-derived from the shape), and bit-identical repeats."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (n, c, f, h, w): blocks = ceil(n * ceil(h/2) * ceil(w/2) / 64) * ceil(f / 64)
SHAPES = [(32, 64, 64, 48, 48), (72, 128, 128, 32, 32), (36, 72, 96, 30, 34), (130, 256, 256, 14, 14)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_c%d_f%d_%dx%d" % s)
def test_tail_pieces_add_up(shape):
    from bcnn_amd import ops
    n, c, f, h, w = shape
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.rand((n, c, h, w), device=DEV, generator=gen) * 2 - 1
    wt = (torch.rand((f, c, 3, 3), device=DEV, generator=gen) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
    b = torch.rand(f, device=DEV, generator=gen) - 0.5
    Z = lambda: torch.zeros(f, device=DEV)

    def forward():
        bn = dict(run_mean=Z(), run_var=Z() + 1, scales=torch.rand(f, device=DEV, generator=gen) + 0.5, saved_mean=Z(),
                  saved_var=Z(), workspace=torch.full((n, f, h, w), float("nan"), device=DEV))
        y = torch.empty((n, f, h, w), device=DEV)
        ops.conv_forward(x, wt, b, y, 3, 1, 1, 1, 0, bn=bn)  # TRAIN: raw output + statistics from the kernel's epilogue
        torch.cuda.synchronize()
        return bn

    import ctypes
    from bcnn_amd import _lib
    L = _lib.load()

    def traced(fn):
        """fn() under the dispatch trace of include/bcnn_hip.h -> (result, set of kernel families that ran)"""
        L.bcnn_hip_trace_enable(1)
        out = fn()
        n_ = L.bcnn_hip_trace_read(None, 0)
        buf = ctypes.create_string_buffer(n_ + 1)
        L.bcnn_hip_trace_read(buf, n_ + 1)
        L.bcnn_hip_trace_enable(0)
        return out, set(buf.value.decode().split())

    bn, ran = traced(forward)
    raw = F.conv2d(x.double(), wt.double(), None, padding=1)
    rel = lambda a, r: float((a.double() - r).abs().max() / max(float(r.abs().max()), 1e-30))
    # the bar follows the kernel that TOOK the shape (asked of the dispatch trace, not re-derived from the shape): F(4x4, 3x3) in
    # fp32 sits at ~1e-5 (tools/exp/wino43_error.py), F(2x2, 3x3) at ~4e-7
    assert any(k.startswith(("wino43", "wino_fused")) for k in ran), ran
    tol = 3e-5 if any(k.startswith("wino43") for k in ran) else 1e-5
    assert rel(bn["workspace"], raw) <= tol
    mean = raw.mean(dim=(0, 2, 3))
    var = (raw * raw).mean(dim=(0, 2, 3)) - mean * mean
    assert rel(bn["saved_mean"], mean) <= tol and rel(bn["saved_var"], var) <= 1e-4
    bn2 = forward()
    assert torch.equal(bn["workspace"], bn2["workspace"]) and torch.equal(bn["saved_mean"], bn2["saved_mean"])
    assert torch.equal(bn["saved_var"], bn2["saved_var"])
    # dX: no batch-norm, no activation -> dy is used as given
    y = torch.empty((n, f, h, w), device=DEV)
    dy = (torch.rand((n, f, h, w), device=DEV, generator=gen) * 2 - 1) * 0.1
    dx = torch.full_like(x, float("nan"))
    dw, db = torch.zeros_like(wt), torch.zeros(f, device=DEV)
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device=DEV)
    _, ran = traced(lambda: ops.conv_backward(x, wt, y, dy.clone(), dx, dw, db, 3, 1, 1, 1, 0, ws))
    torch.cuda.synchronize()
    dxr = F.conv_transpose2d(dy.double(), wt.double(), None, padding=1)
    assert rel(dx, dxr) <= (3e-5 if any(k.startswith("wino43") and k.endswith(":dx") for k in ran) else 1e-5)
    dx2 = torch.full_like(x, float("nan"))
    ops.conv_backward(x, wt, y, dy.clone(), dx2, torch.zeros_like(wt), torch.zeros(f, device=DEV), 3, 1, 1, 1, 0, ws)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
