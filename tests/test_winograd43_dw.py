"""Weight gradient on Winograd F(4x4,3x3) in its transposed form (bcnn_amd/csrc/conv_winograd43_dw.hip) against float64 on the
same inputs: product shapes (the kernel is picked by the product rule -- the dispatch trace says so), beta = 1 onto what
dw held before (the reference's momentum carry, bcnn_conv_layer.c:533-560), determinism, tile ranges that end inside a
tile row / inside an image, a last split that is shorter than the others, 128 channels (2 x 4 blocks)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _trace(on):
    from bcnn_amd import _lib
    L = _lib.load()
    if on:
        L.bcnn_hip_trace_enable(1)
        return None
    n = L.bcnn_hip_trace_read(None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    L.bcnn_hip_trace_read(buf, n + 1)
    L.bcnn_hip_trace_enable(0)
    return set(buf.value.decode().split())


SHAPES = [
    (16, 64, 64, 56, 56),    # ResNet-18 stage 1 (N reduced): 14 tiles per row, 8-tile periods straddle rows and images
    (32, 128, 128, 28, 28),  # stage 2: 7 tiles per row, 2 x 4 channel blocks
    (24, 64, 128, 28, 36),   # H != W, 9 tiles per row, F != C
    (41, 64, 64, 32, 32),    # a tile count that is no multiple of the period or of the split
    (16, 96, 160, 28, 28),   # channel counts that fill neither the last 32-channel nor the last 64-channel block
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_weight_gradient_matches_float64(shape):
    import torch
    import torch.nn.functional as F
    from bcnn_amd import ops
    DEV = "cuda:0"
    n, c, f, h, w = shape
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.rand((n, c, h, w), device=DEV, generator=gen) * 2 - 1
    x = x * (x > 0)                                                     # a ReLU output, like the layer's input in the net
    wt = (torch.rand((f, c, 3, 3), device=DEV, generator=gen) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
    dy = (torch.rand((n, f, h, w), device=DEV, generator=gen) * 2 - 1) * 0.1
    y = torch.empty((n, f, h, w), device=DEV)
    carry = (torch.rand((f, c, 3, 3), device=DEV, generator=gen) * 2 - 1) * 0.5   # what dw holds before: beta = 1
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device=DEV)

    def run():
        dw, db = carry.clone(), torch.zeros(f, device=DEV)
        dx = torch.empty_like(x)
        ops.conv_backward(x, wt, y, dy.clone(), dx, dw, db, 3, 1, 1, 1, 0, ws)
        torch.cuda.synchronize()
        return dw, db

    _trace(True)
    dw, db = run()
    kernels = _trace(False)
    assert "wino43_dw_kernel" in kernels, kernels
    # float64 reference: dW[f][c][kh][kw] = sum_{n,h,w} dy[n][f][h][w] x[n][c][h+kh-1][w+kw-1]
    xd, dyd = x.double().cpu(), dy.double().cpu()
    xp = F.pad(xd, (1, 1, 1, 1))
    ref = torch.zeros((f, c, 3, 3), dtype=torch.float64)
    for kh in range(3):
        for kw in range(3):
            ref[:, :, kh, kw] = torch.einsum("nfhw,nchw->fc", dyd, xp[:, :, kh:kh + h, kw:kw + w])
    got = dw.double().cpu() - carry.double().cpu()
    scale = float(ref.abs().max())
    err = (got - ref).abs()
    # the carry is O(0.5) next to gradients of O(scale): its rounding (2^-24 x 0.5) is part of the sum
    floor = 6e-8 * 0.5
    assert float(err.max()) <= 3e-5 * scale + floor, (float(err.max()) / scale, scale)
    bound = 1e-4 * ref.abs() + 1e-5 * scale + floor
    worst = float((err / bound).max())
    assert worst <= 1.0, worst
    assert float((db.double().cpu() - dyd.sum(dim=(0, 2, 3))).abs().max()) <= 1e-4 * float(dyd.sum(dim=(0, 2, 3)).abs().max()) + 1e-5
    dw2, _ = run()
    assert torch.equal(dw, dw2)
    print("F(4x4,3x3) dW %s: max|err|/max|ref| %.2e, worst element %.3f of its bound" % (shape, float(err.max()) / scale, worst))
