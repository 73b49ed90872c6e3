"""The backward sums of a stand-alone batch-norm node taken from the data-gradient epilogue of the 1x1 convolution behind
it (bcnn_hip_conv_backward_bnsums + bcnn_hip_batchnorm_backward_finalize: one partial per channel and 64 pixels, combined
in double) against the separate sweep (bcnn_hip_batchnorm_backward_sums over the gradient the convolution wrote): the same
sums S1 = sum dz, S2 = sum dz (y - mean) (bcnn_batchnorm_layer.c:263-281) in another fixed order -- dbias, dscales, dmean,
dvar to 1e-5, the convolution's own outputs bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (n, c_in, hw, f): 7 x 7 planes (49: not a multiple of 4 -> scalar partner loads), 14 x 14 (images end inside a lane's 16
# columns), ragged channel counts, n * hw not a multiple of 64
SHAPES = [(4, 64, 14, 128), (3, 96, 7, 160), (2, 40, 28, 72), (5, 64, 6, 64), (2, 128, 56, 64)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_c%d_%dx%d_f%d" % (s[0], s[1], s[2], s[2], s[3]))
def test_sums_from_the_convolution_epilogue_match_the_separate_sweep(shape):
    from bcnn_amd import _lib, ops
    L = _lib.load()
    n, c, hw, f = shape
    rs = np.random.RandomState(9)
    T = lambda *sh: torch.from_numpy(rs.uniform(-1, 1, sh).astype(np.float32)).to(DEV)
    x, wt, bias = T(n, c, hw, hw), T(f, c, 1, 1) * 0.2, T(f)
    y = torch.empty((n, f, hw, hw), device=DEV)
    ops.conv_forward(x, wt, bias, y, 1, 1, 0, 1, 0)
    dy0 = T(n, f, hw, hw)
    prev_y, mean = T(n, c, hw, hw), T(c) * 0.3
    var, scales = torch.rand(c, device=DEV) + 0.5, torch.rand(c, device=DEV) + 0.5
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, hw, hw, f, 1, 1, 0, 1)), device=DEV)
    P = lambda t: 0 if t is None else t.data_ptr()

    def run(fused):
        dy, dx = dy0.clone(), torch.zeros_like(x)
        dw, db = torch.zeros_like(wt), torch.zeros_like(bias)
        dsc, dbb = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        dm, dv = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        if fused:
            sums = torch.empty(L.bcnn_hip_conv_bnsums_size(n, c, hw, hw), device=DEV)
            splits = L.bcnn_hip_conv_backward_bnsums(P(x), P(wt), P(bias), P(y), P(dy), P(dx), P(dw), P(db), n, c, hw, hw, f, 1,
                                                     1, 0, 1, 0, None, None, 0, None, None, None, None, None, None, None, None,
                                                     P(ws), ws.numel(), P(prev_y), P(mean), P(sums), sums.numel())
            assert splits > 0, "this shape is expected on the LDS-DMA kernel"
            L.bcnn_hip_batchnorm_backward_finalize(P(sums), splits, P(scales), P(dsc), P(dbb), P(var), P(dm), P(dv), c)
        else:
            ops.conv_backward(x, wt, y, dy, dx, dw, db, 1, 1, 0, 1, 0, ws)
            ops.batchnorm_backward_sums(dx, scales, dsc, dbb, mean, var, dm, dv, prev_y)
        return dx, dw, db, dsc, dbb, dm, dv

    a, b = run(False), run(True)
    for u, v in zip(a[:3], b[:3]):
        assert torch.equal(u.view(torch.int32), v.view(torch.int32))
    for u, v in zip(a[3:], b[3:]):
        den = float(u.abs().max())
        assert float((u.double() - v.double()).abs().max()) <= 1e-5 * den + 1e-7
