"""Max-pooling that normalises its input on the fly (bcnn_hip_maxpool_forward_bn, pool.hip) -- the pooling node behind a
convolution node with batch-norm inside a fused forward pass -- against the two separate sweeps it replaces
(bcnn_hip_batchnorm_apply, then bcnn_hip_maxpool_forward, itself pinned bit for bit to the oracle's
bcnn_maxpool_layer.c:145-191 by tests/test_hip_parity.py): same values, same argmax indexes, including ties (ReLU makes
whole windows zero: the first element in scan order has to win) and windows hanging over the bottom / right edge."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_CLAMP = 0, 2, 5, 7

# (n, c, h, w, size, stride, act)
SHAPES = [(4, 8, 112, 112, 3, 2, ACT_RELU), (2, 5, 56, 56, 3, 2, ACT_RELU), (3, 7, 28, 28, 2, 2, ACT_RELU),
          (2, 3, 33, 36, 3, 2, ACT_LRELU), (2, 3, 34, 36, 2, 2, ACT_NONE), (1, 2, 9, 12, 3, 2, ACT_CLAMP),
          (2, 4, 7, 8, 3, 2, ACT_RELU), (1, 3, 3, 4, 3, 2, ACT_RELU)]


def _out(h, size, stride):  # PADDING_SAME of the reference (bcnn_maxpool_layer.c:62-83)
    return (h + stride - 1) // stride


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_c%d_%dx%d_k%d_s%d_act%d" % s)
def test_pooling_with_batchnorm_on_the_fly_equals_the_two_sweeps(shape):
    from bcnn_amd import ops
    n, c, h, w, size, stride, act = shape
    oh, ow = _out(h, size, stride), _out(w, size, stride)
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(-2, 2, (n, c, h, w)).astype(np.float32)).to(DEV)
    x[0, 0] = -3.0  # a plane ReLU turns into zeros: every window is a tie
    mean = torch.from_numpy(rs.uniform(-0.3, 0.3, c).astype(np.float32)).to(DEV)
    var = torch.from_numpy(rs.uniform(0.2, 2.0, c).astype(np.float32)).to(DEV)
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.uniform(-0.2, 0.2, c).astype(np.float32)).to(DEV)
    sc[c - 1] = 1.0  # bcnn_scal / bcnn_add_scalar quirks: no multiply for 1, no add for 0 and 1
    b[c - 1] = 0.0
    assert ops.maxpool_bn_fusable(x, oh, ow, size, stride, act)
    y = torch.empty_like(x)
    ops.batchnorm_apply(x, y, sc, b, mean, var, act)
    p1 = torch.empty((n, c, oh, ow), device=DEV)
    i1 = torch.empty((n, c, oh, ow), device=DEV, dtype=torch.int32)
    ops.maxpool_forward(y, p1, i1, size, stride)
    p2, i2 = torch.full_like(p1, 7.0), torch.full_like(i1, -7)
    ops.maxpool_forward_bn(x, p2, i2, size, stride, sc, b, mean, var, act)
    assert torch.equal(p1.view(torch.int32), p2.view(torch.int32))
    assert torch.equal(i1, i2)
