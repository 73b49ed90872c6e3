"""Where the in-tree reference build stops being a valid checker.

The reference's generic blocked sgemm (bcnn_mat.c:2590-2625, used whenever an operand is transposed) advances to
the next column block of B with `j * NC` and to the next depth block of A with `l * KC`, i.e. as if both
operands were untransposed. With MC/KC/NC = 128/384/4096 (bcnn_mat.h:84-86) that is wrong as soon as
  * dW  = gemm(0, 1, F/g, C/g*k*k, OH*OW): the reduction-side operand is transposed and C/g*k*k > 4096;
  * dX  = gemm(1, 0, C/g*k*k, OH*OW, F/g): the weight operand is transposed and F/g > 384.
The forward gemm (nn path) and the USE_BLAS build (cblas_sgemm), which is the target north_star names, are not
affected. The CPU test pins both limits on the reference itself, so that the golden fixtures and net-parity tests
(all generated below the limits) are known to sit on the sound side; the GPU test shows this build agrees with the
mathematical result (torch fp64) on both sides of each limit, 1e-4 relative like every other conv parity test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_bind as rb

TOL = 1e-4
# (C, F): one case just below and one just above each limit, plus ResNet-18's last stage
CASES = [(455, 16), (456, 16), (64, 384), (64, 385), (512, 512)]
HW, N = 4, 2


def _step(net, c, f):
    node = net.conv(f, 3, 1, 1, 1, 0, 0, "input", "out")  # 3x3 / s1 / p1, no BN, no activation
    net.compile()
    rs = np.random.RandomState(c * 1000 + f)
    ix, iw, ib, iy = net.node_src(node, 0), net.node_src(node, 1), net.node_src(node, 2), net.node_dst(node)
    x = rs.uniform(-1, 1, net.shape(ix)).astype(np.float32)
    w = (rs.uniform(-1, 1, net.shape(iw)) * 0.02).astype(np.float32)
    dy = (rs.uniform(-1, 1, net.shape(iy)) * 0.1).astype(np.float32)
    return (ix, iw, ib, iy), x, w, dy


def _expected(x, w, dy):
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w.reshape(w.shape[0], x.shape[1], 3, 3), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xt, wt, None, stride=1, padding=1)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    return y.detach().numpy(), wt.grad.numpy().reshape(w.shape), xt.grad.numpy()


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


def _reference(c, f):
    net = rb.RefNet(mode=rb.MODE_TRAIN, w=HW, h=HW, c=c, n=N, input_grad=True)
    net.L.ref_set_threads(net.net, 4)
    (ix, iw, ib, iy), x, w, dy = _step(net, c, f)
    net.data(ix)[...] = x
    net.data(iw)[...] = w
    net.data(ib)[...] = 0
    net.forward()
    net.grad(iy)[...] = dy
    net.grad(iw)[...] = 0
    net.backward()
    y, dw, dx = _expected(x, w, dy)
    out = _rel(net.data(iy), y), _rel(net.grad(iw), dw), _rel(net.grad(ix), dx)
    net.close()
    return out


def test_reference_in_tree_gemm_is_sound_only_below_its_block_limits():
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    for c, f in CASES:
        ey, edw, edx = _reference(c, f)
        assert ey < TOL, (c, f, ey)  # forward: nn path, sound at any size
        assert (edw < TOL) == (c * 9 <= 4096), ("dW", c, f, edw)
        assert (edx < TOL) == (f <= 384), ("dX", c, f, edx)


@pytest.mark.gpu
@pytest.mark.parametrize("c,f", CASES)
def test_hip_conv_matches_fp64_on_both_sides_of_the_reference_limits(c, f):
    from bcnn_amd import capi
    net = capi.Net(mode=capi.MODE_TRAIN, w=HW, h=HW, c=c, n=N, input_grad=True)
    (ix, iw, ib, iy), x, w, dy = _step(net, c, f)
    net.data(ix)[...] = x
    net.data(iw)[...] = w
    net.data(ib)[...] = 0
    for i in (ix, iw, ib):
        net.upload(i)
    net.forward()
    net.download(iy)
    net.grad(iy)[...] = dy
    net.grad(iw)[...] = 0
    net.upload(iy, True)
    net.upload(iw, True)
    net.backward()
    for i in (ix, iw, iy):
        net.download(i)
    y, dw, dx = _expected(x, w, dy)
    assert _rel(net.grad(iw), dw) < TOL
    assert _rel(net.grad(ix), dx) < TOL
    net.close()
