"""A convolution node with batch-norm + activation feeding a depthwise node that normalises the convolution's raw output on
the fly (MobileNet: [conv 1x1 + BN + ReLU] -> [depthwise 3x3]). In backward the depthwise kernel writes the gradient of the
convolution node's output; bcnn_hip_depthwise_backward_bnin_sums also leaves the per-channel sums that node's batch-norm
backward starts with (S1 = sum g, S2 = sum g (raw - mean), g = dx act'(y): bcnn_batchnorm_layer.c:263-281), and
bcnn_hip_conv_backward_presummed takes them instead of sweeping (dy, raw) again. Against the two separate calls: the
depthwise outputs bit for bit (same kernel, same operations), the batch-norm results to 1e-5 (the same sums in another
fixed order), the convolution gradients behind them to 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (n, c_in, c, hw, stride, with batch-norm node behind the depthwise node): planes of 7 x 7 and 14 x 14 (many per tile, a
# 16-lane row each), 28 x 28 (four per tile, a wave each), 112 x 112 (one plane in bands), odd widths, ragged plane counts
CASES = [(3, 16, 40, 7, 1, 0), (2, 24, 36, 14, 1, 1), (2, 16, 18, 28, 2, 0), (1, 8, 6, 112, 1, 1), (2, 8, 5, 112, 2, 0),
         (3, 12, 10, 9, 1, 1), (2, 8, 3, 56, 1, 0)]


def _close(a, b, tol, what):
    d = (a.double() - b.double()).abs().max().item()
    ref = b.double().abs().max().item()
    assert d <= tol * max(ref, 1e-3), "%s: %.3g of %.3g" % (what, d, ref)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dto%d_%dx%d_s%d_bn%d" % (c[0], c[1], c[2], c[3], c[3], c[4], c[5]))
def test_sums_from_the_depthwise_kernel_match_the_separate_sweep(case):
    from bcnn_amd import _lib, ops
    L = _lib.load()
    n, cin, c, hw, st, with_bn = case
    rs = np.random.RandomState(17)
    T = lambda *sh: torch.from_numpy(rs.uniform(-1, 1, sh).astype(np.float32)).to(DEV)
    P = lambda t: 0 if t is None else t.data_ptr()
    RELU = 2  # BCNN_HIP_ACT_RELU
    # the producer: 1x1 convolution + batch-norm + ReLU, TRAIN mode; its pre-normalisation output stays in `raw`
    x0, w1, b1 = T(n, cin, hw, hw), T(c, cin, 1, 1) * 0.3, T(c) * 0.2
    Z = lambda: torch.zeros(c, device=DEV)
    bn = dict(run_mean=Z(), run_var=Z() + 1, scales=torch.rand(c, device=DEV) + 0.5, saved_mean=Z(), saved_var=Z(),
              workspace=torch.empty((n, c, hw, hw), device=DEV))
    y1 = torch.empty((n, c, hw, hw), device=DEV)
    ops.conv_forward(x0, w1, b1, y1, 1, 1, 0, 1, RELU, bn=bn)
    raw = bn["workspace"]
    oh = (hw + 2 - 3) // st + 1
    wd, bd = T(c, 3, 3) * 0.3, T(c) * 0.1
    y2 = torch.empty((n, c, oh, oh), device=DEV)
    assert L.bcnn_hip_depthwise_bnin_fusable(n, c, hw, hw, 3, st, 1, RELU, RELU)
    L.bcnn_hip_depthwise_forward_bnin(P(raw), P(wd), P(bd), P(y2), n, c, hw, hw, 3, st, 1, RELU, 0, 0, P(bn["saved_mean"]),
                                      P(bn["saved_var"]), P(bn["scales"]), P(b1), RELU)
    dy2 = T(n, c, oh, oh) * 0.1
    # an optional stand-alone batch-norm node behind the depthwise node, backward state as its sums-only worker leaves it
    bmean, bvar = T(c) * 0.2, torch.rand(c, device=DEV) + 0.5
    bsc, bdm, bdv = torch.rand(c, device=DEV) + 0.5, T(c) * 0.01, T(c) * 0.01
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, cin, hw, hw, c, 1, 1, 0, 1)), device=DEV)

    def run(fused):
        dy, dx = dy2.clone(), torch.zeros((n, c, hw, hw), device=DEV)
        dwd, dbd = torch.zeros_like(wd), torch.zeros_like(bd)
        bnp = [P(bmean), P(bvar), P(bsc), P(bdm), P(bdv)] if with_bn else [0] * 5
        inp = [P(bn["saved_mean"]), P(bn["saved_var"]), P(bn["scales"]), P(b1), RELU]
        sums, splits = None, 0
        if fused:
            sums = torch.full((L.bcnn_hip_depthwise_insums_size(n, c, hw, hw, 3, st, 1),), float("nan"), device=DEV)
            splits = L.bcnn_hip_depthwise_backward_bnin_sums(P(raw), P(wd), P(y2), P(dy), P(dx), P(dwd), P(dbd), n, c, hw, hw,
                                                             3, st, 1, RELU, 1, *bnp, *inp, P(sums), sums.numel())
            assert splits > 0 and sums.numel() >= c * splits * 2
        elif with_bn:
            L.bcnn_hip_depthwise_backward_bn_bnin(P(raw), P(wd), P(y2), P(dy), P(dx), P(dwd), P(dbd), n, c, hw, hw, 3, st, 1,
                                                  RELU, 1, *bnp, *inp)
        else:
            L.bcnn_hip_depthwise_backward_bnin(P(raw), P(wd), P(y2), P(dy), P(dx), P(dwd), P(dbd), n, c, hw, hw, 3, st, 1, RELU,
                                               1, *inp)
        # the producer's backward on the gradient just written
        g = dx.clone()
        dx0, dw1, db1 = torch.zeros_like(x0), torch.zeros_like(w1), torch.zeros_like(b1)
        dsc, dm, dv = Z(), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        args = [P(x0), P(w1), P(b1), P(y1), P(g), P(dx0), P(dw1), P(db1), n, cin, hw, hw, c, 1, 1, 0, 1, RELU, 0, 0, 1,
                P(bn["scales"]), P(dsc), P(bn["saved_mean"]), P(bn["saved_var"]), P(dm), P(dv), 0, P(raw), P(ws), ws.numel()]
        if fused:
            L.bcnn_hip_conv_backward_presummed(*args, P(sums), splits, 0, 0, 0, 0)
        else:
            L.bcnn_hip_conv_backward(*args)
        torch.cuda.synchronize()
        return dict(dx=dx, dwd=dwd, dbd=dbd, dy=dy, g=g, dx0=dx0, dw1=dw1, db1=db1, dsc=dsc, dm=dm, dv=dv)

    a, b = run(False), run(True)
    for k in ("dx", "dwd", "dbd", "dy"):
        assert torch.equal(a[k], b[k]), k
    for k in ("db1", "dsc", "dm", "dv", "g"):
        _close(b[k], a[k], 1e-5, k)
    for k in ("dx0", "dw1"):
        _close(b[k], a[k], 1e-4, k)


def test_no_sums_when_the_gradient_accumulates():
    from bcnn_amd import _lib
    L = _lib.load()
    n, c, hw = 2, 8, 14
    rs = np.random.RandomState(5)
    T = lambda *sh: torch.from_numpy(rs.uniform(-1, 1, sh).astype(np.float32)).to(DEV)
    P = lambda t: t.data_ptr()
    raw, wd, y2, dy = T(n, c, hw, hw), T(c, 3, 3), T(n, c, hw, hw), T(n, c, hw, hw)
    dx, dwd, dbd = T(n, c, hw, hw), torch.zeros(c, 3, 3, device=DEV), torch.zeros(c, device=DEV)
    mean, var, sc, bb = T(c) * 0.1, torch.rand(c, device=DEV) + 0.5, torch.rand(c, device=DEV) + 0.5, T(c) * 0.1
    sums = torch.zeros(L.bcnn_hip_depthwise_insums_size(n, c, hw, hw, 3, 1, 1), device=DEV)
    splits = L.bcnn_hip_depthwise_backward_bnin_sums(P(raw), P(wd), P(y2), P(dy), P(dx), P(dwd), P(dbd), n, c, hw, hw, 3, 1, 1, 2,
                                                     0, 0, 0, 0, 0, 0, P(mean), P(var), P(sc), P(bb), 2, P(sums), sums.numel())
    assert splits == 0  # dx is not complete when this kernel is not its only writer
