"""The stem's pooling node and the convolution node in front of it, backward: bcnn_hip_maxpool_bn_backward takes the
batch-norm backward sums (bcnn_batchnorm_layer.c:263-281) over the POOLED gradient and the pre-normalisation values that
won their windows (kept by bcnn_hip_maxpool_forward_bn_keep), then one kernel gathers the un-pooled gradient like
bcnn_hip_maxpool_backward and applies :292-296 to it in registers; bcnn_hip_conv_backward_bn_done adds the weight / data
gradients. Against the separate calls (bcnn_hip_maxpool_backward, bcnn_hip_conv_backward): the same sums in another
fixed order -- batch-norm results and the pre-normalisation gradient to 1e-5, the convolution gradients to 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RELU, LRELU, NONE = 2, 5, 0

# (n, c_in, c, h, w, conv k, conv stride, conv pad, activation): pooled planes 8 x 8, 9 x 10 (odd height: bottom padding),
# 28 x 28 from a 7 x 7 / s2 stem-like layer, one with a source gradient (data gradient behind the fused step)
CASES = [(2, 3, 8, 16, 16, 3, 1, 1, RELU), (3, 4, 6, 18, 20, 3, 1, 1, LRELU), (2, 3, 16, 112, 112, 7, 2, 3, RELU),
         (2, 16, 12, 12, 24, 3, 1, 1, NONE), (1, 3, 5, 7, 8, 3, 1, 1, RELU)]


def _close(a, b, tol, what):
    d = (a.double() - b.double()).abs().max().item()
    ref = b.double().abs().max().item()
    assert d <= tol * max(ref, 1e-3), "%s: %.3g of %.3g" % (what, d, ref)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dto%d_%dx%d_k%ds%d_act%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[8]))
def test_fused_pooling_and_batchnorm_backward_matches_the_separate_calls(case):
    from bcnn_amd import _lib, ops
    L = _lib.load()
    n, cin, c, h, w, k, st, pad, act = case
    rs = np.random.RandomState(23)
    T = lambda *sh: torch.from_numpy(rs.uniform(-1, 1, sh).astype(np.float32)).to(DEV)
    P = lambda t: 0 if t is None else t.data_ptr()
    x0, w1, b1 = T(n, cin, h, w), T(c, cin, k, k) * 0.3, T(c) * 0.2
    oh, ow = ops.conv_out_hw(h, w, k, st, pad)
    assert ow % 4 == 0
    Z = lambda: torch.zeros(c, device=DEV)
    bn = dict(run_mean=Z(), run_var=Z() + 1, scales=torch.rand(c, device=DEV) + 0.5, saved_mean=Z(), saved_var=Z(),
              workspace=torch.empty((n, c, oh, ow), device=DEV))
    y1 = torch.empty((n, c, oh, ow), device=DEV)
    ops.conv_forward(x0, w1, b1, y1, k, st, pad, 1, act, bn=bn)
    raw = bn["workspace"]
    ph, pw = (oh + 1) // 2, ow // 2  # "same" padding: bottom / right only
    yp = torch.empty((n, c, ph, pw), device=DEV)
    idx = torch.empty((n, c, ph, pw), device=DEV, dtype=torch.int32)
    ram = torch.full((n, c, ph, pw), float("nan"), device=DEV)
    assert L.bcnn_hip_maxpool_bn_fusable(n, c, oh, ow, ph, pw, 3, 2, act, P(raw))
    L.bcnn_hip_maxpool_forward_bn_keep(P(raw), P(yp), P(idx), n, c, oh, ow, ph, pw, 3, 2, P(bn["scales"]), P(b1),
                                       P(bn["saved_mean"]), P(bn["saved_var"]), act, P(ram))
    torch.cuda.synchronize()
    assert torch.equal(ram.flatten(), raw.flatten()[idx.flatten().long()])
    # the plain pooling over the normalised tensor gives the same values and indexes
    yp2, idx2 = torch.empty_like(yp), torch.empty_like(idx)
    L.bcnn_hip_maxpool_forward(P(y1), P(yp2), P(idx2), n, c, oh, ow, ph, pw, 3, 2)
    assert torch.equal(yp, yp2) and torch.equal(idx, idx2)
    dpool = T(n, c, ph, pw) * 0.1
    want_dx = cin >= 8
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, cin, h, w, c, k, st, pad, 1)), device=DEV)

    def run(fused):
        g = torch.full((n, c, oh, ow), float("nan"), device=DEV)
        dx0 = torch.zeros_like(x0) if want_dx else None
        dw1, db1 = torch.zeros_like(w1), torch.zeros_like(b1)
        dsc, dm, dv = Z(), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        if fused:
            assert L.bcnn_hip_maxpool_bn_backward_fusable(n, c, oh, ow, ph, pw, 3, 2, act, P(raw), P(dpool), P(idx), P(g))
            L.bcnn_hip_maxpool_bn_backward(P(dpool), P(idx), P(ram), P(raw), P(g), n, c, oh, ow, ph, pw, 3, 2, P(bn["scales"]),
                                           P(dsc), P(b1), P(db1), P(bn["saved_mean"]), P(bn["saved_var"]), P(dm), P(dv), act)
            L.bcnn_hip_conv_backward_bn_done(P(x0), P(w1), P(g), P(dx0), P(dw1), n, cin, h, w, c, k, st, pad, 1, P(ws), ws.numel())
        else:
            L.bcnn_hip_maxpool_backward(P(dpool), P(idx), P(g), n, c, oh, ow, ph, pw, 3, 2, 1)
            L.bcnn_hip_conv_backward(P(x0), P(w1), P(b1), P(y1), P(g), P(dx0), P(dw1), P(db1), n, cin, h, w, c, k, st, pad, 1, act,
                                     0, 0, 1, P(bn["scales"]), P(dsc), P(bn["saved_mean"]), P(bn["saved_var"]), P(dm), P(dv), 0,
                                     P(raw), P(ws), ws.numel())
        torch.cuda.synchronize()
        return dict(g=g, dx0=dx0, dw1=dw1, db1=db1, dsc=dsc, dm=dm, dv=dv)

    a, b = run(False), run(True)
    for key in ("db1", "dsc", "dm", "dv", "g"):
        _close(b[key], a[key], 1e-5, key)
    _close(b["dw1"], a["dw1"], 1e-4, "dw1")
    if want_dx:
        _close(b["dx0"], a["dx0"], 1e-4, "dx0")
