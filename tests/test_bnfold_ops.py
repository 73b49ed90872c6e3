"""The batch-norm fold entry points of include/bcnn_hip.h called directly: bcnn_hip_conv_bnfold_fusable,
bcnn_hip_batchnorm_forward_stats_only, bcnn_hip_conv_set_input_bnfold + bcnn_hip_conv_forward / _backward. A stand-alone
batch-norm (no activation) in front of a 1x1 convolution with its own fused batch-norm, TRAIN mode
(bcnn_batchnorm_layer.c:226-241 into bcnn_conv_layer.c:438-481): folded against float64 of the unfolded chain --
pre-normalisation values up to the per-filter constant W b, saved statistics, the running mean INCLUDING W b, the normalised
output, the weight gradient (dy y^T) diag(a), dX unchanged -- for grouped-free, ragged channel counts and the bias == 0 / 1
special cases of bcnn_add_scalar (bcnn_mat.c:366-412)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # n, c, h, w, f
    (4, 64, 14, 14, 128),
    (3, 40, 9, 11, 72),     # ragged channel / filter counts
    (8, 128, 7, 7, 48),
    (5, 24, 16, 12, 64),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_c%d_%dx%d_f%d" % s)
def test_fold_equals_the_unfolded_chain_in_float64(shape):
    from bcnn_amd import _lib, ops
    L = _lib.load()
    n, c, h, w, f = shape
    assert L.bcnn_hip_conv_bnfold_fusable(n, c, h, w, f) == 1
    assert L.bcnn_hip_conv_bnfold_fusable(n, c, h, w, 16) == 0          # too few filters for the kernels that take the fold
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    rnd = lambda *s: torch.rand(s, device=DEV, generator=gen)
    y = rnd(n, c, h, w) * 2 - 1 + (rnd(1, c, 1, 1) - 0.5)               # the batch-norm's input, per-channel offsets
    wt = (rnd(f, c, 1, 1) * 2 - 1) * (3.0 / c) ** 0.5
    bn1_scales, bn1_bias = rnd(c) + 0.5, rnd(c) - 0.5
    bn1_bias[0] = 0.0
    bn1_bias[1] = 1.0                                                    # bcnn_add_scalar adds nothing for exactly 0 and 1
    bn2_scales, conv_bias = rnd(f) + 0.5, rnd(f) - 0.5
    Zc, Zf = (lambda: torch.zeros(c, device=DEV)), (lambda: torch.zeros(f, device=DEV))
    # ---- the batch-norm node's part: statistics only ----
    m1, v1, rm1, rv1 = Zc(), Zc(), Zc(), Zc() + 1
    P = lambda t: t.data_ptr() if t is not None else None
    L.bcnn_hip_batchnorm_forward_stats_only(P(y), P(rm1), P(rv1), P(bn1_scales), P(bn1_bias), P(m1), P(v1), n, c, h * w, None, 0)
    y64 = y.double().cpu()
    mean1 = y64.mean(dim=(0, 2, 3))
    var1 = (y64 * y64).mean(dim=(0, 2, 3)) - mean1 * mean1
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / max(float(r.abs().max()), 1e-30))
    assert rel(m1, mean1) <= 1e-5 and rel(v1, var1) <= 1e-4
    # ---- the convolution with the fold announced: x = the batch-norm's INPUT ----
    bn = dict(run_mean=Zf(), run_var=Zf() + 1, scales=bn2_scales, saved_mean=Zf(), saved_var=Zf(),
              workspace=torch.full((n, f, h, w), float("nan"), device=DEV))
    out = torch.empty((n, f, h, w), device=DEV)
    L.bcnn_hip_conv_set_input_bnfold(P(m1), P(v1), P(bn1_scales), P(bn1_bias))
    ops.conv_forward(y, wt, conv_bias, out, 1, 1, 0, 1, 2, bn=bn)       # ReLU behind the convolution's own batch-norm
    torch.cuda.synchronize()
    # float64 of the unfolded chain
    a1 = bn1_scales.double().cpu() / torch.sqrt(var1 + 1e-6)
    b_eff = bn1_bias.double().cpu().clone()
    b_eff[1] = 0.0                                                       # the reference adds nothing for a bias of exactly 1
    z64 = (y64 - mean1.view(1, c, 1, 1)) * a1.view(1, c, 1, 1) + b_eff.view(1, c, 1, 1)
    w64 = wt.double().cpu().view(f, c)
    raw64 = torch.einsum("fc,nchw->nfhw", w64, z64)
    mean2 = raw64.mean(dim=(0, 2, 3))
    var2 = (raw64 * raw64).mean(dim=(0, 2, 3)) - mean2 * mean2
    wb = w64 @ (b_eff - mean1 * a1)                                      # the per-filter constant the fold leaves out of `workspace`
    ws = bn["workspace"].double().cpu()
    assert rel(ws + wb.view(1, f, 1, 1), raw64) <= 3e-5
    assert rel(bn["saved_var"], var2) <= 1e-4
    assert rel(bn["saved_mean"].double().cpu() + wb, mean2) <= 3e-5      # saved mean belongs to the stored values ...
    assert rel(bn["run_mean"], 0.1 * mean2) <= 3e-5                      # ... the RUNNING mean is the reference-visible one
    xhat = (raw64 - mean2.view(1, f, 1, 1)) / torch.sqrt(var2 + 1e-6).view(1, f, 1, 1)
    out64 = torch.relu(xhat * bn2_scales.double().cpu().view(1, f, 1, 1) + conv_bias.double().cpu().view(1, f, 1, 1))
    assert rel(out, out64) <= 1e-4
    # ---- backward with the fold announced: dW = (g y^T) diag(a), dX = W^T g against the batch-norm's OUTPUT gradient ----
    dy = (rnd(n, f, h, w) * 2 - 1) * 0.1
    g = dy.clone()
    dz = torch.full((n, c, h, w), float("nan"), device=DEV)
    dw, db = torch.zeros_like(wt), Zf()
    bnb = dict(bn, dscales=Zf(), dmean=Zf(), dvar=Zf())
    wsz = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 1, 1, 0, 1)), device=DEV)
    L.bcnn_hip_conv_set_input_bnfold(P(m1), P(v1), P(bn1_scales), P(bn1_bias))
    ops.conv_backward(y, wt, out, g, dz, dw, db, 1, 1, 0, 1, 2, wsz, None, None, bnb, conv_bias)
    torch.cuda.synchronize()
    # float64: gradient of the convolution's pre-normalisation output from dy (ReLU', batch-norm backward, eps 1e-5)
    gy = dy.double().cpu() * (out64 > 0)
    M = n * h * w
    s2 = bn2_scales.double().cpu().view(1, f, 1, 1)
    istd = 1.0 / torch.sqrt(var2 + 1e-5).view(1, f, 1, 1)
    xc = raw64 - mean2.view(1, f, 1, 1)
    gxh = gy * s2
    dvar = (gxh * xc).sum(dim=(0, 2, 3)).view(1, f, 1, 1) * (-0.5) * istd ** 3
    dmean = -(gxh * istd).sum(dim=(0, 2, 3)).view(1, f, 1, 1) + dvar * (-2.0 * xc).mean(dim=(0, 2, 3)).view(1, f, 1, 1)
    graw = gxh * istd + dvar * 2.0 * xc / M + dmean / M
    assert rel(g, graw) <= 1e-4                                          # the rewritten dst gradient
    dw64 = torch.einsum("nfhw,nchw->fc", graw, z64)                      # what the unfolded chain accumulates ...
    # ... the fold leaves out b (x) sum_q graw, which is rounding noise (the batch sum of a batch-norm input gradient)
    assert rel(dw.view(f, c), dw64) <= 1e-4
    dz64 = torch.einsum("fc,nfhw->nchw", w64, graw)
    assert rel(dz, dz64) <= 1e-4


def test_an_announced_fold_does_not_leak_into_the_next_pass():
    """a host path that announces a fold and then does not reach a convolution call must not hand it to the next 1x1
    batch-norm convolution of the thread: the pass-level prepack entry point (the start of every bcnn_forward /
    bcnn_backward) drops it"""
    from bcnn_amd import _lib, ops
    L = _lib.load()
    n, c, h, w, f = 2, 64, 8, 8, 64
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand((n, c, h, w), device=DEV, generator=gen) * 2 - 1
    wt = (torch.rand((f, c, 1, 1), device=DEV, generator=gen) * 2 - 1) * (3.0 / c) ** 0.5
    Zf = lambda: torch.zeros(f, device=DEV)

    def run():
        bn = dict(run_mean=Zf(), run_var=Zf() + 1, scales=Zf() + 1, saved_mean=Zf(), saved_var=Zf(),
                  workspace=torch.empty((n, f, h, w), device=DEV))
        y = torch.empty((n, f, h, w), device=DEV)
        ops.conv_forward(x, wt, Zf(), y, 1, 1, 0, 1, 0, bn=bn)
        torch.cuda.synchronize()
        return bn["workspace"].clone()

    plain = run()
    junk = torch.rand(c, device=DEV, generator=gen) + 0.5
    L.bcnn_hip_conv_set_input_bnfold(junk.data_ptr(), junk.data_ptr(), junk.data_ptr(), junk.data_ptr())
    L.bcnn_hip_conv_prepack(None, 0, 0)   # a new pass begins: whatever was announced and not consumed is gone
    assert torch.equal(run(), plain)
