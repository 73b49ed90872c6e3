"""The hot path at BASELINE.json's FULL sizes. The oracle cannot run these whole tensors in seconds, so each test
uses what the operator offers at any size:

* per-image independence -- conv, depthwise and maxpool treat every image separately, so image n of the full-size
  result has to equal the CPU oracle run on that one image (same tolerance as the small cases; maxpool bit-exact
  after rebasing the flat indices by n*C*H*W);
* linearity -- conv(2x) == 2*conv(x) BIT-exactly without bias (scaling by two is exact in binary fp), and the
  weight / bias gradient of the whole batch equals the sum of the gradients of its chunks, which run through
  differently shaped launches (other split factors, other tile counts);
* gather / checksum identities for maxpool: y == x.flat[indexes] everywhere, sum(dx) == sum(dy);
* batch-norm statistics and outputs against the reference's formulas evaluated in float64 by torch on the device
  (a checker for a floating-point kernel, bcnn_batchnorm_layer.c:147-242, 263-332).

Shapes: configs[1] (one 3->64 3x3 conv, 224x224, N=128); EVERY convolution shape of the ResNet-18 step at N=128 --
the 7x7/s2 RGB stem (register-staged fallback kernels), the 3x3 layers of all four stages (64ch 56x56 ... 512ch
7x7: other grid quantisations and dW tile variants), the three 3x3/s2 downsampling layers and the three 1x1/s2
projections (raw-view quirk) -- and the stem pool; MobileNet-v1 at N=256: stride-1 and stride-2 depthwise layers
and the 512 -> 1024 pointwise layer."""
import numpy as np
import pytest
import torch

from oracle import orc_bind as ob

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-4  # relative: max|a-b| / max|b| per tensor (north_star)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return ((torch.rand(shape, device=DEV, generator=g) * 2 - 1) * scale).contiguous()


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _np(t):
    return t.detach().cpu().numpy()


def _conv_case(n, c, h, w, f, k, s, p, x, wt, bias, dy, act=0):
    return dict(n=n, c=c, h=h, w=w, f=f, k=k, s=s, p=p, g=1, bn=0, act=act, mode=ob.MODE_TRAIN, input_grad=1,
                x=np.ascontiguousarray(x), wt=np.ascontiguousarray(wt), bias=np.ascontiguousarray(bias),
                dy=np.ascontiguousarray(dy))


def _conv_full_size(n, c, hw, f, k, s, p, act, images, chunk):
    from bcnn_amd import ops
    oh, ow = ops.conv_out_hw(hw, hw, k, s, p)
    x = _rand((n, c, hw, hw), 1)
    wt = _rand((f, c, k, k), 2, (3.0 / (c * k * k)) ** 0.5)
    bias = _rand((f,), 3, 0.1)
    dy0 = _rand((n, f, oh, ow), 4, 1e-2)
    ws = torch.zeros(ops.conv_workspace_size(n, c, hw, hw, f, k, s, p, 1), device=DEV)
    y = torch.full((n, f, oh, ow), 3.0, device=DEV)
    ops.conv_forward(x, wt, bias, y, k, s, p, 1, act)
    dy, dx = dy0.clone(), torch.full_like(x, 7.0)
    dw, db = torch.zeros_like(wt), torch.zeros_like(bias)
    ops.conv_backward(x, wt, y, dy, dx, dw, db, k, s, p, 1, act, ws, bias=bias)
    torch.cuda.synchronize()
    # (1) per-image independence against the oracle
    for i in images:
        cs = _conv_case(1, c, hw, hw, f, k, s, p, _np(x[i:i + 1]), _np(wt), _np(bias), _np(dy0[i:i + 1]), act)
        exp = ob.orc_conv(cs)
        assert _rel(_np(y[i:i + 1]), exp["y"]) <= TOL, ("y", i)
        assert _rel(_np(dx[i:i + 1]), exp["dx"]) <= TOL, ("dx", i)
    # (2) batch linearity of the weight / bias gradients: whole batch == sum over chunks (other launch shapes)
    dw_sum, db_sum = torch.zeros_like(wt), torch.zeros_like(bias)
    for a in range(0, n, chunk):
        xs, ys = x[a:a + chunk].contiguous(), y[a:a + chunk].contiguous()
        dys = dy0[a:a + chunk].clone()
        ops.conv_backward(xs, wt, ys, dys, None, dw_sum, db_sum, k, s, p, 1, act, ws, bias=bias)  # beta = 1
    assert _rel(_np(dw), _np(dw_sum)) <= TOL
    assert _rel(_np(db), _np(db_sum)) <= TOL
    # ... and one chunk against the oracle
    a = n - chunk
    cs = _conv_case(chunk, c, hw, hw, f, k, s, p, _np(x[a:]), _np(wt), _np(bias), _np(dy0[a:]), act)
    exp = ob.orc_conv(cs)
    dw1, db1 = torch.zeros_like(wt), torch.zeros_like(bias)
    ops.conv_backward(x[a:].contiguous(), wt, y[a:].contiguous(), dy0[a:].clone(), None, dw1, db1, k, s, p, 1, act, ws,
                      bias=bias)
    assert _rel(_np(dw1), exp["dw"]) <= TOL
    assert _rel(_np(db1), exp["db"]) <= TOL
    # bias gradient of the whole batch against a float64 reduction
    g = dy.double() if act == 0 else (dy0.double() * (y > 0))
    assert _rel(_np(db), _np(g.sum(dim=(0, 2, 3)))) <= TOL
    # (3) exact linearity of the forward pass: no bias, x -> 2x doubles every output bit-exactly
    zero = torch.zeros_like(bias)
    y1, y2 = torch.empty_like(y), torch.empty_like(y)
    ops.conv_forward(x, wt, zero, y1, k, s, p, 1, act)
    ops.conv_forward(x * 2, wt, zero, y2, k, s, p, 1, act)
    assert torch.equal(y2, y1 * 2)


def test_configs1_conv3x3_n128_224():
    _conv_full_size(n=128, c=3, hw=224, f=64, k=3, s=1, p=1, act=0, images=(0, 77, 127), chunk=4)


def test_resnet18_stage1_conv_n128_56():
    _conv_full_size(n=128, c=64, hw=56, f=64, k=3, s=1, p=1, act=2, images=(0, 127), chunk=8)


def test_resnet18_downsample_conv_n128_56_stride2():
    _conv_full_size(n=128, c=64, hw=56, f=128, k=3, s=2, p=1, act=0, images=(5,), chunk=8)


# every remaining convolution shape of the benchmarked ResNet-18 step, at the benchmark's N = 128
@pytest.mark.parametrize("name,c,hw,f,k,s,p,act,images,chunk", [
    ("stem_7x7s2", 3, 224, 64, 7, 2, 3, 2, (0, 127), 4),          # conv_igemm_kernel / conv_dw_kernel fallbacks
    ("stage2_3x3", 128, 28, 128, 3, 1, 1, 2, (1, 126), 8),
    ("stage3_3x3", 256, 14, 256, 3, 1, 1, 2, (2, 125), 8),          # dW tile variant <2,2,2,1>, 14x14 quantisation
    ("stage4_3x3", 512, 7, 512, 3, 1, 1, 0, (3, 124), 8),           # C*k*k = 4608: above the reference gemm's limit
    ("down3_3x3s2", 128, 28, 256, 3, 2, 1, 2, (4,), 8),
    ("down4_3x3s2", 256, 14, 512, 3, 2, 1, 2, (6,), 8),
    ("proj2_1x1s2", 64, 56, 128, 1, 2, 0, 0, (7, 127), 8),          # raw-view addressing (quirk 1)
    ("proj3_1x1s2", 128, 28, 256, 1, 2, 0, 0, (8,), 8),
    ("proj4_1x1s2", 256, 14, 512, 1, 2, 0, 0, (9,), 8),
])
def test_resnet18_every_conv_shape_n128(name, c, hw, f, k, s, p, act, images, chunk):
    _conv_full_size(n=128, c=c, hw=hw, f=f, k=k, s=s, p=p, act=act, images=images, chunk=chunk)


def test_mobilenet_last_pointwise_n256_7_512_to_1024():
    _conv_full_size(n=256, c=512, hw=7, f=1024, k=1, s=1, p=0, act=2, images=(0, 255), chunk=16)


def test_mobilenet_pointwise_n256_56_64_to_128():
    _conv_full_size(n=256, c=64, hw=56, f=128, k=1, s=1, p=0, act=2, images=(100,), chunk=16)


def test_resnet18_fused_batchnorm_conv_n128_56_against_float64():
    """conv + fused batch-norm + ReLU, TRAIN mode: the fused statistics epilogue and the BN backward that
    recomputes its input, against float64 formulas on the plain conv output."""
    from bcnn_amd import ops
    n, c, hw, f, k = 128, 64, 56, 64, 3
    x = _rand((n, c, hw, hw), 11)
    wt = _rand((f, c, k, k), 12, (3.0 / (c * 9)) ** 0.5)
    bias, scales = _rand((f,), 13, 0.1), _rand((f,), 14, 0.2) + 1.0
    dy0 = _rand((n, f, hw, hw), 15, 1e-2)
    t = torch.empty((n, f, hw, hw), device=DEV)
    ops.conv_forward(x, wt, torch.zeros_like(bias), t, k, 1, 1, 1, 0)  # plain conv output, checked above
    y = torch.empty_like(t)
    bn = dict(run_mean=torch.zeros(f, device=DEV), run_var=torch.zeros(f, device=DEV), scales=scales,
              saved_mean=torch.zeros(f, device=DEV), saved_var=torch.zeros(f, device=DEV),
              workspace=torch.empty_like(t))
    ops.conv_forward(x, wt, bias, y, k, 1, 1, 1, 2, None, bn, 1)
    t64 = t.double()
    m = n * hw * hw
    mean = t64.sum(dim=(0, 2, 3)) / m
    var = (t64 * t64).sum(dim=(0, 2, 3)) / m - mean * mean  # biased, one pass (quirk 4)
    assert _rel(_np(bn["saved_mean"]), _np(mean)) <= TOL
    assert np.allclose(_np(bn["saved_var"]), _np(var), rtol=1e-4, atol=1e-6)
    assert _rel(_np(bn["run_mean"]), _np(0.1 * mean)) <= TOL
    v = lambda a: a.view(1, f, 1, 1)
    xhat = (t64 - v(mean)) / torch.sqrt(v(var) + 1e-6)
    y64 = torch.relu(xhat * v(scales.double()) + v(bias.double()))
    assert _rel(_np(y), _np(y64)) <= TOL
    # backward (bcnn_batchnorm_layer.c:263-332; eps 1e-5 here)
    ws = torch.zeros(ops.conv_workspace_size(n, c, hw, hw, f, k, 1, 1, 1), device=DEV)
    dy, dx = dy0.clone(), torch.empty_like(x)
    dw, db, dsc = torch.zeros_like(wt), torch.zeros_like(bias), torch.zeros_like(bias)
    bn.update(dscales=dsc, dmean=torch.zeros(f, device=DEV), dvar=torch.zeros(f, device=DEV))
    ops.conv_backward(x, wt, y, dy, dx, dw, db, k, 1, 1, 1, 2, ws, bn=bn, bias=bias)
    g = dy0.double() * (y > 0)  # the mask of THIS build's output (checked above); a float64 mask flips at y ~ 0
    assert _rel(_np(db), _np(g.sum(dim=(0, 2, 3)))) <= TOL
    assert _rel(_np(dsc), _np((g * xhat).sum(dim=(0, 2, 3)))) <= TOL
    gs = g * v(scales.double())
    dmean = -gs.sum(dim=(0, 2, 3)) / torch.sqrt(var + 1e-5)
    dvar = -0.5 * (gs * (t64 - v(mean))).sum(dim=(0, 2, 3)) / (var * torch.sqrt(var) + 1e-5)
    gt = gs / torch.sqrt(v(var) + 1e-5) + v(dvar) * 2.0 * (t64 - v(mean)) / m + v(dmean) / m
    assert _rel(_np(dy), _np(gt)) <= TOL  # dy is rewritten in place with the pre-BN gradient
    # the conv gradients that follow consume that tensor: same result as the plain conv backward on it (whose
    # full-size behaviour the tests above pin against the oracle)
    dx2, dw2, db2 = torch.empty_like(x), torch.zeros_like(wt), torch.zeros_like(bias)
    ops.conv_backward(x, wt, t, dy.clone(), dx2, dw2, db2, k, 1, 1, 1, 0, ws, bias=bias)
    assert _rel(_np(dw), _np(dw2)) <= 1e-6
    assert _rel(_np(dx), _np(dx2)) <= 1e-6


def test_resnet18_stem_maxpool_n128_112():
    from bcnn_amd import ops
    n, c, hw, k, s = 128, 64, 112, 3, 2
    oh = (hw + s - 1) // s  # SAME
    x = _rand((n, c, hw, hw), 21)
    x[:, :, 10:14, 20:24] = 0.5  # plateaus: ties have to resolve to the first maximum
    y = torch.empty((n, c, oh, oh), device=DEV)
    idx = torch.empty((n, c, oh, oh), dtype=torch.int32, device=DEV)
    ops.maxpool_forward(x, y, idx, k, s)
    dy = _rand((n, c, oh, oh), 22)
    dx = torch.zeros_like(x)
    ops.maxpool_backward(dy, idx, dx, k, s)
    torch.cuda.synchronize()
    dx2 = torch.full_like(x, 9.0)  # the executor's no-fill mode: assign 0 + sums over whatever is there
    ops.maxpool_backward(dy, idx, dx2, k, s, overwrite=True)
    assert torch.equal(dx2.view(torch.int32), dx.view(torch.int32))  # bitwise, sign of zero included
    assert torch.equal(y, x.flatten()[idx.long().flatten()].view_as(y))  # gather identity, every element
    assert abs(float(dx.double().sum() - dy.double().sum())) <= 1e-6 * float(dy.double().abs().sum())
    for i in (0, 64, 127):  # bit-exact against the oracle: values, rebased indices, scatter order
        cs = dict(n=1, c=c, h=hw, w=hw, k=k, s=s, padding=0, x=_np(x[i:i + 1]), dx0=np.zeros((1, c, hw, hw), np.float32))
        exp = ob.orc_maxpool(cs, _np(dy[i:i + 1]))
        assert np.array_equal(_np(y[i:i + 1]), exp["y"])
        assert np.array_equal(_np(idx[i:i + 1]) - i * c * hw * hw, exp["indexes"])
        assert np.array_equal(_np(dx[i:i + 1]), exp["dx"])


def _depthwise_full_size(n, c, hw, k, stride, images, chunk):
    from bcnn_amd import ops
    oh, ow = ops.conv_out_hw(hw, hw, k, stride, 1)
    x = _rand((n, c, hw, hw), 31)
    wt = _rand((c * k * k,), 32, 0.3)
    bias = _rand((c,), 33, 0.1)
    dy0 = _rand((n, c, oh, ow), 34, 1e-2)
    y = torch.empty((n, c, oh, ow), device=DEV)
    ops.depthwise_forward(x, wt, bias, y, k, stride, 1, 2)
    dy, dx = dy0.clone(), torch.zeros_like(x)
    dw, db = torch.zeros_like(wt), torch.zeros_like(bias)
    ops.depthwise_backward(x, wt, y, dy, dx, dw, db, k, stride, 1, 2)
    torch.cuda.synchronize()
    for i in images:
        cs = dict(n=1, c=c, h=hw, w=hw, k=k, s=stride, p=1, act=2, input_grad=1, x=_np(x[i:i + 1]), wt=_np(wt),
                  bias=_np(bias), dy=_np(dy0[i:i + 1]), dw0=np.zeros(c * k * k, np.float32),
                  db0=np.zeros(c, np.float32), dx0=np.zeros((1, c, hw, hw), np.float32))
        exp = ob.orc_dw(cs)
        assert _rel(_np(y[i:i + 1]), exp["y"]) <= TOL
        assert _rel(_np(dx[i:i + 1]), exp["dx"]) <= TOL
    dw_sum, db_sum = torch.zeros_like(wt), torch.zeros_like(bias)
    for a in range(0, n, chunk):
        ops.depthwise_backward(x[a:a + chunk].contiguous(), wt, y[a:a + chunk].contiguous(), dy0[a:a + chunk].clone(),
                               torch.zeros_like(x[a:a + chunk]), dw_sum, db_sum, k, stride, 1, 2)
    assert _rel(_np(dw), _np(dw_sum)) <= TOL
    assert _rel(_np(db), _np(db_sum)) <= TOL
    # the executor's no-fill mode: dx = 0 + sums over whatever the buffer holds, bit-identical to accumulate-onto-zero
    dx2 = torch.full_like(x, 9.0)
    ops.depthwise_backward(x, wt, y, dy0.clone(), dx2, torch.zeros_like(wt), torch.zeros_like(bias), k, stride, 1, 2,
                           overwrite=True)
    assert torch.equal(dx2.view(torch.int32), dx.view(torch.int32))
    # one chunk's weight / bias gradient against the oracle
    a = n - 4
    cs = dict(n=4, c=c, h=hw, w=hw, k=k, s=stride, p=1, act=2, input_grad=1, x=_np(x[a:]), wt=_np(wt),
              bias=_np(bias), dy=_np(dy0[a:]), dw0=np.zeros(c * k * k, np.float32), db0=np.zeros(c, np.float32),
              dx0=np.zeros((4, c, hw, hw), np.float32))
    exp = ob.orc_dw(cs)
    dw1, db1 = torch.zeros_like(wt), torch.zeros_like(bias)
    ops.depthwise_backward(x[a:].contiguous(), wt, y[a:].contiguous(), dy0[a:].clone(), torch.zeros_like(x[a:]), dw1, db1,
                           k, stride, 1, 2)
    assert _rel(_np(dw1), exp["dw"]) <= TOL
    assert _rel(_np(db1), exp["db"]) <= TOL


def test_mobilenet_first_depthwise_n256_112():
    _depthwise_full_size(n=256, c=32, hw=112, k=3, stride=1, images=(0, 255), chunk=32)


def test_mobilenet_stride2_depthwise_n256_112_to_56():
    _depthwise_full_size(n=256, c=64, hw=112, k=3, stride=2, images=(0, 255), chunk=32)


def test_mobilenet_late_depthwise_n256_14_and_7():
    _depthwise_full_size(n=256, c=512, hw=14, k=3, stride=1, images=(17,), chunk=32)
    _depthwise_full_size(n=256, c=512, hw=14, k=3, stride=2, images=(18,), chunk=32)
    _depthwise_full_size(n=256, c=1024, hw=7, k=3, stride=1, images=(19,), chunk=32)


def test_activation_and_avgpool_at_conv_output_size():
    from bcnn_amd import ops
    n, c, hw = 128, 64, 224  # configs[1]'s output tensor, 1.6 GB
    x = _rand((n, c, hw, hw), 41)
    ref = x.clone()
    ops.activation_forward(x, 2)
    assert torch.equal(x, ref * (ref > 0))  # RELU is x*(x>0): -0.0 for negatives, like the reference
    once = x.clone()
    ops.activation_forward(x, 2)
    assert torch.equal(x, once)  # idempotent
    y = torch.empty((n, c, 1, 1), device=DEV)
    ops.avgpool_forward(ref, y)
    assert _rel(_np(y), _np(ref.double().mean(dim=(2, 3), keepdim=True))) <= TOL  # sequential fp32 sum of 50176 values in the reference
