"""GPU parity tests: the HIP path (through the C-ABI of include/bcnn_hip.h) against
 (1) the golden vectors taken from the unmodified reference (tests/golden), and
 (2) the CPU oracle on larger seeded inputs.
Bars (BASELINE.json north_star): pooling indices bit-exact; conv / batch-norm within 1e-4 relative
(max |a-b| / max |b| per tensor); everything else elementwise-tight."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4          # conv / batchnorm / depthwise / gemm tensors
ACT_TOL = 2e-6          # activation map (exp/log evaluated in double on both sides)
VAR_KEYS = {"saved_var", "run_var", "dvar"}  # E[x^2]-E[x]^2 cancels: compare with abs + rel tolerance


def _check(name, got, exp, tol):
    for key, want in exp.items():
        if key == "dy":
            continue
        have = got[key]
        assert have.shape == want.shape, (name, key, have.shape, want.shape)
        if want.dtype.kind == "i":
            assert np.array_equal(have, want), "%s/%s: indices differ" % (name, key)
            continue
        assert np.array_equal(np.isnan(have), np.isnan(want)), (name, key, "NaN pattern")
        h, w = np.nan_to_num(have), np.nan_to_num(want)
        if key in VAR_KEYS:
            assert np.allclose(h, w, rtol=1e-4, atol=1e-6), (name, key, np.abs(h - w).max())
        else:
            # per-tensor norm AND the element-wise bar |a - b| <= tol |b| + tol / 10 max|b| (worst offender named)
            G.assert_close("%s/%s" % (name, key), h, w, tol, rtol=tol, afrac=tol / 10)


@pytest.mark.parametrize("name", G.names())
def test_hip_matches_reference_golden(name):
    from tests import _hip_cases as HC
    case, exp = G.load(name)
    got = HC.run_hip(case, exp)
    op = str(case["op"])
    if op == "maxpool":
        # bit-exact: values, int32 indices, and the backward scatter (same addition order)
        assert np.array_equal(got["indexes"], exp["indexes"])
        assert np.array_equal(got["y"], exp["y"], equal_nan=True)
        assert np.array_equal(got["dx"], exp["dx"], equal_nan=True)
        return
    _check(name, got, exp, ACT_TOL if op == "act" else REL_TOL)


# ---- convolution with a FUSED PReLU, backward (bcnn_conv_layer.c:188-198 builder slot 3 + 3 bn, :476-481 forward, :517-521
# backward with slope gradients). The reference cannot supply these vectors: it creates the slopes tensor without a gradient
# buffer and its backward accumulates the slope gradients through that NULL pointer (oracle/ref_cases.py: make_conv) -- the
# forward results are golden fixtures (conv_*prelu*_fwd, conv_k5_prelu_predict), the backward is pinned on the oracle, whose PReLU
# map and derivative are themselves pinned by the act_8 fixture.
PRELU_BWD = [  # n, c, h, w, f, k, s, p, bn
    (2, 3, 8, 8, 8, 3, 1, 1, 0),
    (2, 3, 8, 8, 8, 3, 1, 1, 1),
    (3, 32, 10, 10, 64, 3, 1, 1, 1),   # the LDS-DMA GEMM and its epilogue
    (2, 16, 9, 9, 40, 5, 1, 2, 0),
    (2, 64, 8, 8, 64, 1, 1, 0, 1),     # 1x1 raw view
]


@pytest.mark.parametrize("shape", PRELU_BWD, ids=lambda s: "n%d_c%d_%dx%d_f%d_k%d_s%d_p%d_bn%d" % s)
def test_conv_fused_prelu_backward_matches_oracle(shape):
    from oracle import orc_bind as ob
    from tests import _hip_cases as HC
    n, c, h, w, f, k, s, p, bn = shape
    rs = np.random.RandomState(sum(shape))
    u = lambda shp, lo=-1.0, hi=1.0: rs.uniform(lo, hi, shp).astype(np.float32)
    oh, ow = ob.conv_out_hw(h, w, k, s, p)
    a = np.sqrt(3.0 / (c * k * k))
    cs = dict(op="conv", n=n, c=c, h=h, w=w, f=f, k=k, s=s, p=p, g=1, bn=bn, act=8, input_grad=1, mode=ob.MODE_TRAIN,
              x=u((n, c, h, w)), wt=u((f, c, k, k), -a, a), bias=u((f,), -0.5, 0.5), dy=u((n, f, oh, ow)) * np.float32(0.1),
              slopes=u((f,), 0.05, 0.5), dslopes0=u((f,)) * np.float32(0.1),
              dw0=u((f, c, k, k)) * np.float32(0.05), db0=u((f,)) * np.float32(0.05))
    if bn:
        cs.update(run_mean0=u((f,)) * np.float32(0.1), run_var0=u((f,), 0.5, 1.5), scales=u((f,), 0.5, 1.5),
                  dscales0=u((f,)) * np.float32(0.05))
    got = HC.run_hip(cs)
    want = ob.run_oracle(cs)
    assert "dslopes" in want and "dslopes" in got
    _check("conv_prelu_bwd%s" % (shape,), got, want, REL_TOL)
