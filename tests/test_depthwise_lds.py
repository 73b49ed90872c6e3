"""The LDS-staged 3x3 depthwise kernels (bcnn_amd/csrc/depthwise_lds.hip) against the oracle (oracle/bcnn_oracle.c
orc_dw_forward / orc_dw_backward, a restatement of bcnn_depthwise_conv_layer.c:165-293, :295-547) on shapes chosen to
reach every tiling case: planes cut into row bands (with ragged last band), one whole plane per workgroup, several
whole planes per workgroup (with a ragged last group), rows that are / are not a multiple of 16 bytes, odd sizes under
stride 2, and planes down to 1 x 1. Forward and the data gradient are the reference's own sums in the reference's tap
order (separate multiply and add): bit-exact. Weight / bias gradients are sums over the batch in a different (fixed)
order: 1e-4 relative, and bit-identical from run to run."""
import numpy as np
import pytest
import torch

from oracle import orc_bind as ob

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
ACT_NONE, ACT_TANH, ACT_RELU, ACT_RAMP, ACT_SOFTPLUS, ACT_LRELU, ACT_ABS, ACT_CLAMP, ACT_LOGISTIC = 0, 1, 2, 3, 4, 5, 6, 7, 9

# (n, c, h, w, stride, act)
SHAPES = [
    (2, 3, 112, 112, 1, ACT_RELU),    # four bands of 28 rows
    (2, 3, 112, 112, 2, ACT_RELU),    # four bands of 14 output rows
    (1, 2, 100, 72, 1, ACT_LRELU),    # ragged last band (100 = 54 + 46)
    (1, 2, 101, 72, 2, ACT_RELU),     # odd height under stride 2, bands
    (2, 5, 56, 56, 1, ACT_RELU),      # one whole plane per workgroup
    (2, 5, 56, 56, 2, ACT_NONE),
    (3, 7, 28, 28, 1, ACT_RELU),      # four planes per workgroup, 21 planes: ragged last group
    (3, 7, 28, 28, 2, ACT_RAMP),
    (2, 37, 14, 14, 1, ACT_RELU),     # rows of 56 bytes: per-element scatter
    (2, 37, 14, 14, 2, ACT_RELU),
    (3, 50, 7, 7, 1, ACT_RELU),       # 49-float planes
    (3, 50, 7, 7, 2, ACT_CLAMP),
    (1, 3, 33, 35, 1, ACT_RELU),      # odd width: nothing is 16-byte aligned
    (1, 3, 33, 35, 2, ACT_ABS),
    (2, 6, 9, 13, 2, ACT_RELU),
    (2, 4, 1, 1, 1, ACT_RELU),        # a single pixel: only the centre tap meets the image
    (2, 4, 2, 3, 2, ACT_NONE),
    (1, 1, 65, 64, 1, ACT_TANH),      # 4160 floats: just over one tile -> two bands
    (2, 3, 20, 20, 1, ACT_LOGISTIC),
    (2, 3, 20, 20, 1, ACT_SOFTPLUS),  # expensive derivative: separate pass, then the fused kernel
    (1, 2, 24, 300, 1, ACT_RELU),     # wide rows
    # rows too wide for the marching kernels, several bands per plane, an activation whose derivative is neither 0 nor 1: the
    # pre-pass over dy, then the LDS kernel whose bands read their neighbours' edge rows (the halo race)
    (1, 2, 24, 300, 1, ACT_LRELU),
    (1, 2, 25, 300, 2, ACT_LRELU),
    (1, 2, 24, 300, 1, ACT_SOFTPLUS),
    # 2 x 500: neither 3x3 family takes it (more than 64 column groups; an LDS image above 64 KB) -- the refusal has to come
    # BEFORE dy is prepared, or the generic path applies the derivative a second time
    (1, 2, 2, 500, 1, ACT_LRELU),
    (1, 3, 2, 508, 1, ACT_TANH),
    # depthwise_march.hip: an odd number of bands per plane (3: two march down, one up; the odd waves hold fewer bands),
    # ragged last band, both strides; width 6 (lanes of 2 columns, 21 bands per wave); width 5 at stride 1 (lanes of 1 column)
    (2, 3, 41, 40, 1, ACT_RELU),
    (2, 3, 83, 40, 2, ACT_RELU),
    (3, 5, 33, 6, 1, ACT_LRELU),
    (3, 5, 33, 6, 2, ACT_RELU),
    (2, 7, 30, 5, 1, ACT_RELU),
    # small planes in numbers that take several rounds of resident waves (3300 / 3334 waves of nine planes)
    (55, 540, 14, 14, 1, ACT_RELU),
    (300, 100, 7, 7, 1, ACT_RELU),
]


def _np(t):
    return t.detach().cpu().numpy()


def _rel(a, b):
    den = float(np.abs(b).max())
    d = float(np.abs(a.astype(np.float64) - b).max())
    return d if den == 0 else d / den


def _case(n, c, h, w, s, act, seed=0):
    rs = np.random.RandomState(1000 + seed)
    x = rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32)
    wt = rs.uniform(-0.5, 0.5, (c * 9,)).astype(np.float32)
    bias = rs.uniform(-0.2, 0.2, (c,)).astype(np.float32)
    bias[0] = 0.0  # the reference skips the add for 0 and 1 (bcnn_add_bias quirk)
    if c > 1:
        bias[1] = 1.0
    oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    dy = rs.uniform(-1, 1, (n, c, oh, ow)).astype(np.float32)
    dx0 = rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32)
    dw0 = rs.uniform(-1, 1, (c * 9,)).astype(np.float32)
    db0 = rs.uniform(-1, 1, (c,)).astype(np.float32)
    return dict(n=n, c=c, h=h, w=w, k=3, s=s, p=1, act=act, input_grad=1, x=x, wt=wt, bias=bias, dy=dy, dx0=dx0,
                dw0=dw0, db0=db0)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_c%d_%dx%d_s%d_act%d" % s)
def test_forward_and_backward_against_the_oracle(shape):
    from bcnn_amd import ops
    cs = _case(*shape)
    exp = ob.orc_dw(cs)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in cs.items() if isinstance(v, np.ndarray)}
    y = torch.full(exp["y"].shape, 7.0, device=DEV)
    ops.depthwise_forward(t["x"], t["wt"], t["bias"], y, 3, cs["s"], 1, cs["act"])
    exact_act = cs["act"] not in (ACT_TANH, ACT_SOFTPLUS, ACT_LOGISTIC)  # exp() in double on both sides, libm vs ocml
    if exact_act:
        assert np.array_equal(_np(y), exp["y"])
    else:
        assert _rel(_np(y), exp["y"]) <= 1e-6
    # backward on the oracle's own forward output: accumulate onto the given dx / dw / db, dy rewritten in place
    yt = torch.from_numpy(exp["y"]).to(DEV)
    dy, dx, dw, db = t["dy"].clone(), t["dx0"].clone(), t["dw0"].clone(), t["db0"].clone()
    ops.depthwise_backward(t["x"], t["wt"], yt, dy, dx, dw, db, 3, cs["s"], 1, cs["act"])
    if exact_act:
        assert np.array_equal(_np(dy), exp["dy_out"])
        assert np.array_equal(_np(dx), exp["dx"])
    else:
        assert _rel(_np(dy), exp["dy_out"]) <= 1e-6
        assert _rel(_np(dx), exp["dx"]) <= 1e-6
    assert _rel(_np(dw) - cs["dw0"], exp["dw"] - cs["dw0"]) <= TOL
    assert _rel(_np(db) - cs["db0"], exp["db"] - cs["db0"]) <= TOL
    # the executor's no-fill mode: dx = 0 + sums, whatever the buffer held
    cz = dict(cs)
    cz["dx0"] = np.zeros_like(cs["dx0"])
    expz = ob.orc_dw(cz)
    dx2 = torch.full_like(t["x"], 9.0)
    dw2, db2 = t["dw0"].clone(), t["db0"].clone()
    ops.depthwise_backward(t["x"], t["wt"], yt, t["dy"].clone(), dx2, dw2, db2, 3, cs["s"], 1, cs["act"], overwrite=True)
    if exact_act:
        assert np.array_equal(_np(dx2), expz["dx"])
    else:
        assert _rel(_np(dx2), expz["dx"]) <= 1e-6
    # run-to-run determinism of the two-level sums
    assert torch.equal(dw2.view(torch.int32), dw.view(torch.int32))
    assert torch.equal(db2.view(torch.int32), db.view(torch.int32))


def test_non_finite_inputs_stay_in_their_own_pixels():
    """an Inf in x reaches exactly the outputs whose window covers it; zero padding does not turn it into NaN"""
    from bcnn_amd import ops
    n, c, h, w = 1, 2, 28, 28
    x = torch.zeros((n, c, h, w), device=DEV)
    x[0, 1, 0, 0] = float("inf")
    wt = torch.ones(c * 9, device=DEV)
    y = torch.empty_like(x)
    ops.depthwise_forward(x, wt, torch.zeros(c, device=DEV), y, 3, 1, 1, ACT_NONE)
    yn = _np(y)
    assert np.isinf(yn[0, 1, :2, :2]).all() and np.isfinite(yn[0, 0]).all()
    mask = np.ones((h, w), bool)
    mask[:2, :2] = False
    assert (yn[0, 1][mask] == 0).all()


# ---------------------------------------------------------------------------------------------------
# the depthwise layer + stand-alone batch-norm pair (MobileNet block): shared-work entry points against the
# two separate workers
# ---------------------------------------------------------------------------------------------------
PAIR_SHAPES = [(4, 8, 112, 112, 1), (4, 8, 112, 112, 2), (3, 6, 56, 56, 1), (5, 7, 28, 28, 2), (6, 37, 14, 14, 1),
               (6, 37, 14, 14, 2), (8, 50, 7, 7, 1), (2, 3, 33, 35, 1), (2, 5, 9, 13, 2), (300, 100, 7, 7, 1)]


@pytest.mark.parametrize("shape", PAIR_SHAPES, ids=lambda s: "n%d_c%d_%dx%d_s%d" % s)
def test_depthwise_batchnorm_pair_shares_work_without_changing_results(shape):
    from bcnn_amd import capi, ops
    n, c, h, w, s = shape
    act = ACT_RELU
    assert ops.depthwise_bn_fusable(n, c, h, w, 3, s, 1, act)
    cs = _case(n, c, h, w, s, act, seed=7)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in cs.items() if isinstance(v, np.ndarray)}
    oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    rs = np.random.RandomState(77)
    scales = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32)).to(DEV)
    bnb = torch.from_numpy(rs.uniform(-0.3, 0.3, c).astype(np.float32)).to(DEV)
    mode = capi.MODE_TRAIN

    def forward(fused):
        y = torch.empty((n, c, oh, ow), device=DEV)
        z = torch.empty_like(y)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        sm, sv = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        if fused:
            stats = torch.empty(ops.depthwise_stats_size(n, c, h, w, 3, s, 1), device=DEV)
            splits = ops.depthwise_forward_stats(t["x"], t["wt"], t["bias"], y, 3, s, 1, act, stats)
            assert splits > 0
            ops.batchnorm_forward_stats(y, z, rm, rv, scales, bnb, sm, sv, y, mode, stats, splits)
        else:
            ops.depthwise_forward(t["x"], t["wt"], t["bias"], y, 3, s, 1, act)
            ops.batchnorm_forward(y, z, rm, rv, scales, bnb, sm, sv, y, mode)
        return y, z, rm, rv, sm, sv

    y0, z0, rm0, rv0, sm0, sv0 = forward(False)
    y1, z1, rm1, rv1, sm1, sv1 = forward(True)
    assert torch.equal(y0, y1)
    # the statistics are the same sums in a different (fixed) order
    for a, b in ((sm0, sm1), (sv0, sv1), (rm0, rm1), (rv0, rv1)):
        assert _rel(_np(b), _np(a)) <= 1e-5
    assert _rel(_np(z1), _np(z0)) <= 1e-5
    assert np.array_equal(_np(y0), ob.orc_dw(cs)["y"])

    # backward, both routes on the SAME saved statistics: the same operations in the same order -> the same bits
    dz = torch.from_numpy(rs.uniform(-1, 1, (n, c, oh, ow)).astype(np.float32)).to(DEV)

    def backward(fused, overwrite):
        dx = torch.full_like(t["x"], 3.0) if overwrite else t["dx0"].clone()
        dw, db = t["dw0"].clone(), t["db0"].clone()
        dsc, dbb = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        dm, dv = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        if fused:
            ops.batchnorm_backward_sums(dz, scales, dsc, dbb, sm0, sv0, dm, dv, y0)
            ops.depthwise_backward_bn(t["x"], t["wt"], y0, dz, dx, dw, db, 3, s, 1, act, overwrite, sm0, sv0, scales, dm, dv)
        else:
            g = dz.clone()
            dy = torch.empty_like(g)
            ops.batchnorm_backward(g, dy, scales, dsc, dbb, sm0, sv0, dm, dv, y0)
            ops.depthwise_backward(t["x"], t["wt"], y0, dy, dx, dw, db, 3, s, 1, act, overwrite=overwrite)
        return dx, dw, db, dsc, dbb, dm, dv

    for overwrite in (False, True):
        a = backward(False, overwrite)
        b = backward(True, overwrite)
        for u, v in zip(a, b):
            assert torch.equal(u.view(torch.int32), v.view(torch.int32))

