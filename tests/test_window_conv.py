"""Window-in-LDS kernels for 3x3 / stride-1 convolutions with at most three input channels per group
(bcnn_amd/csrc/conv_window.hip: BASELINE configs[1] and every layer shaped like it). The benchmark size is covered by
tests/test_full_size_properties.py::test_configs1_conv3x3_n128_224; here small and ragged shapes pin the parts that size
never reaches -- partial strips (OH not a multiple of the strip height), partial 32-pixel tiles and 16-pixel windows,
half windows (OW % 16 == 8), filters that do not fill the 32-row MFMA tiles, one / two / three channels per group (three
different reduction orders, with and without a free slot for the bias row), groups, padding 0 / 1 / 2, the wide-pitch
instantiation, bias == 1.0 (quirk 2), fused activations, accumulation onto a gradient carry -- against the oracle
(reference bcnn_conv_layer.c:367-587) at the usual 1e-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [  # n, c, h, w, f, p, g, act
    (2, 3, 16, 16, 64, 1, 1, 0),      # one partial strip pair (16 = 8 + 8), one partial tile
    (3, 3, 19, 24, 64, 1, 1, 2),      # OH = 19: strips of 8, 8, 3 (forward) and 7, 7, 5 (dW); OW = 24: half window
    (2, 3, 9, 40, 20, 1, 1, 5),       # 20 filters: one ragged 32-row tile; two tiles per row, the second ragged
    (2, 3, 12, 72, 40, 1, 1, 0),      # 40 filters: the second 32-row tile is ragged; 72 = 2 full tiles + 8 pixels
    (2, 1, 10, 16, 8, 1, 1, 2),       # one input channel: 5 reduction steps, bias in the free half-step
    (2, 2, 10, 16, 33, 1, 1, 0),      # two input channels: 9 tap steps + a step for the bias alone
    (2, 6, 11, 32, 64, 1, 2, 2),      # two groups of three channels, 32 filters each
    (2, 4, 8, 16, 12, 1, 4, 0),       # depthwise-like through the conv node: four groups of one channel, three filters each
    (1, 3, 12, 20, 16, 0, 1, 0),      # VALID: OW = 18 is not a multiple of 8 -> forward on the window kernel, dW on the old one
    (1, 3, 10, 24, 16, 2, 1, 2),      # padding 2: the output is larger than the input (OW = 26)
    (2, 3, 6, 240, 64, 1, 1, 0),      # wide rows: the 264-float pitch instantiation (OW = 240)
    (1, 3, 224, 224, 64, 1, 1, 0),    # configs[1] geometry, one image
]


def _case(n, c, h, w, f, p, g, act, seed, carry):
    from oracle import orc_bind as ob
    rs = np.random.RandomState(seed)
    oh, ow = h + 2 * p - 2, w + 2 * p - 2
    cg = c // g
    cs = dict(op="conv", n=n, c=c, h=h, w=w, f=f, k=3, s=1, p=p, g=g, bn=0, act=act, mode=ob.MODE_TRAIN, input_grad=0,
              x=rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32),
              wt=(rs.uniform(-1, 1, (f, cg, 3, 3)) * (3.0 / (cg * 9)) ** 0.5).astype(np.float32),
              bias=rs.uniform(-0.3, 0.3, f).astype(np.float32),
              dy=(rs.uniform(-1, 1, (n, f, oh, ow)) * 1e-2).astype(np.float32))
    cs["bias"][1] = 1.0  # quirk 2: bcnn_add_scalar skips exactly 1.0f
    if carry:            # beta = 1: the gradients are added onto what the buffers hold (momentum carry)
        cs["dw0"] = rs.uniform(-1, 1, cs["wt"].shape).astype(np.float32)
        cs["db0"] = rs.uniform(-1, 1, f).astype(np.float32)
    return cs


@pytest.mark.parametrize("shape", SHAPES)
def test_window_kernels_match_oracle(shape):
    from oracle import orc_bind as ob
    from tests import _golden as G
    from tests import _hip_cases as HC
    cs = _case(*shape, seed=sum(shape), carry=(shape[0] == 2))
    got = HC.run_hip(cs)
    want = ob.run_oracle(cs)
    for key in ("y", "dy_out", "dw", "db"):
        err = G.rel_err(got[key], want[key])
        assert err <= 1e-4, (shape, key, err)


def test_window_forward_is_deterministic_and_exactly_linear():
    import torch
    from bcnn_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(3)
    x = torch.rand((4, 3, 40, 64), device="cuda:0", generator=g) * 2 - 1
    wt = torch.rand((64, 3, 3, 3), device="cuda:0", generator=g) - 0.5
    zero = torch.zeros(64, device="cuda:0")
    ys = [torch.empty((4, 64, 40, 64), device="cuda:0") for _ in range(3)]
    ops.conv_forward(x, wt, zero, ys[0], 3, 1, 1, 1, 0)
    ops.conv_forward(x, wt, zero, ys[1], 3, 1, 1, 1, 0)
    ops.conv_forward(x * 2, wt, zero, ys[2], 3, 1, 1, 1, 0)
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1])
    assert torch.equal(ys[2], ys[0] * 2)
    # Inf in the input stays where the reference puts it: only outputs whose 3x3 patch contains the pixel are affected
    x[1, 2, 17, 30] = float("inf")
    ops.conv_forward(x, wt, zero, ys[1], 3, 1, 1, 1, 0)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(ys[1])
    assert bad[1, :, 16:19, 29:32].all()
    bad[1, :, 16:19, 29:32] = False
    assert not bad.any()


# ---- 7x7 / stride-2 stem kernel (conv_fwd_stem_kernel) -----------------------------------------------------------------
STEM_SHAPES = [  # n, c, h, w, f, p, g, act, bn
    (2, 3, 32, 32, 64, 3, 1, 2, 0),     # OW = 16: a tile spans two output rows
    (2, 3, 32, 32, 64, 3, 1, 2, 1),     # fused batch-norm: raw output + statistics from the kernel's epilogue
    (3, 3, 30, 36, 40, 3, 1, 5, 0),     # OH = 15: strips of 4, 4, 4, 3; 40 filters: the second block of 32 is ragged
    (2, 3, 22, 28, 20, 2, 1, 0, 1),     # padding 2, one ragged block of filters, statistics
    (2, 1, 20, 24, 8, 3, 1, 2, 0),      # one input channel: 25 steps, bias in the free half-step
    (2, 2, 20, 24, 33, 3, 1, 0, 0),     # two input channels: 49 tap steps + a step for the bias alone
    (2, 6, 18, 20, 64, 3, 2, 2, 1),     # two groups of three channels
    (1, 3, 16, 16, 64, 0, 1, 0, 0),     # VALID
    (1, 3, 224, 224, 64, 3, 1, 2, 1),   # the ResNet-18 stem, one image
    # OW % 4 == 0: the weight gradient runs on conv_dw_stem_kernel too (the shapes above fall back to the LDS-DMA GEMM for it)
    (3, 3, 26, 40, 40, 3, 1, 5, 0),     # OW = 20 (two 16-pixel windows, the second ragged), OH = 13, ragged filters
    (2, 6, 18, 24, 64, 3, 2, 2, 1),     # OW = 12, two groups, fused batch-norm (no bias gradient from the conv)
    (5, 1, 14, 32, 16, 3, 1, 0, 0),     # one input channel: 49 taps, ones column 49; 5 images x 7 rows over 35 workgroups
    (2, 2, 12, 16, 64, 2, 1, 2, 0),     # two input channels, padding 2
]


@pytest.mark.parametrize("shape", STEM_SHAPES)
def test_stem_kernel_matches_oracle(shape):
    from oracle import orc_bind as ob
    from tests import _golden as G
    from tests import _hip_cases as HC
    n, c, h, w, f, p, g, act, bn = shape
    rs = np.random.RandomState(sum(shape))
    oh, ow = (h + 2 * p - 7) // 2 + 1, (w + 2 * p - 7) // 2 + 1
    cg = c // g
    cs = dict(op="conv", n=n, c=c, h=h, w=w, f=f, k=7, s=2, p=p, g=g, bn=bn, act=act, mode=ob.MODE_TRAIN, input_grad=0,
              x=rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32),
              wt=(rs.uniform(-1, 1, (f, cg, 7, 7)) * (3.0 / (cg * 49)) ** 0.5).astype(np.float32),
              bias=rs.uniform(-0.3, 0.3, f).astype(np.float32),
              dy=(rs.uniform(-1, 1, (n, f, oh, ow)) * 1e-2).astype(np.float32))
    cs["bias"][1] = 1.0
    if n != 2:   # beta = 1: gradients are added onto the carry
        cs["dw0"] = rs.uniform(-1, 1, cs["wt"].shape).astype(np.float32)
        cs["db0"] = rs.uniform(-1, 1, f).astype(np.float32)
    if bn:
        cs.update(run_mean0=rs.uniform(-0.1, 0.1, f).astype(np.float32), run_var0=rs.uniform(0.5, 1.5, f).astype(np.float32),
                  scales=rs.uniform(0.5, 1.5, f).astype(np.float32))
    got = HC.run_hip(cs)
    want = ob.run_oracle(cs)
    for key in sorted(want):
        if key in got:
            err = G.rel_err(got[key], want[key])
            assert err <= 1e-4, (shape, key, err)
