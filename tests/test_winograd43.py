"""Winograd F(4x4, 3x3) fused forward / dX kernel (bcnn_amd/csrc/conv_winograd43.hip; the reference's precedent for a
transformed-domain 3x3 convolution is bcnn_conv_layer.c:388-436 on bcnn_mat.c:1403-2138). The product library takes it for
3x3 / s1 layers whose planes are whole 4 x 4 tiles and that fill the chip (the 56 x 56 and 28 x 28 stages of ResNet-18:
covered at benchmark size by tests/test_full_size_properties.py and the ResNet parity tests). Here the EXPERIMENT build
forces it (BCNN_HIP_WINOGRAD43=1) on small shapes -- one tile per image, ragged last unit (tiles not a multiple of 32),
ragged channel blocks (M not a multiple of 32), images that end inside a half-wave, 24 / 40 input channels -- and compares
the batch-norm-fused forward (raw output + statistics from the kernel's epilogue), and dX, with the oracle at 1e-4 per
tensor AND element-wise (1e-4 |ref| + 1e-5 max|ref|)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.environ.get("BCNN_WINO43_CHILD") == "1"

SHAPES = [  # n, c, h, w, f, act, bn
    (2, 16, 8, 8, 32, 2, 1),
    (3, 24, 12, 8, 72, 0, 1),     # ragged channel blocks (72 = 32 + 32 + 8), 24 input channels
    (2, 64, 16, 16, 64, 2, 1),
    (1, 64, 28, 28, 128, 2, 1),   # 49 tiles: a ragged second unit
    (2, 40, 4, 4, 16, 0, 1),      # one tile per image
    (5, 32, 8, 12, 48, 2, 1),     # 30 tiles: one ragged unit; images end inside the half-wave
    (3, 32, 20, 20, 64, 5, 1),    # 75 tiles: three units
    (2, 16, 8, 8, 64, 0, 0),      # no batch-norm: the forward stays on the other kernels, dX runs here
    (4, 128, 8, 8, 128, 2, 1),    # 16 chunks of 8 channels
    # planes that are not whole 4 x 4 tiles: the last tile row / column hangs over (rows / columns beyond the plane read as
    # zeros, are not stored and stay out of the statistics); rows are only 4-byte aligned
    (3, 32, 14, 14, 64, 2, 1),    # 4 x 4 tiles cover 16 x 16
    (5, 64, 7, 7, 96, 2, 1),      # 2 x 2 tiles cover 8 x 8, 7-float rows
    (2, 16, 9, 13, 40, 0, 1),     # one row / one column of overhang, ragged channel block
    (2, 24, 6, 5, 32, 5, 1),      # two columns of the last tile column exist... W = 5: tiles of 4 + 1 columns
    (3, 16, 3, 3, 16, 0, 0),      # a single partial tile per image (dX only)
]


def _case(n, c, h, w, f, act, bn, seed):
    from oracle import orc_bind as ob
    rs = np.random.RandomState(seed)
    cs = dict(op="conv", n=n, c=c, h=h, w=w, f=f, k=3, s=1, p=1, g=1, bn=bn, act=act, mode=ob.MODE_TRAIN, input_grad=1,
              x=rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32),
              wt=(rs.uniform(-1, 1, (f, c, 3, 3)) * (3.0 / (c * 9)) ** 0.5).astype(np.float32),
              bias=rs.uniform(-0.3, 0.3, f).astype(np.float32),
              dy=(rs.uniform(-1, 1, (n, f, h, w)) * 1e-2).astype(np.float32))
    if bn:
        cs.update(run_mean0=rs.uniform(-0.1, 0.1, f).astype(np.float32), run_var0=rs.uniform(0.5, 1.5, f).astype(np.float32),
                  scales=rs.uniform(0.5, 1.5, f).astype(np.float32))
    return cs


@pytest.mark.gpu
@pytest.mark.skipif(not CHILD, reason="runs in the child process that test_winograd43_forced_on_small_shapes spawns")
@pytest.mark.parametrize("shape", SHAPES)
def test_child_winograd43_matches_oracle(shape):
    from oracle import orc_bind as ob
    from tests import _golden as G
    from tests import _hip_cases as HC
    cs = _case(*shape, seed=sum(shape))
    got = HC.run_hip(cs)
    want = ob.run_oracle(cs)
    for key in sorted(want):
        if key in got:
            if key in ("saved_var", "run_var", "dvar"):   # E[x^2] - E[x]^2 cancels: abs + rel like tests/test_hip_parity.py
                assert np.allclose(got[key], want[key], rtol=1e-4, atol=1e-6), (shape, key)
                continue
            if key in ("db",) and shape[6]:               # analytically zero behind a batch-norm: rounding noise on both sides
                continue
            G.assert_close("%s/%s" % (shape, key), got[key], want[key], 1e-4)


@pytest.mark.gpu
def test_winograd43_forced_on_small_shapes():
    exp = os.path.join(ROOT, "bcnn_amd", "lib", "libbcnn_hip_exp.so")
    assert os.path.exists(exp), "experiment build missing: __graft_entry__.build() makes it"
    e = dict(os.environ, BCNN_WINO43_CHILD="1", BCNN_HIP_LIB=exp, BCNN_HIP_WINOGRAD43="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p",
                        "no:cacheprovider", "-k", "child"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "%d passed" % len(SHAPES) in r.stdout, r.stdout[-2000:]


# ---- product path: shapes inside the product rule (whole 4 x 4 tiles, >= 64 channels, >= 512 units of 32 channels x 32 tiles)
# whose unit counts leave a K-split tail: (n, c, f, h, w). Raw forward with batch-norm statistics from the kernel's epilogue /
# the fix-up kernel, and dX, against torch's float64 convolution. The bar is the parity bar's 1e-4; measured ~1e-5 (F(4x4,3x3)
# in fp32: tools/exp/wino43_error.py), asserted at 3e-5 so that a regression in the transforms shows.
PRODUCT_SHAPES = [
    (128, 64, 64, 32, 32),    # 512 units: two full rounds, no tail
    (72, 128, 128, 32, 32),   # 576 units: 64 tail units x 16 chunks dealt out as pieces of 4 chunks (never across units)
    (84, 128, 128, 32, 32),   # 672 units: 160 tail units, pieces of 10 chunks -- most of them span two units
    (70, 128, 80, 32, 32),    # 80 output channels: three channel blocks, the last one ragged; pieces of 11 chunks
    (35, 64, 128, 56, 56),    # the 56 x 56 stage's plane: 196 tiles per image, images end inside units
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", PRODUCT_SHAPES, ids=lambda s: "n%d_c%d_f%d_%dx%d" % s)
def test_product_path_with_tail_against_float64(shape):
    import torch
    import torch.nn.functional as F
    from bcnn_amd import ops
    DEV = "cuda:0"
    n, c, f, h, w = shape
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.rand((n, c, h, w), device=DEV, generator=gen) * 2 - 1
    wt = (torch.rand((f, c, 3, 3), device=DEV, generator=gen) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
    b = torch.rand(f, device=DEV, generator=gen) - 0.5
    Z = lambda: torch.zeros(f, device=DEV)

    def forward():
        bn = dict(run_mean=Z(), run_var=Z() + 1, scales=torch.rand(f, device=DEV, generator=gen) + 0.5, saved_mean=Z(),
                  saved_var=Z(), workspace=torch.full((n, f, h, w), float("nan"), device=DEV))
        y = torch.empty((n, f, h, w), device=DEV)
        ops.conv_forward(x, wt, b, y, 3, 1, 1, 1, 0, bn=bn)  # TRAIN: raw output + statistics from the kernel's epilogue
        torch.cuda.synchronize()
        return bn

    bn = forward()
    raw = F.conv2d(x.double().cpu(), wt.double().cpu(), None, padding=1)
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / max(float(r.abs().max()), 1e-30))
    assert rel(bn["workspace"], raw) <= 3e-5
    mean = raw.mean(dim=(0, 2, 3))
    var = (raw * raw).mean(dim=(0, 2, 3)) - mean * mean
    assert rel(bn["saved_mean"], mean) <= 3e-5 and rel(bn["saved_var"], var) <= 1e-4
    bn2 = forward()
    assert torch.equal(bn["workspace"], bn2["workspace"]) and torch.equal(bn["saved_mean"], bn2["saved_mean"])
    assert torch.equal(bn["saved_var"], bn2["saved_var"])
    # dX: no batch-norm, no activation -> dy is used as given
    y = torch.empty((n, f, h, w), device=DEV)
    dy = (torch.rand((n, f, h, w), device=DEV, generator=gen) * 2 - 1) * 0.1
    dx = torch.full_like(x, float("nan"))
    dw, db = torch.zeros_like(wt), torch.zeros(f, device=DEV)
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, 3, 1, 1, 1)), device=DEV)
    ops.conv_backward(x, wt, y, dy.clone(), dx, dw, db, 3, 1, 1, 1, 0, ws)
    torch.cuda.synchronize()
    dxr = F.conv_transpose2d(dy.double().cpu(), wt.double().cpu(), None, padding=1)
    assert rel(dx, dxr) <= 3e-5
    dx2 = torch.full_like(x, float("nan"))
    ops.conv_backward(x, wt, y, dy.clone(), dx2, torch.zeros_like(wt), torch.zeros(f, device=DEV), 3, 1, 1, 1, 0, ws)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
