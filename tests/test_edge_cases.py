"""Edge cases of the C-ABI that no fixture can carry: empty batches (every entry point must return without
touching memory) and a convolution whose tensors are large enough to use every kernel tier of the dispatcher
against a plain fp32 restatement in torch (cross-check independent of the oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_batch_is_a_no_op():
    import torch
    from bcnn_amd import ops
    dev = "cuda:0"
    x = torch.empty((0, 4, 8, 8), device=dev)
    wt = torch.ones((6, 4, 3, 3), device=dev)
    b = torch.zeros(6, device=dev)
    y = torch.empty((0, 6, 8, 8), device=dev)
    dy = torch.empty_like(y)
    dx = torch.empty_like(x)
    dw = torch.full_like(wt, 2.0)
    db = torch.full_like(b, 3.0)
    ws = torch.zeros(16, device=dev)
    ops.conv_forward(x, wt, b, y, 3, 1, 1, 1, 2)
    ops.conv_backward(x, wt, y, dy, dx, dw, db, 3, 1, 1, 1, 2, ws)
    idx = torch.empty((0, 4, 4, 4), dtype=torch.int32, device=dev)
    p = torch.empty((0, 4, 4, 4), device=dev)
    ops.maxpool_forward(x, p, idx, 2, 2)
    ops.maxpool_backward(p, idx, dx, 2, 2)
    ops.depthwise_forward(x, torch.ones((4, 3, 3), device=dev), torch.zeros(4, device=dev), torch.empty_like(x), 3, 1, 1, 2)
    torch.cuda.synchronize()
    assert float(dw.min()) == 2.0 and float(db.max()) == 3.0  # gradients untouched


@pytest.mark.parametrize("shape", [
    dict(n=5, c=72, h=19, w=17, f=136, k=3, s=1, p=1),    # ragged M / J / columns on the LDS-DMA GEMMs
    dict(n=3, c=96, h=15, w=15, f=64, k=3, s=2, p=1),     # stride-parity classes of unequal size
    dict(n=4, c=64, h=9, w=9, f=192, k=1, s=1, p=0),      # pointwise
])
def test_conv_against_torch_fp32(shape):
    import torch
    import torch.nn.functional as F
    from bcnn_amd import ops
    dev = "cuda:0"
    n, c, h, w, f, k, s, p = (shape[q] for q in "nchwfksp")
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand((n, c, h, w), device=dev, generator=g) * 2 - 1
    wt = (torch.rand((f, c, k, k), device=dev, generator=g) * 2 - 1) * (3.0 / (c * k * k)) ** 0.5
    b = torch.rand(f, device=dev, generator=g) - 0.5
    oh, ow = ops.conv_out_hw(h, w, k, s, p)
    y = torch.empty((n, f, oh, ow), device=dev)
    ops.conv_forward(x, wt, b, y, k, s, p, 1, 0)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    torch.backends.cudnn.allow_tf32 = False
    yr = F.conv2d(xr.double(), wr.double(), b.double(), stride=s, padding=p)
    dy = (torch.rand(y.shape, device=dev, generator=g) * 2 - 1) * 0.1
    yr.backward(dy.double())
    dx = torch.empty_like(x)
    dw = torch.zeros_like(wt)
    db = torch.zeros_like(b)
    ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, k, s, p, 1)), device=dev)
    dyc = dy.clone()
    ops.conv_backward(x, wt, y, dyc, dx, dw, db, k, s, p, 1, 0, ws)
    torch.cuda.synchronize()

    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    assert rel(y, yr.detach()) < 1e-5, rel(y, yr.detach())
    assert rel(dx, xr.grad.double()) < 1e-5
    assert rel(dw, wr.grad.double()) < 1e-5
    assert rel(db, dy.double().sum((0, 2, 3))) < 1e-5


def test_tensor_above_2gib_takes_the_fallback_kernels_and_agrees_with_the_dma_kernels_on_halves():
    """Maximum sizes: the LDS-DMA kernels address operands through 32-bit buffer offsets and hand tensors of
    2 GiB or more to the register-staged kernels. A 2.1 GiB input is convolved whole (fallback path) and as
    two half batches (DMA path); forward, dX and dW must agree -- a cross-check between the two kernel families
    at a size no CPU reference finishes in test time."""
    import torch
    from bcnn_amd import ops
    dev = "cuda:0"
    n, c, h, w, f, k, s, p = 128, 64, 256, 256, 40, 3, 1, 1
    assert n * c * h * w * 4 >= 2**31
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.rand((n, c, h, w), device=dev, generator=g) * 2 - 1
    wt = (torch.rand((f, c, k, k), device=dev, generator=g) * 2 - 1) * (3.0 / (c * k * k)) ** 0.5
    b = torch.rand(f, device=dev, generator=g) - 0.5
    dy = (torch.rand((n, f, h, w), device=dev, generator=g) * 2 - 1) * 0.1

    def run(xs, dys):
        nn = xs.shape[0]
        y = torch.empty((nn, f, h, w), device=dev)
        ops.conv_forward(xs, wt, b, y, k, s, p, 1, 0)
        dx = torch.empty_like(xs)
        dw = torch.zeros_like(wt)
        db = torch.zeros_like(b)
        ws = torch.zeros(max(1, ops.conv_workspace_size(nn, c, h, w, f, k, s, p, 1)), device=dev)
        ops.conv_backward(xs, wt, y, dys.clone(), dx, dw, db, k, s, p, 1, 0, ws)
        torch.cuda.synchronize()
        return y, dx, dw, db

    y, dx, dw, db = run(x, dy)
    half = n // 2
    parts = [run(x[i:i + half].contiguous(), dy[i:i + half].contiguous()) for i in (0, half)]

    def rel(a, r):
        return float((a - r).abs().max() / r.abs().max())
    assert rel(y, torch.cat([q[0] for q in parts])) < 1e-5
    assert rel(dx, torch.cat([q[1] for q in parts])) < 1e-5
    assert rel(dw, parts[0][2] + parts[1][2]) < 1e-4
    assert rel(db, parts[0][3] + parts[1][3]) < 1e-4


@pytest.mark.parametrize("hw", [(7, 8), (5, 12), (16, 16), (9, 72), (3, 260)])
def test_maxpool_3x3_s2_backward_pair_kernel_small_and_ragged(hw):
    """SAME-padded 3x3 / stride-2 pooling with OW == W / 2: the pair-load backward kernel (pool.hip) on odd heights, rows
    shorter / longer than a wave, ties -- bit-exact against the oracle, accumulate and assign-zero-plus-sums modes"""
    import torch
    from bcnn_amd import ops
    from oracle import orc_bind as ob
    h, w = hw
    n, c, k, s = 3, 5, 3, 2
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    rs = np.random.RandomState(h * 1000 + w)
    x = rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32)
    x[:, :, : min(h, 4), 2:6] = 0.25  # a plateau: ties resolve to the first maximum
    dy = rs.uniform(-1, 1, (n, c, oh, ow)).astype(np.float32)
    dx0 = rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32)
    xg, dyg = torch.from_numpy(x).cuda(), torch.from_numpy(dy).cuda()
    y = torch.empty((n, c, oh, ow), device="cuda")
    idx = torch.empty((n, c, oh, ow), dtype=torch.int32, device="cuda")
    ops.maxpool_forward(xg, y, idx, k, s)
    dx = torch.from_numpy(dx0).cuda()
    ops.maxpool_backward(dyg, idx, dx, k, s)
    dxo = torch.full((n, c, h, w), 7.0, device="cuda")
    ops.maxpool_backward(dyg, idx, dxo, k, s, overwrite=True)
    torch.cuda.synchronize()
    exp = ob.orc_maxpool(dict(n=n, c=c, h=h, w=w, k=k, s=s, padding=0, x=x, dx0=dx0.copy()), dy)
    assert np.array_equal(idx.cpu().numpy(), exp["indexes"])
    assert np.array_equal(dx.cpu().numpy(), exp["dx"])
    expz = ob.orc_maxpool(dict(n=n, c=c, h=h, w=w, k=k, s=s, padding=0, x=x, dx0=np.zeros_like(dx0)), dy)
    assert np.array_equal(dxo.cpu().numpy().view(np.int32), expz["dx"].view(np.int32))
