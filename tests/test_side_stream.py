"""bcnn_backward queues the weight gradients of a pass on a second HIP stream of the library (DESIGN.md section 4.10,
include/bcnn_hip.h: bcnn_hip_conv_side_stream_mode, include/bcnn/bcnn.h: bcnn_set_weight_gradient_stream). What that must
not change: the gradients (against the same pass with every kernel on the caller's stream), their determinism under the
overlap, and what a gradient-ready callback may read -- a range it is told about has to be complete in the caller's
stream order, also when the weight gradients inside it were computed on the other stream."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu



def _trace(on):
    from bcnn_amd import _lib
    L = _lib.load()
    if on:
        L.bcnn_hip_trace_enable(1)
        return None
    n = L.bcnn_hip_trace_read(None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    L.bcnn_hip_trace_read(buf, n + 1)
    L.bcnn_hip_trace_enable(0)
    return set(buf.value.decode().split())


def _block_graph(net, ch):
    """stem + two residual blocks + a strided block: 3x3 layers wide enough for the fused Winograd weight-gradient kernel
    when ch >= 64, a 1x1 / s2 shortcut and a 3x3 / s2 layer for the LDS-DMA weight-gradient kernel, batch-norm everywhere"""
    from bcnn_amd import capi
    net.conv(ch, 3, 1, 1, 1, 1, capi.ACT_RELU, "input", "stem")
    src = "stem"
    for b in range(2):
        net.conv(ch, 3, 1, 1, 1, 1, capi.ACT_RELU, src, "b%d_c1" % b)
        net.conv(ch, 3, 1, 1, 1, 1, capi.ACT_NONE, "b%d_c1" % b, "b%d_c2" % b)
        net.eltwise(capi.ACT_RELU, "b%d_c2" % b, src, "b%d_out" % b)
        src = "b%d_out" % b
    net.conv(2 * ch, 3, 2, 1, 1, 1, capi.ACT_RELU, src, "s_c1")
    net.conv(2 * ch, 3, 1, 1, 1, 1, capi.ACT_NONE, "s_c1", "s_c2")
    net.conv(2 * ch, 1, 2, 0, 1, 1, capi.ACT_NONE, src, "s_ds")
    net.eltwise(capi.ACT_RELU, "s_c2", "s_ds", "s_out")
    net.avgpool("s_out", "gap")
    net.fullc(10, capi.ACT_NONE, "gap", "fc")
    net.softmax("fc", "sm")
    net.cost("sm", "label", "cost", 1.0)


def _tensors(net):
    out = []
    i = 0
    while net.L.bcnn_peek_tensor(net.net, i):
        out.append(i)
        i += 1
    return out


def _make(ch, n, hw, seed):
    from bcnn_amd import capi
    net = capi.Net(mode=capi.MODE_TRAIN, n=n, w=hw, h=hw, c=3, input_grad=True)
    _block_graph(net, ch)
    net.compile()
    rs = np.random.RandomState(seed)
    net.data(0)[...] = rs.uniform(-1, 1, net.shape(0)).astype(np.float32)
    lab = np.zeros(net.shape(1), np.float32)
    for b in range(n):
        lab[b, rs.randint(10)] = 1.0
    net.data(1)[...] = lab
    net.upload(0)
    net.upload(1)
    return net


def _pass(net, side):
    """one forward + backward from zeroed parameter gradients; every gradient tensor of the net, downloaded"""
    net.L.bcnn_set_weight_gradient_stream(net.net, side)
    idx = _tensors(net)
    for i in idx[2:]:
        g = net.grad(i)
        if g is not None:
            g[...] = 0.0
            net.upload(i, with_grad=True)
    net.forward()
    net.backward()
    net.sync()
    out = {}
    for i in idx:
        if net.grad(i) is not None:
            net.download(i)
            out[i] = net.grad(i).copy()
    return out


@pytest.mark.parametrize("ch,n,hw", [(8, 4, 16), (64, 16, 32)])
def test_gradients_do_not_depend_on_the_stream_of_the_weight_gradients(ch, n, hw):
    net = _make(ch, n, hw, seed=5)
    _trace(True)
    side = _pass(net, 1)
    kernels = _trace(False)
    assert any(k.startswith(("conv_dw", "wino_dw")) for k in kernels), kernels
    if ch >= 64:
        assert "wino_dw_fused_kernel" in kernels, kernels   # the kernel whose plan changes with the stream (192 of 256 CUs)
    again = _pass(net, 1)
    main = _pass(net, 0)
    assert side.keys() == main.keys() and len(side) > 20
    for i in side:
        # the overlap must not make a pass non-deterministic ...
        assert np.array_equal(side[i], again[i]), "tensor %d differs between two passes on the side stream" % i
        # ... and only the split of a K-split weight-gradient kernel (another fixed summation order) may differ from the
        # one-stream pass
        ref = np.abs(main[i]).max()
        err = np.abs(side[i] - main[i]).max()
        assert err <= 2e-6 * ref + 1e-12, "tensor %d: %g of %g between the two stream arrangements" % (i, err, ref)
    net.close()


def test_gradient_ready_callback_reads_complete_ranges_on_the_callers_stream():
    """inside the callback a copy of the reported range is queued on the CALLER's stream -- no host synchronisation -- and
    has to hold the final gradients, although the weight gradients of that range came from the library's second stream"""
    from bcnn_amd import _lib, capi
    L = _lib.load()
    stream = L.bcnn_hip_stream_create()
    L.bcnn_hip_set_stream(stream)
    try:
        net = _make(64, 16, 32, seed=9)
        net.set_data_parallel(0, 1)
        p, gsize = net.gradient_arena()
        arena = torch.as_tensor(capi.DeviceArray(p, gsize), device="cuda:0")
        ext = torch.cuda.ExternalStream(stream, device=torch.device("cuda:0"))
        snaps = []

        def ready(first, count):
            with torch.cuda.stream(ext):
                snaps.append((first, count, arena[first:first + count].clone()))

        net.L.bcnn_set_weight_gradient_stream(net.net, 1)
        net.forward()
        net.sync()
        arena.zero_()
        torch.cuda.synchronize()
        net.set_gradient_ready_callback(ready)
        net.backward()
        net.sync()
        torch.cuda.synchronize()
        net.set_gradient_ready_callback(None)
        assert len(snaps) >= 2
        final = arena.clone()
        assert float(final.abs().max()) > 0
        for first, count, snap in snaps:
            assert torch.equal(snap, final[first:first + count]), \
                "range [%d, %d) was reported before its gradients were complete on the caller's stream" % (first, first + count)
        net.close()
    finally:
        L.bcnn_hip_set_stream(None)
        L.bcnn_hip_stream_destroy(stream)
