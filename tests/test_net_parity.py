"""GPU parity at the bcnn_net / bcnn_node level: the same graph is built through the PUBLIC C API on
 (1) the unmodified reference (oracle/_ref/libbcnn_ref.so, CPU, driven by oracle/ref_bind.py) and
 (2) this repo's libbcnn.so (C99 host + HIP back-end, driven by bcnn_amd/capi.py),
with identical parameters and inputs; every tensor's data and gradient is compared after
forward / backward / SGD update. This is the drop-in check of SURVEY.md section 8b."""
import numpy as np
import pytest
import torch  # noqa: F401  (first, so that one HIP runtime serves torch and libbcnn_hip.so)

from oracle import ref_bind as rb
from tests import _golden as G

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4


def _need_ref():
    if not rb.available():
        pytest.skip("oracle/_ref/libbcnn_ref.so not present (built from /root/reference by oracle/Makefile)")


def stack_graph(net):
    """conv(3->16, relu) -> BN -> maxpool 3/2 SAME -> avgpool (the SURVEY.md section 8c agreement stack)"""
    net.conv(16, 3, 1, 1, 1, 0, rb.ACT_RELU, "input", "c1")
    net.batchnorm("c1", "bn1")
    net.maxpool(3, 2, rb.PADDING_SAME, "bn1", "p1")
    net.avgpool("p1", "avg")


def resnet_block_graph(net):
    """stem conv+BN+relu, a residual block with identity shortcut, a down-sampling block with the 1x1/s2
    projection (quirk 1) -- the topology of examples/cifar10/cifar10_example.c:65-143 in miniature --
    then avgpool, fc, softmax, cost."""
    net.conv(8, 3, 1, 1, 1, 1, rb.ACT_RELU, "input", "stem")
    net.conv(8, 3, 1, 1, 1, 1, rb.ACT_RELU, "stem", "b1c1")
    net.conv(8, 3, 1, 1, 1, 1, rb.ACT_NONE, "b1c1", "b1c2")
    net.eltwise(rb.ACT_RELU, "stem", "b1c2", "b1")
    net.conv(16, 3, 2, 1, 1, 1, rb.ACT_RELU, "b1", "b2c1")
    net.conv(16, 3, 1, 1, 1, 1, rb.ACT_NONE, "b2c1", "b2c2")
    net.conv(16, 1, 2, 0, 1, 1, rb.ACT_NONE, "b1", "b2p")
    net.eltwise(rb.ACT_RELU, "b2p", "b2c2", "b2")
    net.avgpool("b2", "avg")
    net.fullc(10, rb.ACT_NONE, "avg", "fc")
    net.softmax("fc", "sm")
    net.cost("sm", "label", "cost", 1.0)


def lenet_graph(net):
    """examples/mnist/mnist_example.c:30-55"""
    net.conv(8, 3, 1, 1, 1, 0, rb.ACT_RELU, "input", "conv1")
    net.batchnorm("conv1", "bn1")
    net.maxpool(2, 2, rb.PADDING_SAME, "bn1", "pool1")
    net.conv(8, 3, 1, 1, 1, 0, rb.ACT_RELU, "pool1", "conv2")
    net.batchnorm("conv2", "bn2")
    net.maxpool(2, 2, rb.PADDING_SAME, "bn2", "pool2")
    net.fullc(32, rb.ACT_RELU, "pool2", "fc1")
    net.batchnorm("fc1", "bn3")
    net.fullc(10, rb.ACT_RELU, "bn3", "fc2")
    net.softmax("fc2", "softmax")
    net.cost("softmax", "label", "cost", 1.0)


def depthwise_graph(net):
    """MobileNet-v1 unit (BASELINE configs[4]): depthwise 3x3 (fused relu) -> BN -> pointwise 1x1 conv (+BN+relu)"""
    net.conv(8, 3, 2, 1, 1, 1, rb.ACT_RELU, "input", "stem")
    net.depthwise(3, 1, 1, rb.ACT_RELU, "stem", "dw1")
    net.batchnorm("dw1", "dwbn1")
    net.conv(16, 1, 1, 0, 1, 1, rb.ACT_RELU, "dwbn1", "pw1")
    net.depthwise(3, 2, 1, rb.ACT_RELU, "pw1", "dw2")
    net.conv(16, 1, 1, 0, 1, 1, rb.ACT_RELU, "dw2", "pw2")
    net.avgpool("pw2", "avg")


def depthwise_small_planes_graph(net):
    """the tail of MobileNet-v1 in miniature: depthwise layers on 28 x 28, 14 x 14 and 7 x 7 planes, both strides, every one
    between a 1x1 convolution with batch-norm (whose apply sweep the depthwise kernel takes over) and a stand-alone
    batch-norm (whose backward it applies): the marching kernels with lanes of 4, 2 and 1 columns and both fusions"""
    net.conv(8, 3, 2, 1, 1, 1, rb.ACT_RELU, "input", "stem")          # 56 -> 28
    src = "stem"
    for i, (width, stride) in enumerate(((16, 1), (16, 2), (24, 1), (24, 2), (32, 1)), start=1):
        net.depthwise(3, stride, 1, rb.ACT_RELU, src, "dw%d" % i)      # planes 28, 28 -> 14, 14, 14 -> 7, 7
        net.batchnorm("dw%d" % i, "dwbn%d" % i)
        net.conv(width, 1, 1, 0, 1, 1, rb.ACT_RELU, "dwbn%d" % i, "pw%d" % i)
        src = "pw%d" % i
    net.avgpool(src, "avg")


def linear_bottleneck_graph(net):
    """MobileNet-v2 style tails around a FOLDED batch-norm: [depthwise] -> [batchnorm] -> [1x1 conv + BN,
    no activation] -> eltwise (the convolution's backward rides on the eltwise node: bcnn_hip_conv_backward_residual), and the
    same chain in front of a 3x3 / s2 max-pooling (bcnn_hip_conv_backward_bn_done). In both branches the weight gradient has
    to be formed against the batch-norm's INPUT with the fold's column factors, not against the batch-norm's (unwritten)
    output tensor. 40 channels: the LDS-DMA GEMM that takes the fold wants more than 32 filters."""
    net.conv(40, 3, 1, 1, 1, 1, rb.ACT_RELU, "input", "stem")
    net.depthwise(3, 1, 1, rb.ACT_RELU, "stem", "dw1")
    net.batchnorm("dw1", "dwbn1")
    net.conv(40, 1, 1, 0, 1, 1, rb.ACT_NONE, "dwbn1", "pw1")
    net.eltwise(rb.ACT_RELU, "stem", "pw1", "res1")
    net.depthwise(3, 1, 1, rb.ACT_RELU, "res1", "dw2")
    net.batchnorm("dw2", "dwbn2")
    net.conv(48, 1, 1, 0, 1, 1, rb.ACT_RELU, "dwbn2", "pw2")
    net.maxpool(3, 2, rb.PADDING_SAME, "pw2", "p2")
    net.avgpool("p2", "avg")


def prelu_node_graph(net):
    """a stand-alone PReLU activation node (the only stand-alone activation the reference's CPU build can run,
    bcnn_activation_layer.c:148-163): its slopes are stepped with batch_size = weights->n = 1, not the net's
    batch (bcnn_activation_layer.c:262-291) -- two SGD steps pin that divisor and the decay term"""
    net.conv(8, 3, 1, 1, 1, 0, rb.ACT_NONE, "input", "c1")
    net.activation(rb.ACT_PRELU, "c1")
    net.maxpool(2, 2, rb.PADDING_SAME, "c1", "p1")
    net.conv(8, 3, 1, 1, 1, 1, rb.ACT_RELU, "p1", "c2")
    net.avgpool("c2", "avg")


GRAPHS = {
    "prelu_node": (prelu_node_graph, dict(w=12, h=12, c=3, n=4), False),
    "stack": (stack_graph, dict(w=16, h=12, c=3, n=3), False),
    "resnet_block": (resnet_block_graph, dict(w=16, h=16, c=3, n=4), True),
    # n = 16: bn3 normalises fc outputs over the batch only; with 4 samples per channel its statistics amplify
    # 1e-7 rounding noise (the reference's own results move with heap alignment of its AVX loops) to ~1e-4
    "lenet": (lenet_graph, dict(w=12, h=12, c=1, n=16), True),
    "mobilenet_unit": (depthwise_graph, dict(w=16, h=16, c=3, n=2), False),
    "mobilenet_small_planes": (depthwise_small_planes_graph, dict(w=56, h=56, c=3, n=4), False),
    "linear_bottleneck": (linear_bottleneck_graph, dict(w=16, h=16, c=3, n=4), False),
}

# graphs whose point is a particular fused path: the dispatch trace (include/bcnn_hip.h) must show it ran
EXPECT_TRACE = {
    "linear_bottleneck": ["bnfold:fwd", "bnfold:dw", "maxpool_fwd_s2_bn_kernel", "maxpool_bwd_pair_bn_kernel"],
    "mobilenet_small_planes": ["dwm_fwd_kernel:bnin", "dwm_bwd_kernel:bn+bnin"],  # (<= 32 filters: no fold here)
}


ZERO_FLOOR = {
    # absolute slack for values that are analytically zero -- the bias gradient of a layer in front of a batch-norm: both
    # sides hold the rounding noise of a sum over N * H * W terms, which grows with the square root of the count (3136
    # terms per channel on this graph's 28 x 28 planes against <= 512 on the others)
    "mobilenet_small_planes": 4e-7,
    "linear_bottleneck": 4e-7,  # 1024 terms per channel, gradients of O(0.1)
}


def _compare(tag, a, b, tol=REL_TOL, floor=1e-7):
    assert a.shape == b.shape, (tag, a.shape, b.shape)
    # |a-b| <= tol*max|b| + 1e-7: gradients that are analytically zero (e.g. the bias of a BN feeding
    # another BN) are ~1e-9 noise on both sides and carry no relative information
    a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    diff = float(np.max(np.abs(a64 - b64))) if a64.size else 0.0
    bound = tol * float(np.max(np.abs(b64))) + floor if b64.size else 0.0
    assert diff <= bound, "%s: max abs diff %.3g > %.3g (rel %.3g)" % (tag, diff, bound, G.rel_err(a, b))


@pytest.mark.parametrize("gname", sorted(GRAPHS))
def test_net_matches_reference(gname):
    _need_ref()
    from bcnn_amd import capi
    build, shp, has_cost = GRAPHS[gname]
    rs = np.random.RandomState(7)
    # the builders of BOTH libraries draw their Xavier weights from libc rand(): seed it, so the parameters (copied
    # from the reference below) are the same in every run instead of depending on what else called rand() before
    import ctypes
    ctypes.CDLL(None).srand(20240607)
    ref = rb.RefNet(mode=rb.MODE_TRAIN, **shp)
    ref.L.ref_set_threads(ref.net, 4)  # the reference oversubscribes badly with one OpenMP thread per core
    hip = capi.Net(mode=capi.MODE_TRAIN, **shp)
    build(ref)
    build(hip)
    ref.compile()
    hip.compile()
    ref.L.bcnn_set_sgd_optimizer(ref.net, 0.01, 0.9)
    ref.L.bcnn_set_weight_regularizer(ref.net, 5e-4)
    hip.set_sgd(0.01, 0.9, 5e-4)
    nt = ref.L.ref_num_tensors(ref.net)
    names = [ref.L.ref_tensor_name(ref.net, i).decode() for i in range(nt)]
    # identical parameters: take the reference's (rand()-initialised) values; perturb BN params so they matter
    for i in range(2, nt):
        d = ref.data(i)
        nm = names[i]
        if nm.endswith("_scales"):
            d[...] = rs.uniform(0.5, 1.5, d.shape)
        elif nm.endswith("_b"):
            d[...] = rs.uniform(-0.2, 0.2, d.shape)
        elif nm.endswith("_run_var"):
            d[...] = rs.uniform(0.5, 1.5, d.shape)
        elif "prelu" in nm:
            d[...] = rs.uniform(0.1, 0.3, d.shape)
        assert hip.shape(i) == ref.shape(i), (nm, hip.shape(i), ref.shape(i))
        hip.data(i)[...] = d
        hip.upload(i)
    x = rs.uniform(-1, 1, ref.shape(0)).astype(np.float32)
    ref.data(0)[...] = x
    hip.data(0)[...] = x
    hip.upload(0)
    if has_cost:
        lab = np.zeros(ref.shape(1), np.float32)
        for b in range(lab.shape[0]):
            lab[b, rs.randint(lab.shape[1])] = 1.0
        ref.data(1)[...] = lab
        hip.data(1)[...] = lab
        hip.upload(1)
    last = nt - 1
    from bcnn_amd import _lib
    _lib.load().bcnn_hip_trace_enable(1)
    for it in range(2):  # two steps: the second one runs on updated weights and the momentum carry
        ref.forward()
        hip.forward()
        if not has_cost:
            dy = (rs.uniform(-1, 1, ref.shape(last)) * 0.1).astype(np.float32)
            ref.grad(last)[...] = dy
            hip.download(last)
            hip.grad(last)[...] = dy
            hip.upload(last, with_grad=True)
        ref.backward()
        hip.backward()
        for i in range(nt):
            if not ref.tensor(i).data:
                continue  # the label tensor has no storage in graphs without a cost node
            hip.download(i)
            _compare("%s it%d %s data" % (gname, it, names[i]), hip.data(i), ref.data(i))
            if ref.grad(i) is not None and i != 1:
                _compare("%s it%d %s grad" % (gname, it, names[i]), hip.grad(i), ref.grad(i),
                         floor=ZERO_FLOOR.get(gname, 1e-7))
        ref.L.bcnn_update(ref.net)
        hip.update()
        for i in range(2, nt):
            hip.download(i)
            _compare("%s it%d %s data after update" % (gname, it, names[i]), hip.data(i), ref.data(i))
    tl = _lib.load().bcnn_hip_trace_read(None, 0)
    tbuf = ctypes.create_string_buffer(tl + 1)
    _lib.load().bcnn_hip_trace_read(tbuf, tl + 1)
    _lib.load().bcnn_hip_trace_enable(0)
    ran = set(tbuf.value.decode().split())
    missing = [k for k in EXPECT_TRACE.get(gname, []) if k not in ran]
    assert not missing, "%s: %s did not run (trace: %s)" % (gname, missing, sorted(ran))
    # bit-exact pooling indices
    nn = ref.L.ref_num_nodes(ref.net)
    import torch
    for node in range(nn):
        if ref.L.ref_node_type(ref.net, node) == 5:  # BCNN_LAYER_MAXPOOL
            want = ref.maxpool_indexes(node)
            ptr = hip.node_state(node, 0)
            got = torch.empty(want.shape, dtype=torch.int32, device="cuda:0")
            from bcnn_amd import _lib
            _lib.load().bcnn_hip_memcpy_d2d(got.data_ptr(), ptr, got.numel() * 4)
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy(), want), "maxpool indexes differ in node %d" % node
    ref.close()
    hip.close()


def test_fused_prelu_net_forward_matches_reference():
    """bcnn_add_convolutional_layer(..., BCNN_ACT_PRELU, ...) with and without the fused batch-norm (slopes = src slot
    3 + 3 * batch_norm, bcnn_conv_layer.c:188-198, :476-481). Forward only, TRAIN mode: the reference's backward through
    such a node accumulates slope gradients through a NULL pointer (the tensor is created without a gradient buffer, :192;
    bcnn_activation_layer.c:214) -- the backward is pinned on the oracle (tests/test_hip_parity.py)."""
    _need_ref()
    import ctypes
    from bcnn_amd import capi
    ctypes.CDLL(None).srand(20240608)
    shp = dict(w=12, h=10, c=3, n=3)
    ref = rb.RefNet(mode=rb.MODE_TRAIN, **shp)
    ref.L.ref_set_threads(ref.net, 4)
    hip = capi.Net(mode=capi.MODE_TRAIN, **shp)
    for net in (ref, hip):
        net.conv(8, 3, 1, 1, 1, 1, rb.ACT_PRELU, "input", "c1")
        net.conv(40, 3, 2, 1, 1, 0, rb.ACT_PRELU, "c1", "c2")
        net.conv(48, 1, 1, 0, 1, 1, rb.ACT_PRELU, "c2", "c3")
        net.avgpool("c3", "avg")
        net.compile()
    rs = np.random.RandomState(3)
    nt = ref.L.ref_num_tensors(ref.net)
    names = [ref.L.ref_tensor_name(ref.net, i).decode() for i in range(nt)]
    assert sum("prelu_slopes" in nm for nm in names) == 3
    for i in range(2, nt):
        if not ref.tensor(i).data:
            continue
        d = ref.data(i)
        if "prelu" in names[i]:
            d[...] = rs.uniform(0.05, 0.5, d.shape)
        elif names[i].endswith("_scales"):
            d[...] = rs.uniform(0.5, 1.5, d.shape)
        elif names[i].endswith("_b"):
            d[...] = rs.uniform(-0.2, 0.2, d.shape)
        assert hip.shape(i) == ref.shape(i), names[i]
        hip.data(i)[...] = d
        hip.upload(i)
    x = rs.uniform(-1, 1, ref.shape(0)).astype(np.float32)
    ref.data(0)[...] = x
    hip.data(0)[...] = x
    hip.upload(0)
    ref.forward()
    hip.forward()
    for i in range(nt):
        if not ref.tensor(i).data:
            continue
        hip.download(i, False)
        _compare("fused prelu %s" % names[i], hip.data(i), ref.data(i))
    ref.close()
    hip.close()
