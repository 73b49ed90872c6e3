"""Graph-level parity AT THE BENCHMARK'S DISPATCH. The small graphs of tests/test_teacher_forced.py and
tests/test_net_parity.py run planes of 24 x 24 and less: the size-dependent rules send them to other kernels than the ones
bench.py times (wino43_wanted needs >= 256 units; the stem / pooling pair kernels want 16-byte rows of 112 floats). Here the
two benchmark graphs run at 224 x 224 with a batch large enough that the product rules pick the product kernels -- asserted
with the dispatch trace of include/bcnn_hip.h (bcnn_hip_trace_*), not assumed -- against the unmodified reference
(oracle/_ref/libbcnn_ref.so) through the public bcnn_net API on both sides:

 1. teacher-forced walk (tests/test_teacher_forced.py::_walk): every node worker alone on the reference's own inputs, forward and
    backward, 1e-4 per tensor + the element-wise bar. The single-node workers never fuse across nodes, so this pins the KERNELS
    (F(4x4,3x3) forward / dX with their K-split tails, the stem kernels, the marching depthwise kernels, ...).
 2. whole-pass forward, reference fed by the student: bcnn_forward runs the fused pass (stem convolution -> pooling pair,
    convolution -> eltwise, batch-norm folded into the 1x1 convolution, depthwise kernels that normalise their input on the
    fly); then, node by node, the reference's worker gets THIS build's input tensors of that node and has to produce this
    build's outputs -- per-node bar again (1e-4 + element-wise), nothing hides behind 20 stacked batch-norms, and what is
    compared is what the pass-level host linking (bcnn_link_*, apply_skipped, data_pending, prepack jobs) produced.
 3. whole-pass backward against this build's own node-by-node walk from the same forward state: bcnn_backward (the pooling /
    batch-norm backward pair, residual sweeps, sums left by consumers, the fold's dW) must reproduce the gradients of the
    unfused workers that (1) pins against the reference at this very size.

Reference: bcnn_net.c:410-429 (executor), examples/cifar10/cifar10_example.c:65-143 (the residual graph)."""
import ctypes

import numpy as np
import pytest

from oracle import ref_bind as rb
from tests import _golden as G
from tests import test_teacher_forced as TF

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _trace_start():
    from bcnn_amd import _lib
    _lib.load().bcnn_hip_trace_enable(1)


def _trace_stop():
    from bcnn_amd import _lib
    L = _lib.load()
    n = L.bcnn_hip_trace_read(None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    L.bcnn_hip_trace_read(buf, n + 1)
    L.bcnn_hip_trace_enable(0)
    names = buf.value.decode().split()
    counts = {}
    for k in names:
        counts[k] = counts.get(k, 0) + 1
    return counts


def _need(counts, wanted, what):
    missing = [k for k in wanted if counts.get(k, 0) == 0]
    assert not missing, "%s: the product kernels %s did not run; the trace holds %s" % (what, missing, counts)
    print("%s ran on: %s" % (what, ", ".join("%s x%d" % kv for kv in sorted(counts.items()))))


RESNET = dict(graph="build_resnet18", shape=dict(w=224, h=224, c=3, n=48), kw=dict(base=64))
MOBILENET = dict(graph="build_mobilenet_v1", shape=dict(w=224, h=224, c=3, n=32), kw={})

# single-node workers: the kernels themselves
RESNET_WALK_KERNELS = ["wino43b_kernel:fwd", "wino43b_kernel:dx", "wino43b_tail_fixup", "wino_fused_kernel:fwd",
                       "wino_fused_kernel:dx", "wino43_dw_kernel", "wino_dw_fused_kernel", "conv_fwd_stem_kernel", "conv_dw_stem_kernel",
                       "conv_igemm_dma_kernel:fwd", "conv_igemm_dma_kernel:dx", "conv_dw_dma_kernel"]
MOBILENET_WALK_KERNELS = ["dwm_fwd_kernel", "dwm_bwd_kernel", "conv_igemm_dma_kernel:fwd", "conv_igemm_dma_kernel:dx",
                          "conv_dw_dma_kernel"]
# whole passes: the fused pairs on top
RESNET_PASS_FWD = ["conv_fwd_stem_kernel", "maxpool_fwd_s2_bn_kernel", "wino43b_kernel:fwd", "wino43b_tail_fixup",
                   "wino_fused_kernel:fwd", "conv_igemm_dma_kernel:fwd"]
RESNET_PASS_BWD = ["maxpool_bwd_pair_bn_kernel", "conv_dw_stem_kernel", "wino43b_kernel:dx", "wino43b_tail_fixup",
                   "wino_fused_kernel:dx", "wino43_dw_kernel", "wino_dw_fused_kernel", "conv_igemm_dma_kernel:dx", "conv_dw_dma_kernel"]
MOBILENET_PASS_FWD = ["dwm_fwd_kernel:bnin", "bnfold:fwd", "conv_igemm_dma_kernel:fwd"]
MOBILENET_PASS_BWD = ["dwm_bwd_kernel:bn+bnin", "bnfold:dw", "conv_igemm_dma_kernel:dx+bnsums", "conv_dw_dma_kernel"]


def test_resnet18_224_teacher_forced_walk_on_the_product_kernels():
    _trace_start()
    worst, fp64 = TF._walk(RESNET["graph"], RESNET["shape"], 10, **RESNET["kw"])
    _need(_trace_stop(), RESNET_WALK_KERNELS, "ResNet-18 224x224 N=48 teacher-forced walk")
    assert fp64  # stage 4 crosses the reference's gemm limits (quirk 8): those tensors went to float64 + the oracle


def test_mobilenet_v1_224_teacher_forced_walk_on_the_product_kernels():
    _trace_start()
    TF._walk(MOBILENET["graph"], MOBILENET["shape"], 10, **MOBILENET["kw"])
    _need(_trace_stop(), MOBILENET_WALK_KERNELS, "MobileNet-v1 224x224 N=32 teacher-forced walk")


class _Checker:
    def __init__(self, tol=TOL):
        self.tol = tol
        self.worst = (0.0, "")
        self.worst_elem = (0.0, "")

    def __call__(self, a, b, what, floor=TF.ABS_FLOOR):
        err = TF._rel(a, b, floor)
        if err > self.worst[0]:
            self.worst = (err, what)
        assert err <= self.tol, (what, err)
        a64, b64 = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
        bound = G.ELEM_RTOL * np.abs(b64) + G.ELEM_AFRAC * float(np.abs(b64).max()) + floor
        ratio = np.abs(a64 - b64) / bound
        j = int(np.argmax(ratio))
        if ratio[j] > self.worst_elem[0]:
            self.worst_elem = (float(ratio[j]), what)
        assert ratio[j] <= 1.0, (what, "element %d: %.9g against %.9g = %.2f x its bound" % (j, a64[j], b64[j], ratio[j]))


def _full_pass(cfg, want_fwd, want_bwd, label):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    ref, hip, names, convs = TF._build(cfg["graph"], cfg["shape"], 10, **cfg["kw"])
    nn, nt = ref.num_nodes(), len(names)
    # identical parameters and inputs on the device (TF._build left them in the reference)
    for t in range(nt):
        if ref.tensor(t).data:
            hip.data(t)[...] = ref.data(t)
            hip.upload(t)
    # ---- 2. the fused forward pass, then the reference node by node on THIS build's inputs ---------------------------------
    _trace_start()
    hip.forward()
    _need(_trace_stop(), want_fwd, label + " bcnn_forward")
    for t in range(nt):
        if ref.tensor(t).data:
            hip.download(t, False)  # produces what the fused pass did not write (bcnn_materialize_data)
    chk = _Checker()
    produced = {0, 1}  # the net input, the label and every node output: what a node reads from OTHER nodes (the rest are its parameters)
    for i in range(nn):
        produced.add(TF._node_tensors(ref, i)[1][0])
    for i in range(nn):
        src, dst = TF._node_tensors(ref, i)
        for t in src:
            if t in produced and ref.tensor(t).data:
                ref.data(t)[...] = hip.data(t)
        ref.forward_node(i)
        for t in dst + [t for t in src[1:] if t not in produced]:  # outputs, and the running statistics the node moved
            chk(hip.data(t), ref.data(t), "node %d %s" % (i, names[t]))
    print("%s forward pass, reference fed node by node with this build's inputs: worst deviation %.2e (%s), worst element %.3f "
          "of its bound (%s)" % (label, chk.worst[0], chk.worst[1], chk.worst_elem[0], chk.worst_elem[1]))

    # ---- 3. the fused backward pass against this build's own unfused walk from the same state ---------------------------------
    grad_ids = [t for t in range(nt) if hip.grad(t) is not None and ref.tensor(t).data]
    def zero_grads():
        for t in grad_ids:
            hip.grad(t)[...] = 0
            hip.upload(t, True)
    zero_grads()
    hip.forward()
    _trace_start()
    hip.backward()
    _need(_trace_stop(), want_bwd, label + " bcnn_backward")
    fused = {}
    for t in grad_ids:
        hip.download(t, True)
        fused[t] = hip.grad(t).copy()
    zero_grads()
    hip.forward_node(nn - 1)  # the cost node forms its gradient in its FORWARD worker (bcnn_cost_layer.c), which the zero fill took
    # ... from the SAME forward state (the fused pass's: what it did not write is produced on demand by the single-node workers,
    # bcnn_materialize_data). A second, unfused forward would differ in the last bits, flip ReLU masks of elements at the kink
    # and the difference would be amplified through ~20 batch-norm layers: that is the drift tests/test_resnet18_parity.py reports.
    for i in range(nn - 1, -1, -1):
        hip.backward_node(i)
    chk2 = _Checker()
    bias_of = {}  # bias tensor -> the node output whose gradient it sums per channel
    for i in range(nn):
        src, dst = TF._node_tensors(ref, i)
        for t in src[2:]:  # a convolution's / depthwise layer's src[2], a stand-alone batch-norm's src[4]
            if names[t].endswith("_b"):
                bias_of[t] = dst[0]
    for t in grad_ids:
        hip.download(t, True)
        floor = TF.ABS_FLOOR
        if t in bias_of and bias_of[t] in fused and fused[bias_of[t]].ndim == 4:
            # a bias gradient is a per-channel sum of the node's output gradient, analytically zero in front of a batch-norm:
            # rounding noise of that sum on both sides (tests/test_teacher_forced.py::_sum_floor)
            floor = max(floor, TF._sum_floor(fused[bias_of[t]]))
        chk2(fused[t], hip.grad(t), "d(%s)" % names[t], floor=floor)
    print("%s backward pass against the node-by-node walk of the same build: worst deviation %.2e (%s), worst element %.3f of "
          "its bound (%s)" % (label, chk2.worst[0], chk2.worst[1], chk2.worst_elem[0], chk2.worst_elem[1]))
    ref.close()
    hip.close()


def test_resnet18_224_whole_passes_on_the_product_dispatch():
    _full_pass(RESNET, RESNET_PASS_FWD, RESNET_PASS_BWD, "ResNet-18 224x224 N=48")


def test_mobilenet_v1_224_whole_passes_on_the_product_dispatch():
    _full_pass(MOBILENET, MOBILENET_PASS_FWD, MOBILENET_PASS_BWD, "MobileNet-v1 224x224 N=32")
