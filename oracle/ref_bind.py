"""ctypes binding of oracle/_ref/libbcnn_ref.so -- the UNMODIFIED reference built by oracle/Makefile.

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_golden.py and bench.py's
cpu_baseline leg. The product (bcnn_amd/) never imports this module.

The reference's public C API (inc/bcnn/bcnn.h:285-1043) is called exactly as a user of bcnn would;
oracle/ref_driver.c adds accessors for layer-private state.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libbcnn_ref.so")

# enums, inc/bcnn/bcnn.h:105-112, 164-175, 201-205, 238-242
MODE_PREDICT, MODE_TRAIN, MODE_VALID = 0, 1, 2
(ACT_NONE, ACT_TANH, ACT_RELU, ACT_RAMP, ACT_SOFTPLUS, ACT_LRELU, ACT_ABS, ACT_CLAMP, ACT_PRELU,
 ACT_LOGISTIC) = range(10)
PADDING_SAME, PADDING_VALID, PADDING_CAFFE = 0, 1, 2
FILLER_FIXED, FILLER_XAVIER, FILLER_MSRA = 0, 1, 2
LOG_SILENT = 3
LOSS_EUCLIDEAN = 0
METRIC_ERROR_RATE = 0


class Tensor(C.Structure):
    """struct bcnn_tensor (CPU build), inc/bcnn/bcnn.h:242-255."""
    _fields_ = [("n", C.c_int), ("c", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("has_grad", C.c_int), ("name", C.c_char_p),
                ("data", C.POINTER(C.c_float)), ("grad_data", C.POINTER(C.c_float))]


def available():
    return os.path.exists(REF_SO)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(REF_SO)
    vp, i, f, cp = C.c_void_p, C.c_int, C.c_float, C.c_char_p
    fp = C.POINTER(C.c_float)
    L.bcnn_init_net.argtypes = [C.POINTER(vp), i]; L.bcnn_init_net.restype = i
    L.bcnn_end_net.argtypes = [C.POINTER(vp)]; L.bcnn_end_net.restype = None
    L.bcnn_set_log_context.argtypes = [vp, vp, i]; L.bcnn_set_log_context.restype = None
    L.bcnn_set_input_shape.argtypes = [vp, i, i, i, i]; L.bcnn_set_input_shape.restype = None
    L.bcnn_compile_net.argtypes = [vp]; L.bcnn_compile_net.restype = i
    L.bcnn_set_mode.argtypes = [vp, i]; L.bcnn_set_mode.restype = i
    L.bcnn_forward.argtypes = [vp]; L.bcnn_forward.restype = None
    L.bcnn_backward.argtypes = [vp]; L.bcnn_backward.restype = None
    L.bcnn_update.argtypes = [vp]; L.bcnn_update.restype = None
    L.bcnn_set_sgd_optimizer.argtypes = [vp, f, f]; L.bcnn_set_sgd_optimizer.restype = None
    L.bcnn_set_weight_regularizer.argtypes = [vp, f]; L.bcnn_set_weight_regularizer.restype = None
    L.bcnn_get_tensor_index_by_name.argtypes = [vp, cp]; L.bcnn_get_tensor_index_by_name.restype = i
    L.bcnn_add_convolutional_layer.argtypes = [vp, i, i, i, i, i, i, i, i, i, cp, cp]
    L.bcnn_add_convolutional_layer.restype = i
    L.bcnn_add_depthwise_conv_layer.argtypes = [vp, i, i, i, i, i, i, cp, cp]
    L.bcnn_add_depthwise_conv_layer.restype = i
    L.bcnn_add_batchnorm_layer.argtypes = [vp, cp, cp]; L.bcnn_add_batchnorm_layer.restype = i
    L.bcnn_add_maxpool_layer.argtypes = [vp, i, i, i, cp, cp]; L.bcnn_add_maxpool_layer.restype = i
    L.bcnn_add_avgpool_layer.argtypes = [vp, cp, cp]; L.bcnn_add_avgpool_layer.restype = i
    L.bcnn_add_eltwise_layer.argtypes = [vp, i, cp, cp, cp]; L.bcnn_add_eltwise_layer.restype = i
    L.bcnn_add_activation_layer.argtypes = [vp, i, cp]; L.bcnn_add_activation_layer.restype = i
    L.bcnn_save_weights.argtypes = [vp, cp]; L.bcnn_save_weights.restype = i
    L.bcnn_load_weights.argtypes = [vp, cp]; L.bcnn_load_weights.restype = i
    L.bcnn_add_fullc_layer.argtypes = [vp, i, i, i, i, cp, cp]; L.bcnn_add_fullc_layer.restype = i
    L.bcnn_add_softmax_layer.argtypes = [vp, cp, cp]; L.bcnn_add_softmax_layer.restype = i
    L.bcnn_add_cost_layer.argtypes = [vp, i, i, f, cp, cp, cp]; L.bcnn_add_cost_layer.restype = i
    # internal but exported helpers (non-static in the reference)
    L.bcnn_forward_activation_cpu.argtypes = [fp, i, fp, i, i, i]
    L.bcnn_forward_activation_cpu.restype = None
    L.bcnn_backward_activation_cpu.argtypes = [fp, fp, i, fp, fp, i, i, i]
    L.bcnn_backward_activation_cpu.restype = None
    L.bcnn_im2col.argtypes = [fp, i, i, i, i, i, i, fp]; L.bcnn_im2col.restype = None
    L.bcnn_col2im.argtypes = [fp, i, i, i, i, i, i, fp]; L.bcnn_col2im.restype = None
    L.bcnn_add_bias.argtypes = [fp, fp, i, i, i, i]; L.bcnn_add_bias.restype = None
    L.bcnn_grad_bias.argtypes = [fp, fp, i, i, i]; L.bcnn_grad_bias.restype = None
    L.bcnn_scales.argtypes = [fp, fp, i, i, i, i]; L.bcnn_scales.restype = None           # bcnn_mat.c:772-781
    L.bcnn_grad_scales.argtypes = [fp, fp, i, i, i, fp]; L.bcnn_grad_scales.restype = None  # bcnn_mat.c:783-796
    L.bcnn_scal.argtypes = [i, f, fp]; L.bcnn_scal.restype = None                          # bcnn_mat.c:319-364
    # driver accessors (oracle/ref_driver.c)
    L.ref_num_nodes.argtypes = [vp]; L.ref_num_nodes.restype = i
    L.ref_num_tensors.argtypes = [vp]; L.ref_num_tensors.restype = i
    L.ref_node_type.argtypes = [vp, i]; L.ref_node_type.restype = i
    L.ref_node_num_src.argtypes = [vp, i]; L.ref_node_num_src.restype = i
    L.ref_node_src.argtypes = [vp, i, i]; L.ref_node_src.restype = i
    L.ref_node_dst.argtypes = [vp, i, i]; L.ref_node_dst.restype = i
    L.ref_tensor.argtypes = [vp, i]; L.ref_tensor.restype = C.POINTER(Tensor)
    L.ref_tensor_name.argtypes = [vp, i]; L.ref_tensor_name.restype = cp
    L.ref_set_mode_raw.argtypes = [vp, i]; L.ref_set_mode_raw.restype = None
    L.ref_set_threads.argtypes = [vp, i]; L.ref_set_threads.restype = None
    L.ref_get_threads.argtypes = [vp]; L.ref_get_threads.restype = i
    L.ref_forward_node.argtypes = [vp, i]; L.ref_forward_node.restype = None
    L.ref_backward_node.argtypes = [vp, i]; L.ref_backward_node.restype = None
    L.ref_maxpool_indexes.argtypes = [vp, i]; L.ref_maxpool_indexes.restype = C.POINTER(C.c_int)
    L.ref_bn_field.argtypes = [vp, i, i]; L.ref_bn_field.restype = fp
    L.ref_gemm.argtypes = [vp, i, i, i, i, i, f, fp, i, fp, i, f, fp, i]; L.ref_gemm.restype = None
    L.ref_time_fwd_bwd.argtypes = [vp, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ref_time_fwd_bwd.restype = C.c_double
    _lib = L
    return L


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


class RefNet:
    """A bcnn_net of the unmodified reference, driven through its public API."""

    def __init__(self, mode=MODE_TRAIN, w=8, h=8, c=3, n=2, input_grad=False):
        self.L = lib()
        self.net = C.c_void_p()
        st = self.L.bcnn_init_net(C.byref(self.net), mode)
        assert st == 0
        self.L.bcnn_set_log_context(self.net, None, LOG_SILENT)
        self.L.bcnn_set_input_shape(self.net, w, h, c, n)
        if input_grad:
            # the reference builds the input with has_grad=0 (src/bcnn_net.c:280-285); a test that
            # wants d(input) flips the public flag before bcnn_compile_net allocates the tensor
            self.tensor(0).has_grad = 1
        self.compiled = False

    # --- builders -----------------------------------------------------------------------------
    def conv(self, f, k, s, p, g=1, bn=0, act=ACT_NONE, src="input", dst="conv", init=FILLER_XAVIER):
        st = self.L.bcnn_add_convolutional_layer(self.net, f, k, s, p, g, bn, init, act, 0,
                                                 src.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def depthwise(self, k, s, p, act=ACT_NONE, src="input", dst="dw"):
        st = self.L.bcnn_add_depthwise_conv_layer(self.net, k, s, p, 0, FILLER_XAVIER, act,
                                                  src.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def batchnorm(self, src="input", dst="bn"):
        st = self.L.bcnn_add_batchnorm_layer(self.net, src.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def maxpool(self, k, s, padding=PADDING_SAME, src="input", dst="pool"):
        st = self.L.bcnn_add_maxpool_layer(self.net, k, s, padding, src.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def avgpool(self, src="input", dst="avg"):
        st = self.L.bcnn_add_avgpool_layer(self.net, src.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def eltwise(self, act, src1, src2, dst):
        st = self.L.bcnn_add_eltwise_layer(self.net, act, src1.encode(), src2.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def fullc(self, out, act=ACT_NONE, src="input", dst="fc"):
        st = self.L.bcnn_add_fullc_layer(self.net, out, FILLER_XAVIER, act, 0, src.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def softmax(self, src, dst):
        st = self.L.bcnn_add_softmax_layer(self.net, src.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def cost(self, src, label="label", dst="cost", scale=1.0):
        st = self.L.bcnn_add_cost_layer(self.net, LOSS_EUCLIDEAN, METRIC_ERROR_RATE, scale,
                                        src.encode(), label.encode(), dst.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def activation(self, act, src):
        st = self.L.bcnn_add_activation_layer(self.net, act, src.encode())
        assert st == 0, st
        return self.L.ref_num_nodes(self.net) - 1

    def compile(self):
        assert self.L.bcnn_compile_net(self.net) == 0
        self.compiled = True

    def save_weights(self, path):
        return self.L.bcnn_save_weights(self.net, path.encode())

    def load_weights(self, path):
        return self.L.bcnn_load_weights(self.net, path.encode())

    # --- tensors ------------------------------------------------------------------------------
    def tensor(self, idx):
        return self.L.ref_tensor(self.net, idx).contents

    def index(self, name):
        return self.L.bcnn_get_tensor_index_by_name(self.net, name.encode())

    def shape(self, idx):
        t = self.tensor(idx)
        return (t.n, t.c, t.h, t.w)

    def _view(self, ptr, shape):
        size = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(size,)).reshape(shape)

    def data(self, idx):
        t = self.tensor(idx)
        return self._view(t.data, (t.n, t.c, t.h, t.w))

    def grad(self, idx):
        t = self.tensor(idx)
        if not t.grad_data:
            return None
        return self._view(t.grad_data, (t.n, t.c, t.h, t.w))

    def node_src(self, node, i):
        return self.L.ref_node_src(self.net, node, i)

    def node_dst(self, node, i=0):
        return self.L.ref_node_dst(self.net, node, i)

    def node_num_src(self, node):
        return self.L.ref_node_num_src(self.net, node)

    def maxpool_indexes(self, node):
        shp = self.shape(self.node_dst(node))
        p = self.L.ref_maxpool_indexes(self.net, node)
        return np.ctypeslib.as_array(p, shape=(int(np.prod(shp)),)).reshape(shp)

    def bn_field(self, node, which, size):
        p = self.L.ref_bn_field(self.net, node, which)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(size,))

    # --- execution ----------------------------------------------------------------------------
    def forward(self):
        self.L.bcnn_forward(self.net)

    def backward(self):
        self.L.bcnn_backward(self.net)

    def forward_node(self, node):
        self.L.ref_forward_node(self.net, node)

    def backward_node(self, node):
        self.L.ref_backward_node(self.net, node)

    def num_nodes(self):
        return self.L.ref_num_nodes(self.net)

    def set_mode(self, mode):
        self.L.ref_set_mode_raw(self.net, mode)

    def time_fwd_bwd(self, warmup, iters):
        f, b = C.c_double(), C.c_double()
        t = self.L.ref_time_fwd_bwd(self.net, warmup, iters, C.byref(f), C.byref(b))
        return t, f.value, b.value

    def threads(self):
        return self.L.ref_get_threads(self.net)

    def close(self):
        if self.net:
            self.L.bcnn_end_net(C.byref(self.net))
            self.net = None
