"""ctypes binding of lib/libbcnn.so -- the C99 host runtime (bcnn_net / bcnn_node API of include/bcnn/bcnn.h)
on top of the HIP back-end. Mirrors the reference's public API one to one; `Net` is a thin convenience
wrapper used by tests and bench.py (same method names as oracle/ref_bind.RefNet, which drives the
unmodified reference, so a graph can be built on both with the same code)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BCNN_LIB", os.path.join(_HERE, "lib", "libbcnn.so"))  # override: experiment build

MODE_PREDICT, MODE_TRAIN, MODE_VALID = 0, 1, 2
(ACT_NONE, ACT_TANH, ACT_RELU, ACT_RAMP, ACT_SOFTPLUS, ACT_LRELU, ACT_ABS, ACT_CLAMP, ACT_PRELU,
 ACT_LOGISTIC) = range(10)
PADDING_SAME, PADDING_VALID, PADDING_CAFFE = 0, 1, 2
FILLER_FIXED, FILLER_XAVIER, FILLER_MSRA = 0, 1, 2
LOG_SILENT = 3


class Tensor(C.Structure):
    """struct bcnn_tensor with BCNN_USE_HIP (include/bcnn/bcnn.h)."""
    _fields_ = [("n", C.c_int), ("c", C.c_int), ("h", C.c_int), ("w", C.c_int), ("has_grad", C.c_int),
                ("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("grad_data", C.POINTER(C.c_float)),
                ("data_gpu", C.c_void_p), ("grad_data_gpu", C.c_void_p)]


_lib = None


def build():
    from . import _lib as hip
    hip.build()
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "host")], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    from . import _lib as hip
    hip.load()  # loads libbcnn_hip.so first (and torch before it, see _lib.load)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("bcnn_amd: %s missing -- run __graft_entry__.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, f, cp, sz = C.c_void_p, C.c_int, C.c_float, C.c_char_p, C.c_size_t
    tp = C.POINTER(Tensor)
    sig = {
        "bcnn_init_net": (i, [C.POINTER(vp), i]), "bcnn_end_net": (None, [C.POINTER(vp)]),
        "bcnn_set_log_context": (None, [vp, vp, i]), "bcnn_set_input_shape": (None, [vp, i, i, i, i]),
        "bcnn_compile_net": (i, [vp]), "bcnn_set_mode": (i, [vp, i]), "bcnn_resize_net": (i, [vp, i, i, i, i]),
        "bcnn_forward": (None, [vp]), "bcnn_backward": (None, [vp]), "bcnn_update": (None, [vp]),
        "bcnn_train_on_batch": (f, [vp]),
        "bcnn_set_sgd_optimizer": (None, [vp, f, f]), "bcnn_set_weight_regularizer": (None, [vp, f]),
        "bcnn_get_tensor_index_by_name": (i, [vp, cp]), "bcnn_get_tensor_by_index": (tp, [vp, i]),
        "bcnn_get_batch_size": (i, [vp]),
        "bcnn_add_convolutional_layer": (i, [vp, i, i, i, i, i, i, i, i, i, cp, cp]),
        "bcnn_add_depthwise_conv_layer": (i, [vp, i, i, i, i, i, i, cp, cp]),
        "bcnn_add_batchnorm_layer": (i, [vp, cp, cp]), "bcnn_add_maxpool_layer": (i, [vp, i, i, i, cp, cp]),
        "bcnn_add_avgpool_layer": (i, [vp, cp, cp]), "bcnn_add_activation_layer": (i, [vp, i, cp]),
        "bcnn_add_eltwise_layer": (i, [vp, i, cp, cp, cp]), "bcnn_add_fullc_layer": (i, [vp, i, i, i, i, cp, cp]),
        "bcnn_add_softmax_layer": (i, [vp, cp, cp]), "bcnn_add_cost_layer": (i, [vp, i, i, f, cp, cp, cp]),
        "bcnn_upload_tensor": (i, [vp, i, i]), "bcnn_download_tensor": (i, [vp, i, i]),
        "bcnn_set_data_parallel": (i, [vp, i, i]), "bcnn_set_data_parallel_comm": (i, [vp, i, i, cp]),
        "bcnn_set_weight_gradient_stream": (None, [vp, i]),
        "bcnn_set_gradient_ready_callback": (None, [vp, vp, vp]),
        "bcnn_get_gradient_arena": (vp, [vp, C.POINTER(sz)]), "bcnn_get_parameter_arena": (vp, [vp, C.POINTER(sz)]),
        "bcnn_synchronize": (None, [vp]), "bcnn_peek_tensor": (tp, [vp, i]), "bcnn_get_num_nodes": (i, [vp]),
        "bcnn_get_node_tensor": (i, [vp, i, i, i]), "bcnn_get_node_state": (vp, [vp, i, i]),
        "bcnn_forward_node": (i, [vp, i]), "bcnn_backward_node": (i, [vp, i]),
        "bcnn_load_net": (i, [vp, cp, cp]), "bcnn_save_weights": (i, [vp, cp]), "bcnn_load_weights": (i, [vp, cp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


class DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch.as_tensor can alias it."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class Net:
    def __init__(self, mode=MODE_TRAIN, w=8, h=8, c=3, n=2, input_grad=False, silent=True):
        self.L = lib()
        self.net = C.c_void_p()
        assert self.L.bcnn_init_net(C.byref(self.net), mode) == 0
        if silent:
            self.L.bcnn_set_log_context(self.net, None, LOG_SILENT)
        self.L.bcnn_set_input_shape(self.net, w, h, c, n)
        self.num_nodes = 0
        self._node_io = []
        if input_grad:
            self.tensor(0).has_grad = 1

    # builders return the node index, like ref_bind.RefNet
    def _added(self, st):
        assert st == 0, "builder failed with status %d" % st
        self.num_nodes += 1
        return self.num_nodes - 1

    def conv(self, f, k, s, p, g=1, bn=0, act=ACT_NONE, src="input", dst="conv", init=FILLER_XAVIER):
        return self._added(self.L.bcnn_add_convolutional_layer(self.net, f, k, s, p, g, bn, init, act, 0,
                                                                src.encode(), dst.encode()))

    def depthwise(self, k, s, p, act=ACT_NONE, src="input", dst="dw"):
        return self._added(self.L.bcnn_add_depthwise_conv_layer(self.net, k, s, p, 0, FILLER_XAVIER, act,
                                                                 src.encode(), dst.encode()))

    def batchnorm(self, src, dst):
        return self._added(self.L.bcnn_add_batchnorm_layer(self.net, src.encode(), dst.encode()))

    def maxpool(self, k, s, padding=PADDING_SAME, src="input", dst="pool"):
        return self._added(self.L.bcnn_add_maxpool_layer(self.net, k, s, padding, src.encode(), dst.encode()))

    def avgpool(self, src, dst):
        return self._added(self.L.bcnn_add_avgpool_layer(self.net, src.encode(), dst.encode()))

    def activation(self, act, src):
        return self._added(self.L.bcnn_add_activation_layer(self.net, act, src.encode()))

    def eltwise(self, act, src1, src2, dst):
        return self._added(self.L.bcnn_add_eltwise_layer(self.net, act, src1.encode(), src2.encode(), dst.encode()))

    def fullc(self, out, act=ACT_NONE, src="input", dst="fc"):
        return self._added(self.L.bcnn_add_fullc_layer(self.net, out, FILLER_XAVIER, act, 0, src.encode(), dst.encode()))

    def softmax(self, src, dst):
        return self._added(self.L.bcnn_add_softmax_layer(self.net, src.encode(), dst.encode()))

    def cost(self, src, label="label", dst="cost", scale=1.0):
        return self._added(self.L.bcnn_add_cost_layer(self.net, 0, 0, scale, src.encode(), label.encode(), dst.encode()))

    def compile(self):
        assert self.L.bcnn_compile_net(self.net) == 0

    def resize(self, w, h, c, need_realloc=True):
        """bcnn_resize_net (reference bcnn_net.c:287-335): batch 1, destination tensors re-shaped (and re-allocated)"""
        return self.L.bcnn_resize_net(self.net, w, h, c, 1 if need_realloc else 0)

    # tensors: host views; call download()/upload() around them
    def index(self, name):
        return self.L.bcnn_get_tensor_index_by_name(self.net, name.encode())

    def tensor(self, idx):
        # raw struct WITHOUT the implicit device->host refresh of bcnn_get_tensor_by_index
        return self.L.bcnn_peek_tensor(self.net, idx).contents

    def node_src(self, node, i):
        return self.L.bcnn_get_node_tensor(self.net, node, 0, i)

    def node_dst(self, node, i=0):
        return self.L.bcnn_get_node_tensor(self.net, node, 1, i)

    def node_state(self, node, which):
        return self.L.bcnn_get_node_state(self.net, node, which)

    def shape(self, idx):
        t = self.tensor(idx)
        return (t.n, t.c, t.h, t.w)

    def _view(self, ptr, shape):
        return np.ctypeslib.as_array(ptr, shape=(int(np.prod(shape)),)).reshape(shape)

    def data(self, idx):
        t = self.tensor(idx)
        return self._view(t.data, (t.n, t.c, t.h, t.w))

    def grad(self, idx):
        t = self.tensor(idx)
        return self._view(t.grad_data, (t.n, t.c, t.h, t.w)) if t.grad_data else None

    def upload(self, idx, with_grad=False):
        assert self.L.bcnn_upload_tensor(self.net, idx, 1 if with_grad else 0) == 0

    def download(self, idx, with_grad=True):
        assert self.L.bcnn_download_tensor(self.net, idx, 1 if with_grad else 0) == 0

    def forward(self):
        self.L.bcnn_forward(self.net)

    def backward(self):
        self.L.bcnn_backward(self.net)

    def update(self):
        self.L.bcnn_update(self.net)

    def forward_node(self, node):
        assert self.L.bcnn_forward_node(self.net, node) == 0

    def backward_node(self, node):
        assert self.L.bcnn_backward_node(self.net, node) == 0

    def sync(self):
        self.L.bcnn_synchronize(self.net)

    def save_weights(self, path):
        """bcnn_save_weights (reference bcnn_net.c:597-681); returns the bcnn_status"""
        return self.L.bcnn_save_weights(self.net, path.encode())

    def load_weights(self, path):
        """bcnn_load_weights (reference bcnn_net.c:1485-1558); returns the bcnn_status"""
        return self.L.bcnn_load_weights(self.net, path.encode())

    def set_sgd(self, lr, momentum, decay=0.0):
        self.L.bcnn_set_sgd_optimizer(self.net, lr, momentum)
        self.L.bcnn_set_weight_regularizer(self.net, decay)

    def set_data_parallel(self, rank, world):
        assert self.L.bcnn_set_data_parallel(self.net, rank, world) == 0

    def set_data_parallel_comm(self, rank, world, id_path=None):
        """RCCL inside the library: bcnn_backward all-reduces the gradient arena itself (include/bcnn/bcnn.h)"""
        assert self.L.bcnn_set_data_parallel_comm(self.net, rank, world, id_path.encode() if id_path else None) == 0

    def set_gradient_ready_callback(self, fn):
        """fn(first_float, num_floats) is called inside backward() as tail ranges of the gradient arena
        become final (see include/bcnn/bcnn.h); None removes it."""
        if fn is None:
            self._grad_cb = None
            self.L.bcnn_set_gradient_ready_callback(self.net, None, None)
            return
        proto = C.CFUNCTYPE(None, C.c_size_t, C.c_size_t, C.c_void_p)
        self._grad_cb = proto(lambda first, count, user: fn(int(first), int(count)))  # keep alive
        self.L.bcnn_set_gradient_ready_callback(self.net, C.cast(self._grad_cb, C.c_void_p), None)

    def gradient_arena(self):
        n = C.c_size_t()
        p = self.L.bcnn_get_gradient_arena(self.net, C.byref(n))
        return p, n.value

    def parameter_arena(self):
        n = C.c_size_t()
        p = self.L.bcnn_get_parameter_arena(self.net, C.byref(n))
        return p, n.value

    def close(self):
        if self.net:
            self.L.bcnn_end_net(C.byref(self.net))
            self.net = None
