"""bcnn_load_net: the INI graph loader of the HIP host library against the unmodified
reference's loader on the same config files -- same graph (node types, tensor names and shapes), same learner
settings, same parameters after loading a model file, in both dialects (bcnn *.bcnnmodel and Darknet *.weights)."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import ref_bind as rb

pytestmark = pytest.mark.gpu

BCNN_CFG = """
############ General parameters ############
[network]
output_model = out.bcnnmodel
data_format=mnist
source_train = ./train-images.idx3-ubyte
input_width=12
input_height = 10
input_channels=3
batch_size=2
optimizer=sgd
momentum=0.8
decay=0.001
learning_rate=0.02
decay_type=sigmoid
gamma=.00002
step=400

# a comment, then the layers
[convolutional]
filters=8
size=3
stride=1
pad=1
bn=1
init=xavier
function=relu
src=input
dst=conv1

[maxpool]
size=2
stride=2
src=conv1
dst=pool1

[depthwise-conv]
size=3
stride=1
pad=1
function=lrelu
src=pool1
dst=dw1

[batchnorm]
src=dw1
dst=bn1

[activation]
function=prelu
src=bn1

[conv]
filters=8
size=1
stride=1
pad=0
src=bn1
dst=pw1

[eltwise]
src=pool1,pw1
dst=sum1
function=relu

[avgpool]
src=sum1
dst=gap

[connected]
output=5
src=gap
dst=fc

[softmax]
src=fc
dst=prob

[cost]
src=prob
dst=out
loss=euclidean
metric=error
"""

DARKNET_CFG = """
[net]
batch=2
width=8
height=8
channels=3

[convolutional]
batch_normalize=1
filters=4
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
filters=4
size=1
stride=1
pad=1
activation=linear

[shortcut]
from=-2
activation=linear
"""


def load_both(cfg_path, model_path, mode):
    from bcnn_amd import capi
    L = rb.lib()
    L.bcnn_load_net.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.bcnn_load_net.restype = C.c_int
    ref = rb.RefNet.__new__(rb.RefNet)
    ref.L, ref.net, ref.compiled = L, C.c_void_p(), False
    assert L.bcnn_init_net(C.byref(ref.net), mode) == 0
    L.bcnn_set_log_context(ref.net, None, rb.LOG_SILENT)
    st_ref = L.bcnn_load_net(ref.net, cfg_path.encode(), model_path.encode() if model_path else None)
    net = capi.Net.__new__(capi.Net)
    net.L, net.net = capi.lib(), C.c_void_p()
    assert net.L.bcnn_init_net(C.byref(net.net), mode) == 0
    net.L.bcnn_set_log_context(net.net, None, 4)
    st = net.L.bcnn_load_net(net.net, cfg_path.encode(), model_path.encode() if model_path else None)
    return ref, st_ref, net, st


def same_graph(ref, net):
    nt = ref.L.ref_num_tensors(ref.net)
    nn = ref.L.ref_num_nodes(ref.net)
    assert net.L.bcnn_get_num_nodes(net.net) == nn
    for i in range(nn):
        assert net.L.bcnn_get_node_tensor(net.net, i, 1, 0) == ref.node_dst(i)
        for k in range(ref.node_num_src(i)):
            assert net.L.bcnn_get_node_tensor(net.net, i, 0, k) == ref.node_src(i, k), (i, k)
    for i in range(nt):
        t = net.L.bcnn_peek_tensor(net.net, i)
        assert t, "tensor %d missing" % i
        assert (t.contents.name or b"").decode() == ref.L.ref_tensor_name(ref.net, i).decode()
        assert (t.contents.n, t.contents.c, t.contents.h, t.contents.w) == ref.shape(i), i
    assert not net.L.bcnn_peek_tensor(net.net, nt)
    return nt


def test_bcnn_config_builds_the_reference_graph_and_loads_its_model(tmp_path):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    cfg = tmp_path / "net.conf"
    cfg.write_text(BCNN_CFG)
    ref, st_ref, net, st = load_both(str(cfg), None, rb.MODE_TRAIN)
    assert st_ref == 0 and st == 0
    nt = same_graph(ref, net)
    assert nt > 20
    assert ref.L.bcnn_compile_net(ref.net) == 0 and net.L.bcnn_compile_net(net.net) == 0
    # give the reference distinctive parameters, save them, rebuild both nets with that model file
    rs = np.random.RandomState(3)
    for i in range(2, nt):
        nm = ref.L.ref_tensor_name(ref.net, i).decode()
        if any(nm.endswith(s) for s in ("_w", "_b", "_run_mean", "_run_var", "_scales", "_w_prelu")):
            a = ref.data(i)
            lo, hi = (0.5, 1.5) if ("run_var" in nm or "scales" in nm) else (-0.5, 0.5)
            a[...] = rs.uniform(lo, hi, a.shape).astype(np.float32)
    model = str(tmp_path / "m.bcnnmodel")
    assert ref.save_weights(model) == 0
    ref2, st_ref2, net2, st2 = load_both(str(cfg), model, rb.MODE_TRAIN)
    assert st_ref2 == 0 and st2 == 0
    same_graph(ref2, net2)
    from bcnn_amd import capi
    n2 = capi.Net.__new__(capi.Net)
    n2.L, n2.net = net2.L, net2.net
    for i in range(2, nt):
        t = ref2.tensor(i)
        if t.data and ref2.L.ref_tensor_name(ref2.net, i).decode().split("_")[-1] in ("w", "b", "mean", "var", "scales", "prelu"):
            n2.download(i, False)
            np.testing.assert_array_equal(n2.data(i), ref2.data(i), err_msg=ref2.L.ref_tensor_name(ref2.net, i).decode())


def test_darknet_dialect(tmp_path):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    cfg = tmp_path / "tiny.cfg"
    cfg.write_text(DARKNET_CFG)
    rs = np.random.RandomState(1)
    model = tmp_path / "tiny.weights"
    with open(model, "wb") as fp:
        fp.write(struct.pack("<iii", 0, 2, 0) + struct.pack("<Q", 7))
        for cnt in (4, 4, 4, 4, 4 * 3 * 9, 4, 4 * 4):
            fp.write(rs.uniform(-1, 1, cnt).astype(np.float32).tobytes())
    ref, st_ref, net, st = load_both(str(cfg), str(model), rb.MODE_PREDICT)
    assert st_ref == 0 and st == 0
    nt = same_graph(ref, net)
    from bcnn_amd import capi
    n2 = capi.Net.__new__(capi.Net)
    n2.L, n2.net = net.L, net.net
    for i in range(2, nt):
        nm = ref.L.ref_tensor_name(ref.net, i).decode()
        if nm.split("_")[-1] in ("w", "b", "mean", "var", "scales"):
            n2.download(i, False)
            np.testing.assert_array_equal(n2.data(i), ref.data(i), err_msg=nm)


def test_loader_error_paths(tmp_path):
    from bcnn_amd import capi

    def status(text, model=None):
        p = tmp_path / "c.conf"
        p.write_text(text)
        net = capi.Net.__new__(capi.Net)
        net.L, net.net = capi.lib(), C.c_void_p()
        assert net.L.bcnn_init_net(C.byref(net.net), capi.MODE_TRAIN) == 0
        net.L.bcnn_set_log_context(net.net, None, 4)
        return net.L.bcnn_load_net(net.net, str(p).encode(), model.encode() if model else None)
    assert status("") == 1                                            # empty file
    assert status("[convolutional]\nfilters=2\n") == 1                # first section must be [net]
    assert status("[net]\n[convolutional]\nsrc=input\ndst=a\n") == 1  # empty [net]
    assert status("[net]\nbatch=2\nwidth=4\nheight=4\nchannels=1\n[whatever]\nsrc=input\ndst=a\n") == 1
    assert status("[net]\nbatch=2\nwidth=4\nheight=4\nchannels=1\n[conv]\nfilters=2\nsrc=input\n") == 1  # no dst
    assert status("[net]\nbatch=2\nwidth=4\nheight=4\nchannels=1\nbad line\n") == 1
    assert status("[net]\nbatch=2\nwidth=4\nheight=4\nchannels=1\n[conv]\nfilters=2\nsrc=input\ndst=a\n") == 0
    assert status("[net]\nbatch=2\n", model="noextension") == 2         # BCNN_INVALID_DATA like the reference
