"""Model files: bcnn_save_weights / bcnn_load_weights of the HIP build against the
unmodified reference's own reader and writer (reference src/bcnn_net.c:597-681, 1219-1558).
  * a file written by the reference loads here and reproduces the reference's parameters and forward output;
  * a file written here is BYTE-IDENTICAL to the one the reference writes for the same parameters;
  * Darknet-layout files (*.weights) are read the same way by both;
  * error paths return the reference's status codes."""
import filecmp
import os
import struct

import numpy as np
import pytest

from oracle import ref_bind as rb

pytestmark = pytest.mark.gpu
INVALID_PARAMETER, INVALID_MODEL = 1, 3


def graph(net):
    net.conv(8, 3, 1, 1, 1, 1, rb.ACT_RELU, "input", "c1")        # conv + fused batch-norm
    net.conv(6, 3, 2, 1, 2, 0, rb.ACT_LRELU, "c1", "c2")          # grouped, no batch-norm
    net.depthwise(3, 1, 1, rb.ACT_RELU, "c2", "dw")
    net.batchnorm("dw", "bn")                                      # stand-alone batch-norm node
    net.activation(rb.ACT_PRELU, "bn")                             # PReLU node: slopes are stored
    net.fullc(5, rb.ACT_NONE, "bn", "fc")


def both(mode):
    from bcnn_amd import capi
    ref = rb.RefNet(mode=mode, n=2, w=10, h=10, c=3)
    ref.L.ref_set_threads(ref.net, 2)
    net = capi.Net(mode=mode, n=2, w=10, h=10, c=3)
    graph(ref)
    graph(net)
    ref.compile()
    net.compile()
    return ref, net


ACTIVATIONS = {"input", "label", "c1", "c2", "dw", "bn", "fc"}


def all_small_tensors(ref):
    """every tensor that is not an activation: weights, biases, batch-norm state, PReLU slopes"""
    nt = ref.L.ref_num_tensors(ref.net)
    return [i for i in range(nt) if ref.L.ref_tensor_name(ref.net, i).decode() not in ACTIVATIONS]


def randomise(ref, seed):
    rs = np.random.RandomState(seed)
    for i in all_small_tensors(ref):
        a = ref.data(i)
        nm = ref.L.ref_tensor_name(ref.net, i).decode()
        lo, hi = (0.5, 1.5) if ("run_var" in nm or "scales" in nm) else (-0.5, 0.5)
        a[...] = rs.uniform(lo, hi, a.shape).astype(np.float32)


def test_reference_file_loads_here_and_round_trips_byte_identical(tmp_path):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    ref, net = both(rb.MODE_TRAIN)
    ids = all_small_tensors(ref)
    assert len(ids) >= 12, [ref.L.ref_tensor_name(ref.net, i) for i in ids]
    randomise(ref, 5)
    f_ref = str(tmp_path / "ref.bcnnmodel")
    assert ref.save_weights(f_ref) == 0
    assert net.load_weights(f_ref) == 0
    for i in ids:
        net.download(i, False)
        np.testing.assert_array_equal(net.data(i), ref.data(i), err_msg=ref.L.ref_tensor_name(ref.net, i).decode())
    # our writer produces the same bytes as the reference's (before the TRAIN forward moves the running statistics)
    f_hip = str(tmp_path / "hip.bcnnmodel")
    assert net.save_weights(f_hip) == 0
    assert filecmp.cmp(f_ref, f_hip, shallow=False), (os.path.getsize(f_ref), os.path.getsize(f_hip))
    # forward agrees: same parameters, same input
    x = np.random.RandomState(9).uniform(-1, 1, (2, 3, 10, 10)).astype(np.float32)
    ref.data(0)[...] = x
    net.data(0)[...] = x
    net.upload(0)
    ref.forward()
    net.forward()
    out = ref.index("fc")
    net.download(out, False)
    np.testing.assert_allclose(net.data(out), ref.data(out), rtol=1e-4, atol=1e-5)
    # and the reference reads our file (written after the forward: updated running statistics) back to our state
    f_hip2 = str(tmp_path / "hip_after_forward.bcnnmodel")
    assert net.save_weights(f_hip2) == 0
    randomise(ref, 77)
    assert ref.load_weights(f_hip2) == 0
    for i in ids:
        net.download(i, False)
        np.testing.assert_array_equal(ref.data(i), net.data(i))


def test_predict_mode_folds_batchnorm_like_the_reference(tmp_path):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    src, _ = both(rb.MODE_TRAIN)
    randomise(src, 21)
    f = str(tmp_path / "m.bcnnmodel")
    assert src.save_weights(f) == 0
    ref, net = both(rb.MODE_PREDICT)
    assert ref.load_weights(f) == 0 and net.load_weights(f) == 0
    for i in all_small_tensors(ref):
        net.download(i, False)
        np.testing.assert_allclose(net.data(i), ref.data(i), rtol=0, atol=0,
                                   err_msg=ref.L.ref_tensor_name(ref.net, i).decode())
    # inference on the loaded model: the reference runs its Winograd 3x3 path and the folded batch-norm here
    x = np.random.RandomState(2).uniform(-1, 1, (2, 3, 10, 10)).astype(np.float32)
    ref.data(0)[...] = x
    net.data(0)[...] = x
    net.upload(0)
    ref.forward()
    net.forward()
    for name in ("c1", "c2", "dw", "bn", "fc"):
        i = ref.index(name)
        net.download(i, False)
        scale = max(float(np.abs(ref.data(i)).max()), 1e-6)
        assert float(np.abs(net.data(i) - ref.data(i)).max()) <= 1e-4 * scale, name


def test_darknet_layout(tmp_path):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    from bcnn_amd import capi

    def g(net):
        net.conv(4, 3, 1, 1, 1, 1, rb.ACT_RELU, "input", "c1")
        net.fullc(3, rb.ACT_NONE, "c1", "fc")
    ref = rb.RefNet(mode=rb.MODE_TRAIN, n=1, w=6, h=6, c=2)
    net = capi.Net(mode=capi.MODE_TRAIN, n=1, w=6, h=6, c=2)
    g(ref); g(net)
    ref.compile(); net.compile()
    rs = np.random.RandomState(4)
    f = str(tmp_path / "tiny.weights")
    with open(f, "wb") as fp:
        fp.write(struct.pack("<iii", 0, 2, 0) + struct.pack("<Q", 1234))        # major, minor, revision, seen (u64)
        # conv: biases, scales, means, variances, weights   | fc: biases, weights
        for count in (4, 4, 4, 4, 4 * 2 * 3 * 3, 3, 3 * 4 * 6 * 6):
            fp.write(rs.uniform(-1, 1, count).astype(np.float32).tobytes())
    assert ref.load_weights(f) == 0 and net.load_weights(f) == 0
    for i in all_small_tensors(ref):
        net.download(i, False)
        np.testing.assert_array_equal(net.data(i), ref.data(i), err_msg=ref.L.ref_tensor_name(ref.net, i).decode())


def test_error_paths(tmp_path):
    from bcnn_amd import capi
    net = capi.Net(mode=capi.MODE_TRAIN, n=1, w=6, h=6, c=2)
    net.conv(4, 3, 1, 1, 1, 0, rb.ACT_RELU, "input", "c1")
    net.compile()
    assert net.load_weights(str(tmp_path / "does_not_exist.bcnnmodel")) == INVALID_PARAMETER
    assert net.save_weights(str(tmp_path / "no_such_dir" / "x.bcnnmodel")) == INVALID_PARAMETER
    bad = str(tmp_path / "bad_magic.bcnnmodel")
    open(bad, "wb").write(b"NOPE" + b"\0" * 64)
    assert net.load_weights(bad) == INVALID_MODEL
    good = str(tmp_path / "good.bcnnmodel")
    assert net.save_weights(good) == 0
    cut = str(tmp_path / "truncated.bcnnmodel")
    open(cut, "wb").write(open(good, "rb").read()[:-8])
    assert net.load_weights(cut) == INVALID_MODEL
    assert net.load_weights(str(tmp_path / "model.onnx")) == INVALID_PARAMETER  # missing file wins over format
    onnx = str(tmp_path / "m.onnx")
    open(onnx, "wb").write(b"\0" * 16)
    assert net.load_weights(onnx) == INVALID_MODEL
