"""The reference's command-line tool, unchanged, on this build: `bcnn-cl train <config>` (src/cli/bcnn_cl.c) reads an INI
file, builds the net with bcnn_load_net, opens the data set with bcnn_set_data_loader, trains with bcnn_train_on_batch,
evaluates in VALID mode every eval_period iterations, writes predictions and check-points. oracle/Makefile (`make cli`,
run by __graft_entry__.build() where the reference tree is mounted) links that one source file twice: against the
reference library (oracle/_ref/bcnn-cl-ref, CPU) and against this build's headers + libbcnn.so / libbip.so / libbcnn_hip.so
(oracle/_ref/bcnn-cl-hip). Both run here on the same synthetic MNIST-format files with the example's augmentation
(examples/mnist_cl/mnist.conf: shift, scale, rotation); neither seeds libc, so both start from rand()'s default state and
see the same initial weights and the same augmented batches.

Compared: the prediction file of the final evaluation (softmax outputs, 6 decimals), the saved model and the check-point
(weight files of identical layout, values within 1e-4), the logged train / test error rates."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(os.path.join(REFDIR, "bcnn-cl-ref")) and
                                      os.path.exists(os.path.join(REFDIR, "bcnn-cl-hip"))),
                                 reason="oracle/_ref/bcnn-cl-* not built (make -C oracle cli needs the reference tree)")]

CONF = """[network]
output_model=out.bcnnmodel
out_pred=pred.txt
eval_test=1
eval_period=4
save_model=6
num_pred=32
max_batches=10
data_format=%(fmt)s
source_train=%(train)s
%(label_train)s
source_test=%(test)s
%(label_test)s
range_shift_x=5
range_shift_y=5
min_scale=0.85
max_scale=1.15
rotation_range=30
input_width=%(side)d
input_height=%(side)d
input_channels=%(chan)d
batch_size=16
optimizer=sgd
momentum=0.9
decay=0.0005
learning_rate=0.003
decay_type=sigmoid
gamma=.00002
step=400000

[convolutional]
filters=16
size=3
stride=1
pad=1
init=xavier
batchnorm=1
function=relu
src=input
dst=conv1

[maxpool]
size=2
stride=2
src=conv1
dst=pool1

[convolutional]
filters=32
size=3
stride=1
pad=1
init=xavier
function=relu
src=pool1
dst=conv2

[avgpool]
src=conv2
dst=gap

[connected]
output=10
init=xavier
src=gap
dst=fc2

[softmax]
src=fc2
dst=soft

[cost]
src=soft
dst=out
loss=euclidean
metric=error
"""


def _mnist(d, name, n, seed):
    rs = np.random.RandomState(seed)
    with open(os.path.join(d, name + "-images"), "wb") as f:
        f.write(struct.pack(">IIII", 2051, n, 28, 28) + rs.randint(0, 256, (n, 28, 28)).astype(np.uint8).tobytes())
    with open(os.path.join(d, name + "-labels"), "wb") as f:
        f.write(struct.pack(">II", 2049, n) + rs.randint(0, 10, n).astype(np.uint8).tobytes())


def _cifar(d, name, n, seed):
    rs = np.random.RandomState(seed)
    with open(os.path.join(d, name), "wb") as f:
        for _ in range(n):
            f.write(bytes([int(rs.randint(0, 10))]) + rs.randint(0, 256, 3072).astype(np.uint8).tobytes())


def _run(exe, cwd):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "bcnn_amd", "lib") + ":" + REFDIR + ":" +
               os.environ.get("LD_LIBRARY_PATH", ""), OMP_NUM_THREADS="8")
    r = subprocess.run([os.path.join(REFDIR, exe), "train", "m.conf"], cwd=cwd, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    log = r.stdout + r.stderr
    assert "Training ended successfully" in log, log[-2000:]
    errs = [(int(m.group(1)), float(m.group(2)), float(m.group(3)))
            for m in re.finditer(r"iter-batches= (\d+) train-error= ([\d.]+) test-error= ([\d.]+)", log)]
    pred = np.loadtxt(os.path.join(cwd, "pred.txt"), dtype=np.float64)
    return errs, pred, open(os.path.join(cwd, "out.bcnnmodel"), "rb").read(), \
        open(os.path.join(cwd, "out.bcnnmodel_iter6.bcnnmodel"), "rb").read()


def _weights_close(a, b):
    """weight files have the same byte layout (tests/test_weights_io.py); compare them as float32 streams"""
    assert len(a) == len(b)
    n = len(a) // 4 * 4
    fa, fb = np.frombuffer(a[:n], np.float32), np.frombuffer(b[:n], np.float32)
    ok = np.isfinite(fa) & np.isfinite(fb) & (np.abs(fa) < 1e6) & (np.abs(fb) < 1e6)   # skip header words read as floats
    assert (ok == (np.isfinite(fa) & (np.abs(fa) < 1e6))).all()
    err = np.abs(fa[ok] - fb[ok]).max() / max(np.abs(fa[ok]).max(), 1e-30)
    assert err <= 1e-4, err


@pytest.mark.parametrize("fmt", ["mnist", "cifar10"])
def test_unchanged_bcnn_cl_trains_like_on_the_reference(tmp_path, fmt):
    runs = []
    for exe in ("bcnn-cl-ref", "bcnn-cl-hip"):
        d = tmp_path / exe
        d.mkdir()
        if fmt == "mnist":
            _mnist(str(d), "train", 64, 1)
            _mnist(str(d), "test", 32, 2)
            kv = dict(fmt="mnist", train="train-images", label_train="label_train=train-labels", test="test-images",
                      label_test="label_test=test-labels", side=28, chan=1)
        else:
            _cifar(str(d), "train.bin", 48, 3)
            _cifar(str(d), "test.bin", 32, 4)
            kv = dict(fmt="cifar10", train="train.bin", label_train="", test="test.bin", label_test="", side=32, chan=3)
        (d / "m.conf").write_text(CONF % kv)
        runs.append(_run(exe, str(d)))
    (e_ref, p_ref, m_ref, c_ref), (e_hip, p_hip, m_hip, c_hip) = runs
    assert [e[0] for e in e_ref] == [e[0] for e in e_hip] == [4, 8]
    for a, b in zip(e_ref, e_hip):
        assert abs(a[1] - b[1]) <= 1.0 / 64 + 1e-6 and abs(a[2] - b[2]) <= 1.0 / 32 + 1e-6, (e_ref, e_hip)   # one arg-max flip
    assert p_ref.shape == p_hip.shape and p_ref.size == 32 * 10
    assert np.abs(p_ref - p_hip).max() <= 2e-5, np.abs(p_ref - p_hip).max()
    _weights_close(m_ref, m_hip)
    _weights_close(c_ref, c_hip)
