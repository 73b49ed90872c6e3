"""Run-to-run determinism of the device path (every reduction is two-level with a fixed order, no
atomics): two identical nets on identical inputs must agree BIT FOR BIT on every tensor and gradient after two
training steps. The graph reaches the LDS-DMA GEMMs (forward, dX by stride classes, dW split-q + finalize), the
fused batch-norm statistics, depthwise, pooling, eltwise, fc, softmax and the chunked SGD."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_two_identical_nets_agree_bitwise():
    import torch  # noqa: F401  (single HIP runtime)
    from bcnn_amd import capi

    def graph(net):
        net.conv(64, 3, 1, 1, 1, 1, capi.ACT_RELU, "input", "c1")
        net.maxpool(3, 2, capi.PADDING_SAME, "c1", "p1")
        net.conv(64, 3, 1, 1, 1, 1, capi.ACT_RELU, "p1", "c2")
        net.conv(128, 3, 2, 1, 1, 1, capi.ACT_NONE, "c2", "c3")
        net.conv(128, 1, 2, 0, 1, 1, capi.ACT_NONE, "c2", "proj")
        net.eltwise(capi.ACT_RELU, "proj", "c3", "e1")
        net.depthwise(3, 1, 1, capi.ACT_RELU, "e1", "dw")
        net.batchnorm("dw", "bn")
        net.avgpool("bn", "gap")
        net.fullc(10, capi.ACT_NONE, "gap", "fc")
        net.softmax("fc", "prob")
        net.cost("prob", "label", "cost", 1.0)

    def make():
        net = capi.Net(mode=capi.MODE_TRAIN, w=32, h=32, c=16, n=8)
        graph(net); net.compile(); net.set_sgd(0.01, 0.9, 5e-4)
        return net

    a, b = make(), make()
    nt = 0
    while a.L.bcnn_peek_tensor(a.net, nt): nt += 1
    rs = np.random.RandomState(1)
    for i in range(nt):
        if a.tensor(i).data:
            a.download(i, False)
            if i == 0: a.data(0)[...] = rs.uniform(-1, 1, a.shape(0)).astype(np.float32)
            if i == 1:
                lab = np.zeros(a.shape(1), np.float32); lab[np.arange(8), rs.randint(0, 10, 8)] = 1.0; a.data(1)[...] = lab
            b.data(i)[...] = a.data(i)
            a.upload(i); b.upload(i)
    bad = 0
    for step in range(2):
        for net in (a, b):
            net.forward(); net.backward()
        for i in range(nt):
            if not a.tensor(i).data: continue
            a.download(i); b.download(i)
            if not np.array_equal(a.data(i).view(np.uint32), b.data(i).view(np.uint32)): bad += 1; print("data differs", step, i)
            ga, gb = a.grad(i), b.grad(i)
            if ga is not None and not np.array_equal(ga.view(np.uint32), gb.view(np.uint32)): bad += 1; print("grad differs", step, i)
        a.update(); b.update()
    assert bad == 0, "%d tensors differ between two identical runs" % bad


@pytest.mark.gpu
def test_first_device_touch_keeps_the_callers_rand_sequence():
    """The builders draw their initial weights from libc rand() like the reference's (bcnn_tensor.c:53-58). The HIP
    runtime's lazy initialisation (first allocation / first launch) re-seeds that generator; the library parks the
    caller's state aside around it (runtime.hip), so srand(seed) before building a net gives the same parameters in
    every run and on every rank. Needs a process that has not touched the device yet."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
libc = ctypes.CDLL(None)
libc.srand(7); base = [libc.rand() for _ in range(8)]
from bcnn_amd import capi
libc.srand(7)
net = capi.Net(mode=capi.MODE_TRAIN, w=16, h=16, c=3, n=2)      # first device allocation happens in here
got = [libc.rand() for _ in range(8)]
assert got == base, (got, base)
libc.srand(7)
net.conv(8, 3, 1, 1, 1, 0, capi.ACT_RELU, "input", "c1")
w = net.data(net.index("input_w")).ravel()
libc.srand(7)
a = (3.0 / 27) ** 0.5
import numpy as np
want = np.array([a * (2 * (libc.rand() / 2147483647.0) - 1) for _ in range(w.size)], np.float32)
assert np.allclose(w, want, rtol=1e-6, atol=1e-7), (w[:4], want[:4])
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
