"""bcnn_resize_net (reference src/bcnn_net.c:287-335): a fully convolutional PREDICT net built for one input extent and
resized to a smaller one has to give what a net built for that extent gives, with the same parameters -- shapes by the
reference's rules (batch 1; convolution and max-pooling outputs from their hyper-parameters, everything else a copy of its
source's shape), tensors re-allocated on host and device."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(capi, w, h, n):
    ctypes.CDLL(None).srand(11)
    net = capi.Net(mode=capi.MODE_PREDICT, w=w, h=h, c=3, n=n)
    net.conv(16, 3, 1, 1, 1, 1, capi.ACT_RELU, "input", "c1")
    net.maxpool(2, 2, capi.PADDING_SAME, "c1", "p1")
    net.conv(32, 3, 2, 1, 1, 0, capi.ACT_LRELU, "p1", "c2")
    net.activation(capi.ACT_RELU, "c2")
    net.conv(8, 1, 1, 0, 1, 0, capi.ACT_NONE, "c2", "c3")
    net.compile()
    return net


def test_resized_net_equals_a_net_built_for_that_extent():
    from bcnn_amd import capi
    big = _build(capi, 48, 40, 2)
    small = _build(capi, 32, 24, 1)          # same srand -> same Xavier draws -> same parameters
    assert big.resize(32, 24, 3, True) == 0
    names = ("input", "c1", "p1", "c2", "c3")
    for name in names:
        assert big.shape(big.index(name)) == small.shape(small.index(name)), name
    assert big.shape(big.index("input")) == (1, 3, 24, 32)
    assert big.shape(big.index("c2")) == (1, 32, 6, 8)
    x = np.random.RandomState(3).uniform(-1, 1, (1, 3, 24, 32)).astype(np.float32)
    for net in (big, small):
        net.data(0)[...] = x
        net.upload(0)
        net.forward()
    for name in ("c1", "p1", "c2", "c3"):
        a, b = big.index(name), small.index(name)
        big.download(a, False)
        small.download(b, False)
        assert np.array_equal(big.data(a), small.data(b)), name
    big.close()
    small.close()


def test_resize_rejects_nonsense():
    from bcnn_amd import capi
    net = _build(capi, 16, 16, 1)
    assert net.resize(0, 16, 3) != 0
    net.close()


def test_net_resized_to_a_larger_extent_equals_a_net_built_for_it():
    """the usual detection use of bcnn_resize_net: a batch-1 fully convolutional net grows (16 x 16 -> 48 x 40). The
    pooling node's index buffer and the batch-norm workspaces follow the new shapes (the reference leaves them at their old
    size and overruns them); results equal those of a net built for the larger extent."""
    from bcnn_amd import capi
    small = _build(capi, 16, 16, 1)
    big = _build(capi, 48, 40, 1)
    assert small.resize(48, 40, 3, True) == 0
    for name in ("input", "c1", "p1", "c2", "c3"):
        assert small.shape(small.index(name)) == big.shape(big.index(name)), name
    x = np.random.RandomState(5).uniform(-1, 1, (1, 3, 40, 48)).astype(np.float32)
    for net in (small, big):
        net.data(0)[...] = x
        net.upload(0)
        net.forward()
    for name in ("c1", "p1", "c2", "c3"):
        a, b = small.index(name), big.index(name)
        small.download(a, False)
        big.download(b, False)
        assert np.array_equal(small.data(a), big.data(b)), name
    # without need_realloc a shape that outgrew the layers' private buffers is refused, not run past them
    tiny = _build(capi, 16, 16, 1)
    assert tiny.resize(64, 64, 3, False) != 0
    for net in (small, big, tiny):
        net.close()
