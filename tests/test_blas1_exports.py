"""The stand-alone per-channel / BLAS-1 exports of include/bcnn_hip.h (bcnn_hip_add_bias, _scales, _grad_scales,
_grad_bias, _scal) and the event API. The conv path fuses these operations into its epilogues, so the goldens only
reach them indirectly; here each one is called through the C-ABI and compared with the reference's own helpers
(bcnn_add_bias / bcnn_scales / bcnn_grad_scales / bcnn_grad_bias / bcnn_scal, src/kernels/bcnn_mat.c:319-364,
761-811) from oracle/_ref, and with the C restatement. Shapes cover the 16-byte vector path (HW % 4 == 0, large
planes), the scalar path (odd HW) and the flat-map path (small planes)."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc_bind as ob
from oracle import ref_bind as rb

SHAPES = [(3, 5, 37), (2, 4, 1024), (2, 3, 4100), (4, 8, 49)]  # (n, c, hw)


def _case(n, c, hw, seed):
    rs = np.random.RandomState(seed)
    x = rs.uniform(-1, 1, (n, c, hw)).astype(np.float32)
    g = rs.uniform(-1, 1, (n, c, hw)).astype(np.float32)
    v = rs.uniform(-0.5, 0.5, c).astype(np.float32)
    v[0] = 1.0   # bcnn_add_scalar's AVX path adds nothing for exactly 1.0f (quirk 2); scal by 1 is a no-op
    if c > 2:
        v[2] = 0.0
    acc0 = rs.uniform(-1, 1, c).astype(np.float32)  # the gradient helpers accumulate onto what is there
    return x, g, v, acc0


def _ref_results(x, g, v, acc0):
    L = rb.lib()
    n, c, hw = x.shape
    out = {}
    y = x.copy(); L.bcnn_add_bias(rb.fptr(y), rb.fptr(v), n, c, hw, 1); out["add_bias"] = y
    y = x.copy(); L.bcnn_scales(rb.fptr(y), rb.fptr(v), n, c, hw, 1); out["scales"] = y
    a = acc0.copy(); L.bcnn_grad_bias(rb.fptr(a), rb.fptr(g), n, c, hw); out["grad_bias"] = a
    a = acc0.copy(); L.bcnn_grad_scales(rb.fptr(x), rb.fptr(g), n, c, hw, rb.fptr(a)); out["grad_scales"] = a
    for alpha in (0.0, 1.0, 0.37):
        y = x.copy().reshape(-1); L.bcnn_scal(y.size, alpha, rb.fptr(y)); out["scal_%g" % alpha] = y.reshape(x.shape)
    return out


def _orc_results(x, g, v, acc0):
    L = ob.lib()
    n, c, hw = x.shape
    out = {}
    y = x.copy(); L.orc_add_bias(ob.P(y), ob.P(v), n, c, hw); out["add_bias"] = y
    y = x.copy(); L.orc_scales(ob.P(y), ob.P(v), n, c, hw); out["scales"] = y
    a = acc0.copy(); L.orc_grad_bias(ob.P(a), ob.P(g), n, c, hw); out["grad_bias"] = a
    a = acc0.copy(); L.orc_grad_scales(ob.P(x), ob.P(g), n, c, hw, ob.P(a)); out["grad_scales"] = a
    return out


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(float(np.abs(b).max()), 1e-30))


@pytest.mark.parametrize("n,c,hw", SHAPES)
def test_oracle_per_channel_helpers_match_the_reference(n, c, hw):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    x, g, v, acc0 = _case(n, c, hw, 3)
    ref, orc = _ref_results(x, g, v, acc0), _orc_results(x, g, v, acc0)
    assert np.array_equal(orc["add_bias"], ref["add_bias"])   # elementwise: bit-exact
    assert np.array_equal(orc["scales"], ref["scales"])
    assert _rel(orc["grad_bias"], ref["grad_bias"]) <= 2e-6
    assert _rel(orc["grad_scales"], ref["grad_scales"]) <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,hw", SHAPES)
def test_hip_per_channel_exports_match_the_reference(n, c, hw):
    import torch
    from bcnn_amd import _lib
    L = _lib.load()
    x, g, v, acc0 = _case(n, c, hw, 3)
    want = _ref_results(x, g, v, acc0) if rb.available() else _orc_results(x, g, v, acc0)
    dev = "cuda:0"
    D = lambda a: torch.from_numpy(a.copy()).to(dev)
    H = lambda t: t.cpu().numpy()
    vd, gd = D(v), D(g)
    y = D(x); L.bcnn_hip_add_bias(y.data_ptr(), vd.data_ptr(), n, c, hw); L.bcnn_hip_sync()
    assert np.array_equal(H(y), want["add_bias"])
    y = D(x); L.bcnn_hip_scales(y.data_ptr(), vd.data_ptr(), n, c, hw); L.bcnn_hip_sync()
    assert np.array_equal(H(y), want["scales"])
    a = D(acc0); L.bcnn_hip_grad_bias(a.data_ptr(), gd.data_ptr(), n, c, hw); L.bcnn_hip_sync()
    assert _rel(H(a), want["grad_bias"]) <= 1e-5
    xd = D(x)
    a = D(acc0); L.bcnn_hip_grad_scales(xd.data_ptr(), gd.data_ptr(), n, c, hw, a.data_ptr()); L.bcnn_hip_sync()
    assert _rel(H(a), want["grad_scales"]) <= 1e-5
    for alpha in (0.0, 1.0, 0.37):
        y = D(x); L.bcnn_hip_scal(y.numel(), C.c_float(alpha), y.data_ptr()); L.bcnn_hip_sync()
        exp = want.get("scal_%g" % alpha)
        if exp is None:
            exp = (x * np.float32(alpha)) if alpha != 1.0 else x
        assert np.array_equal(H(y), exp), alpha


@pytest.mark.gpu
def test_event_api_orders_and_times_work_on_the_launch_stream():
    import torch
    from bcnn_amd import _lib
    L = _lib.load()
    st = L.bcnn_hip_stream_create()
    prev = L.bcnn_hip_get_stream()
    L.bcnn_hip_set_stream(st)
    try:
        assert L.bcnn_hip_get_stream() == st
        e0, e1 = L.bcnn_hip_event_create(), L.bcnn_hip_event_create()
        n = 1 << 26
        buf = L.bcnn_hip_malloc_f32(n)
        L.bcnn_hip_event_record(e0)
        for _ in range(8):
            L.bcnn_hip_fill_f32(buf, n, C.c_float(1.5))       # 8 x 256 MB of stores on OUR stream
        L.bcnn_hip_event_record(e1)
        L.bcnn_hip_event_sync(e1)
        ms = L.bcnn_hip_event_elapsed_ms(e0, e1)
        assert 0.05 < ms < 1000.0, ms                           # 2 GB cannot be written in less than ~0.25 ms
        host = np.empty(16, np.float32)
        L.bcnn_hip_memcpy_d2h(host.ctypes.data, buf, 64)
        assert np.all(host == 1.5)                              # the event really covered the work
        L.bcnn_hip_event_destroy(e0)
        L.bcnn_hip_event_destroy(e1)
        L.bcnn_hip_free(buf)
    finally:
        L.bcnn_hip_sync()
        L.bcnn_hip_set_stream(prev)
        L.bcnn_hip_stream_destroy(st)
