"""ctypes binding of oracle/libbcnn_oracle.so (this repo's C restatement) -- TEST INFRASTRUCTURE ONLY.

`run_oracle(case)` evaluates a parity case (same dicts as oracle/ref_cases.py) and returns the same
output keys the reference runner returns, so tests compare key by key.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORC_SO = os.path.join(_HERE, "libbcnn_oracle.so")
F32 = np.float32
MODE_PREDICT, MODE_TRAIN, MODE_VALID = 0, 1, 2

_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORC_SO):
        build()
    L = C.CDLL(ORC_SO)
    i, f = C.c_int, C.c_float
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    sig = {
        "orc_im2col": [fp, i, i, i, i, i, i, fp],
        "orc_col2im": [fp, i, i, i, i, i, i, fp],
        "orc_gemm": [i, i, i, i, i, f, fp, i, fp, i, f, fp, i],
        "orc_add_bias": [fp, fp, i, i, i],
        "orc_grad_bias": [fp, fp, i, i, i],
        "orc_scales": [fp, fp, i, i, i],
        "orc_grad_scales": [fp, fp, i, i, i, fp],
        "orc_act_forward": [fp, i, fp, i, i, i],
        "orc_act_backward": [fp, fp, i, fp, fp, i, i, i],
        "orc_bn_forward": [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, i, i, i, i],
        "orc_bn_backward": [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, i, i, i],
        "orc_conv_forward": [fp, fp, fp, fp, i, i, i, i, i, i, i, i, i, i, fp, i, fp, fp, fp, fp, fp,
                             fp, fp, i, fp],
        "orc_conv_backward": [fp, fp, fp, fp, fp, fp, fp, i, i, i, i, i, i, i, i, i, i, fp, fp, i, fp,
                              fp, fp, fp, fp, fp, fp, fp, fp],
        "orc_maxpool_forward": [fp, fp, ip, i, i, i, i, i, i, i, i],
        "orc_maxpool_backward": [fp, ip, fp, i],
        "orc_avgpool_forward": [fp, fp, i, i, i, i],
        "orc_avgpool_backward": [fp, fp, i, i, i, i],
        "orc_dw_forward": [fp, fp, fp, fp, i, i, i, i, i, i, i, i],
        "orc_dw_backward": [fp, fp, fp, fp, fp, fp, fp, i, i, i, i, i, i, i, i],
        "orc_sgd_update": [fp, fp, fp, fp, i, i, i, f, f, f],
        "orc_adam_update": [fp, fp, fp, fp, fp, fp, i, i, i, i, f, f, f, f, f],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = None
    _lib = L
    return L


def P(a):
    if a is None:
        return None
    assert a.dtype == F32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(C.POINTER(C.c_float))


def PI(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int))


def conv_out_hw(h, w, k, s, p):
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


def maxpool_out_hw(h, w, k, s, padding):
    """src/layers/bcnn_maxpool_layer.c:62-83 (SAME / VALID / CAFFE)."""
    def one(x):
        if padding == 0:
            return (x + s - 1) // s
        if padding == 1:
            return (x - k + s) // s
        return int(np.ceil(np.float32(x - k) / np.float32(s))) + 1
    return one(h), one(w)


def orc_conv(cs):
    L = lib()
    n, c, h, w, f, k, s, p, g = (int(cs[q]) for q in ("n", "c", "h", "w", "f", "k", "s", "p", "g"))
    bn, act, mode = int(cs["bn"]), int(cs["act"]), int(cs["mode"])
    oh, ow = conv_out_hw(h, w, k, s, p)
    y = np.zeros((n, f, oh, ow), F32)
    col = np.zeros(((c // g) * k * k * oh * ow,), F32)
    z = lambda: np.zeros(f, F32)
    rm = cs["run_mean0"].copy() if bn else z()
    rv = cs["run_var0"].copy() if bn else z()
    sc = cs["scales"].copy() if bn else z()
    sm, sv = z(), z()
    xn = np.zeros_like(y) if bn else None
    ws = np.zeros_like(y) if bn else None
    slopes = cs.get("slopes")
    L.orc_conv_forward(P(cs["x"]), P(cs["wt"]), P(cs["bias"]), P(y), n, c, h, w, f, k, s, p, g, act,
                       P(slopes), bn, P(rm), P(rv), P(sc), P(sm), P(sv), P(xn), P(ws), mode, P(col))
    out = {"y": y.copy()}
    if bn:
        out["run_mean"], out["run_var"] = rm.copy(), rv.copy()
        if mode == MODE_TRAIN:
            out["saved_mean"], out["saved_var"] = sm.copy(), sv.copy()
    if mode == MODE_TRAIN and not int(cs.get("forward_only", 0)):
        dy = cs["dy"].copy()
        dw = cs["dw0"].copy() if "dw0" in cs else np.zeros_like(cs["wt"])
        db = cs["db0"].copy() if "db0" in cs else z()
        dsc = cs["dscales0"].copy() if "dscales0" in cs else z()
        dm, dv = z(), z()
        dx = np.full_like(cs["x"], 7.0) if int(cs["input_grad"]) else None
        dsl = cs["dslopes0"].copy() if "dslopes0" in cs else None
        L.orc_conv_backward(P(cs["x"]), P(cs["wt"]), P(y), P(dy), P(dx), P(dw), P(db), n, c, h, w, f,
                            k, s, p, g, act, P(slopes), P(dsl), bn, P(sc), P(dsc), P(sm), P(sv), P(dm),
                            P(dv), P(xn), P(ws), P(col))
        out.update(dy_out=dy, dw=dw, db=db)
        if dsl is not None:
            out["dslopes"] = dsl
        if dx is not None:
            out["dx"] = dx
        if bn:
            out.update(dscales=dsc, dmean=dm, dvar=dv)
    return out


def orc_bn(cs):
    L = lib()
    n, c, h, w, mode = (int(cs[q]) for q in ("n", "c", "h", "w", "mode"))
    x = cs["x"]
    y = np.zeros_like(x)
    rm, rv = cs["run_mean0"].copy(), cs["run_var0"].copy()
    sm, sv = np.zeros(c, F32), np.zeros(c, F32)
    xn, ws = np.zeros_like(x), np.zeros_like(x)
    L.orc_bn_forward(P(x), P(y), P(rm), P(rv), P(cs["scales"]), P(cs["bias"]), P(sm), P(sv), P(xn),
                     P(ws), n, c, h * w, mode)
    out = {"y": y.copy(), "run_mean": rm, "run_var": rv}
    if mode == MODE_TRAIN:
        out["saved_mean"], out["saved_var"] = sm.copy(), sv.copy()
        dy = cs["dy"].copy()
        dx = np.full_like(x, 7.0)
        db = cs["db0"].copy() if "db0" in cs else np.zeros(c, F32)
        dsc = cs["dscales0"].copy() if "dscales0" in cs else np.zeros(c, F32)
        dm, dv = np.zeros(c, F32), np.zeros(c, F32)
        L.orc_bn_backward(P(dy), P(dx), P(cs["scales"]), P(dsc), P(db), P(sm), P(sv), P(dm), P(dv),
                          P(xn), P(ws), n, c, h * w)
        out.update(dy_out=dy, dx=dx, db=db, dscales=dsc, dmean=dm, dvar=dv)
    return out


def orc_maxpool(cs, dy=None):
    L = lib()
    n, c, h, w, k, s, padding = (int(cs[q]) for q in ("n", "c", "h", "w", "k", "s", "padding"))
    oh, ow = maxpool_out_hw(h, w, k, s, padding)
    y = np.zeros((n, c, oh, ow), F32)
    idx = np.zeros((n, c, oh, ow), np.int32)
    L.orc_maxpool_forward(P(cs["x"]), P(y), PI(idx), n, c, h, w, oh, ow, k, s)
    out = {"y": y, "indexes": idx}
    if dy is not None:
        dx = cs["dx0"].copy()
        L.orc_maxpool_backward(P(np.ascontiguousarray(dy)), PI(idx), P(dx), y.size)
        out["dx"] = dx
        out["dy"] = dy
    return out


def orc_avgpool(cs):
    L = lib()
    n, c, h, w = (int(cs[q]) for q in ("n", "c", "h", "w"))
    y = np.zeros((n, c, 1, 1), F32)
    L.orc_avgpool_forward(P(cs["x"]), P(y), n, c, h, w)
    dx = cs["dx0"].copy()
    L.orc_avgpool_backward(P(cs["dy"]), P(dx), n, c, h, w)
    return {"y": y, "dx": dx}


def orc_act(cs):
    L = lib()
    y = cs["x"].copy()
    L.orc_act_forward(P(y), y.size, P(cs["slopes"]), int(cs["hw"]), int(cs["c"]), int(cs["act"]))
    dx = cs["dy"].copy()
    ds = cs["dslopes0"].copy()
    L.orc_act_backward(P(y), P(dx), y.size, P(cs["slopes"]), P(ds), int(cs["hw"]), int(cs["c"]),
                       int(cs["act"]))
    return {"y": y, "dx": dx, "dslopes": ds}


def orc_dw(cs):
    L = lib()
    n, c, h, w, k, s, p, act = (int(cs[q]) for q in ("n", "c", "h", "w", "k", "s", "p", "act"))
    oh, ow = conv_out_hw(h, w, k, s, p)
    y = np.zeros((n, c, oh, ow), F32)
    L.orc_dw_forward(P(cs["x"]), P(cs["wt"]), P(cs["bias"]), P(y), n, c, h, w, k, s, p, act)
    out = {"y": y.copy()}
    dy = cs["dy"].copy()
    dw, db = cs["dw0"].copy(), cs["db0"].copy()
    dx = cs["dx0"].copy() if int(cs["input_grad"]) else None
    L.orc_dw_backward(P(cs["x"]), P(cs["wt"]), P(y), P(dy), P(dx), P(dw), P(db), n, c, h, w, k, s, p,
                      act)
    out.update(dy_out=dy, dw=dw, db=db)
    if dx is not None:
        out["dx"] = dx
    return out


def orc_im2col(cs):
    L = lib()
    c, h, w, k, s, p = (int(cs[q]) for q in ("c", "h", "w", "k", "s", "p"))
    oh, ow = conv_out_hw(h, w, k, s, p)
    col = np.full((c * k * k, oh * ow), 9.0, F32)
    L.orc_im2col(P(cs["x"]), c, h, w, k, p, s, P(col))
    im = np.full((c, h, w), 9.0, F32)
    L.orc_col2im(P(cs["col_in"]), c, h, w, k, p, s, P(im))
    return {"col": col, "im": im}


def orc_gemm(cs):
    L = lib()
    Cm = cs["C0"].copy()
    L.orc_gemm(int(cs["ta"]), int(cs["tb"]), int(cs["m"]), int(cs["n"]), int(cs["k"]),
               float(cs["alpha"]), P(cs["A"]), cs["A"].shape[1], P(cs["B"]), cs["B"].shape[1],
               float(cs["beta"]), P(Cm), int(cs["n"]))
    return {"C": Cm}


def orc_optim(cs):
    from oracle.ref_cases import run_optim
    L = lib()

    def sgd(w, b, dw, db, batch, lr, mom, decay):
        L.orc_sgd_update(P(w), P(b), P(dw), P(db), w.size, b.size, batch, lr, mom, decay)

    def adam(w, b, dw, db, m, v, batch, it, b1, b2, lr, mom, decay):
        L.orc_adam_update(P(w), P(b), P(dw), P(db), P(m), P(v), w.size, b.size, batch, it, b1, b2, lr, mom, decay)
    return run_optim(cs, sgd, adam)


def run_oracle(case, golden_out=None):
    op = str(case["op"])
    if op == "maxpool":
        return orc_maxpool(case, None if golden_out is None else golden_out.get("dy"))
    return {"conv": orc_conv, "bn": orc_bn, "avgpool": orc_avgpool, "act": orc_act, "dw": orc_dw,
            "im2col": orc_im2col, "gemm": orc_gemm, "optim": orc_optim}[op](case)
