"""Single-layer parity cases run on the UNMODIFIED reference (oracle/_ref) -- TEST INFRASTRUCTURE ONLY.

Each `ref_*` function takes a case dict (shape parameters + seeded input arrays made by `make_*`)
and returns the arrays the reference produces for it. tests/golden/make_golden.py stores
{inputs, outputs} as .npz fixtures; tests/ re-run the same dict through oracle/bcnn_oracle.c and
through the HIP C-ABI and compare.
"""
import ctypes as C

import numpy as np

from . import ref_bind as rb

F32 = np.float32


def _u(rs, shape, lo=-1.0, hi=1.0):
    return rs.uniform(lo, hi, size=shape).astype(F32)


def conv_out_hw(h, w, k, s, p):
    # src/layers/bcnn_conv_layer.c:126-134
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


# --------------------------------------------------------------------------------------------
# conv (+ fused BN, + fused activation)          src/layers/bcnn_conv_layer.c:367-587
# --------------------------------------------------------------------------------------------
def make_conv(seed, n, c, h, w, f, k, s, p, g=1, bn=0, act=rb.ACT_NONE, input_grad=True,
              mode=rb.MODE_TRAIN, bias_one=False, carry=False, name=None, via_model_file=False, forward_only=False):
    rs = np.random.RandomState(seed)
    oh, ow = conv_out_hw(h, w, k, s, p)
    cg = c // g
    a = np.sqrt(3.0 / (cg * k * k))
    case = dict(op="conv", n=n, c=c, h=h, w=w, f=f, k=k, s=s, p=p, g=g, bn=bn, act=act,
                input_grad=int(input_grad), mode=mode, name=name or "conv")
    if via_model_file:
        case["via_model_file"] = 1
    if forward_only:
        case["forward_only"] = 1
    case["x"] = _u(rs, (n, c, h, w))
    case["wt"] = _u(rs, (f, cg, k, k), -a, a)
    case["bias"] = _u(rs, (f,), -0.5, 0.5)
    if bias_one:
        case["bias"][1] = 1.0  # quirk 2: bcnn_add_scalar no-op for exactly 1.0f (bcnn_mat.c:381-383)
        case["bias"][2] = 0.0
    case["dy"] = (_u(rs, (n, f, oh, ow)) * 0.1).astype(F32)
    if carry:  # momentum carry already sitting in the gradient buffers (bcnn_learner.c:67-83)
        case["dw0"] = (_u(rs, (f, cg, k, k)) * 0.05).astype(F32)
        case["db0"] = (_u(rs, (f,)) * 0.05).astype(F32)
    if bn:
        case["run_mean0"] = (_u(rs, (f,)) * 0.1).astype(F32)
        case["run_var0"] = _u(rs, (f,), 0.5, 1.5)
        case["scales"] = _u(rs, (f,), 0.5, 1.5)
        if carry:
            case["dscales0"] = (_u(rs, (f,)) * 0.05).astype(F32)
    if act == rb.ACT_PRELU:
        # fused PReLU: per-filter slopes (bcnn_conv_layer.c:188-198, 476-481). The reference creates that tensor WITHOUT a
        # gradient buffer (:192 "no gradients") and its backward hands the NULL to bcnn_backward_activation_cpu, which
        # accumulates slope gradients into it (bcnn_activation_layer.c:214): a TRAIN-mode backward through such a node is a
        # segmentation fault in the reference. Fixtures of this kind are forward-only; the backward is pinned on the oracle.
        case["slopes"] = _u(rs, (f,), 0.05, 0.5)
        if not forward_only:
            case["dslopes0"] = (_u(rs, (f,)) * 0.1).astype(F32)
    return case


def ref_conv(case):
    cs = case
    net = rb.RefNet(mode=cs["mode"], w=cs["w"], h=cs["h"], c=cs["c"], n=cs["n"],
                    input_grad=bool(cs["input_grad"]))
    node = net.conv(cs["f"], cs["k"], cs["s"], cs["p"], cs["g"], cs["bn"], cs["act"], "input", "out")
    net.compile()
    out = {}
    i_x, i_w, i_b = net.node_src(node, 0), net.node_src(node, 1), net.node_src(node, 2)
    i_y = net.node_dst(node)
    net.data(i_x)[...] = cs["x"]
    net.data(i_w)[...] = cs["wt"]
    net.data(i_b).reshape(-1)[...] = cs["bias"]
    if cs["bn"]:
        i_rm, i_rv, i_sc = net.node_src(node, 3), net.node_src(node, 4), net.node_src(node, 5)
        net.data(i_rm).reshape(-1)[...] = cs["run_mean0"]
        net.data(i_rv).reshape(-1)[...] = cs["run_var0"]
        net.data(i_sc).reshape(-1)[...] = cs["scales"]
    i_sl = None
    if cs["act"] == rb.ACT_PRELU:  # the slopes of a fused PReLU: src slot 3 + 3 * batch_norm (bcnn_conv_layer.c:188-198)
        i_sl = net.node_src(node, 3 + 3 * int(cs["bn"]))
        net.data(i_sl).reshape(-1)[...] = cs["slopes"]
    if cs.get("via_model_file"):
        # PREDICT-mode 3x3/s1 convolutions of the reference read Winograd-transformed weights that only
        # bcnn_load_weights prepares (bcnn_net.c:1326-1346): round-trip the parameters through a model file
        import os
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.bcnnmodel")
            assert net.save_weights(path) == 0 and net.load_weights(path) == 0
    net.forward()
    out["y"] = net.data(i_y).copy()
    f = cs["f"]
    if cs["bn"]:
        out["run_mean"] = net.data(i_rm).reshape(-1).copy()
        out["run_var"] = net.data(i_rv).reshape(-1).copy()
        if cs["mode"] == rb.MODE_TRAIN:
            out["saved_mean"] = net.bn_field(node, 0, f).copy()
            out["saved_var"] = net.bn_field(node, 1, f).copy()
    if cs["mode"] == rb.MODE_TRAIN and not cs.get("forward_only"):
        assert i_sl is None, "the reference's backward through a fused PReLU dereferences a NULL gradient buffer"
        net.grad(i_y)[...] = cs["dy"]
        if "dw0" in cs:
            net.grad(i_w)[...] = cs["dw0"]
            net.grad(i_b).reshape(-1)[...] = cs["db0"]
            if cs["bn"]:
                net.grad(i_sc).reshape(-1)[...] = cs["dscales0"]
        if cs["input_grad"]:
            # garbage that a correct conv backward must OVERWRITE (col2im zero-fills, bcnn_mat.c:944)
            net.grad(i_x)[...] = 7.0
        net.backward()
        out["dy_out"] = net.grad(i_y).copy()
        out["dw"] = net.grad(i_w).copy()
        out["db"] = net.grad(i_b).reshape(-1).copy()
        if cs["input_grad"]:
            out["dx"] = net.grad(i_x).copy()
        if cs["bn"]:
            out["dscales"] = net.grad(i_sc).reshape(-1).copy()
            out["dmean"] = net.bn_field(node, 2, f).copy()
            out["dvar"] = net.bn_field(node, 3, f).copy()
    net.close()
    return out


# --------------------------------------------------------------------------------------------
# stand-alone batchnorm                         src/layers/bcnn_batchnorm_layer.c:196-332
# --------------------------------------------------------------------------------------------
def make_bn(seed, n, c, h, w, mode=rb.MODE_TRAIN, carry=False, shift=0.0, name=None):
    rs = np.random.RandomState(seed)
    case = dict(op="bn", n=n, c=c, h=h, w=w, mode=mode, name=name or "bn")
    case["x"] = (_u(rs, (n, c, h, w)) + shift).astype(F32)
    case["run_mean0"] = (_u(rs, (c,)) * 0.1).astype(F32)
    case["run_var0"] = _u(rs, (c,), 0.5, 1.5)
    case["scales"] = _u(rs, (c,), 0.5, 1.5)
    case["bias"] = _u(rs, (c,), -0.5, 0.5)
    case["dy"] = (_u(rs, (n, c, h, w)) * 0.1).astype(F32)
    if carry:
        case["db0"] = (_u(rs, (c,)) * 0.05).astype(F32)
        case["dscales0"] = (_u(rs, (c,)) * 0.05).astype(F32)
    return case


def _mk_tensor(arr, grad=None, name=b"t"):
    """A struct bcnn_tensor (public layout, inc/bcnn/bcnn.h:242-255) over numpy storage."""
    t = rb.Tensor()
    shp = arr.shape if arr.ndim == 4 else (1, 1, 1, arr.size)
    t.n, t.c, t.h, t.w = shp
    t.has_grad = 1 if grad is not None else 0
    t.name = name
    t.data = rb.fptr(arr)
    if grad is not None:
        t.grad_data = rb.fptr(grad)
    return t


def ref_bn(case):
    """Stand-alone BN node. `bcnn_add_batchnorm_layer` refuses to be the first node
    (bcnn_batchnorm_layer.c:42-44), so the node's worker functions bcnn_forward_batchnorm_cpu /
    bcnn_backward_batchnorm_cpu (non-static, bcnn_batchnorm_layer.h:52-67) are called directly with
    src != dst, exactly as bcnn_forward_batchnorm_layer_cpu does (:244-260)."""
    import ctypes as C
    cs = case
    L = rb.lib()
    c = cs["c"]
    x = cs["x"].copy(); dx = np.full_like(x, 7.0)
    y = np.zeros_like(x); dy = np.zeros_like(x)
    rm = cs["run_mean0"].copy(); rv = cs["run_var0"].copy()
    sc = cs["scales"].copy(); dsc = cs.get("dscales0", np.zeros(c, F32)).copy()
    b = cs["bias"].copy(); db = cs.get("db0", np.zeros(c, F32)).copy()
    sm = np.zeros(c, F32); dsm = np.zeros(c, F32)
    sv = np.zeros(c, F32); dsv = np.zeros(c, F32)
    xn = np.zeros_like(x); ws = np.zeros_like(x)
    t_x, t_y = _mk_tensor(x, dx), _mk_tensor(y, dy)
    t_rm, t_rv = _mk_tensor(rm), _mk_tensor(rv)
    t_sc, t_b = _mk_tensor(sc, dsc), _mk_tensor(b, db)
    t_sm, t_sv = _mk_tensor(sm, dsm), _mk_tensor(sv, dsv)
    P = C.POINTER(rb.Tensor)
    fp = C.POINTER(C.c_float)
    for fn in (L.bcnn_forward_batchnorm_cpu, L.bcnn_backward_batchnorm_cpu):
        fn.argtypes = [P, P, P, P, P, P, P, P, fp, fp, C.c_int, C.c_int]
        fn.restype = None
    args = [C.byref(t) for t in (t_x, t_y, t_rm, t_rv, t_sc, t_b, t_sm, t_sv)]
    L.bcnn_forward_batchnorm_cpu(*args, rb.fptr(xn), rb.fptr(ws), cs["mode"], 8)
    out = {"y": y.copy(), "run_mean": rm.copy(), "run_var": rv.copy()}
    if cs["mode"] == rb.MODE_TRAIN:
        out["saved_mean"] = sm.copy()
        out["saved_var"] = sv.copy()
        dy[...] = cs["dy"]
        L.bcnn_backward_batchnorm_cpu(*args, rb.fptr(xn), rb.fptr(ws), cs["mode"], 8)
        out["dy_out"] = dy.copy()
        out["dx"] = dx.copy()
        out["db"] = db.copy()
        out["dscales"] = dsc.copy()
        out["dmean"] = dsm.copy()
        out["dvar"] = dsv.copy()
    return out


# --------------------------------------------------------------------------------------------
# maxpool                                        src/layers/bcnn_maxpool_layer.c:145-191, 258-273
# --------------------------------------------------------------------------------------------
def make_maxpool(seed, n, c, h, w, k, s, padding=rb.PADDING_SAME, ties=False, name=None):
    rs = np.random.RandomState(seed)
    case = dict(op="maxpool", n=n, c=c, h=h, w=w, k=k, s=s, padding=padding, name=name or "maxpool")
    x = _u(rs, (n, c, h, w))
    if ties:  # few distinct values => many equal maxima: "first strict max wins" must hold
        x = np.round(x * 2.0).astype(F32) / 2.0
        x[0, 0, :, :] = -3.0e38  # below -FLT_MAX never happens; equal to very negative values
        x[-1, -1, 0, 0] = np.nan  # NaN never wins (val > max is false)
    case["x"] = x.astype(F32)
    case["dx0"] = (_u(rs, (n, c, h, w)) * 0.1).astype(F32)  # bwd accumulates into src.grad
    return case


def maxpool_out_hw(h, w, k, s, padding):
    # src/layers/bcnn_maxpool_layer.c:62-83
    def one(x):
        if padding == rb.PADDING_SAME:
            return (x + s - 1) // s
        if padding == rb.PADDING_VALID:
            return (x - k + s) // s
        return int(np.ceil(np.float32(x - k) / np.float32(s))) + 1
    return one(h), one(w)


def ref_maxpool(case):
    cs = case
    net = rb.RefNet(mode=rb.MODE_TRAIN, w=cs["w"], h=cs["h"], c=cs["c"], n=cs["n"], input_grad=True)
    node = net.maxpool(cs["k"], cs["s"], cs["padding"], "input", "out")
    net.compile()
    i_x, i_y = net.node_src(node, 0), net.node_dst(node)
    net.data(i_x)[...] = cs["x"]
    net.forward()
    out = {"y": net.data(i_y).copy(), "indexes": net.maxpool_indexes(node).copy()}
    rs = np.random.RandomState(1234)
    dy = (_u(rs, out["y"].shape) * 0.1).astype(F32)
    out["dy"] = dy  # stored with the outputs: its shape depends on the padding rule
    net.grad(i_y)[...] = dy
    net.grad(i_x)[...] = cs["dx0"]
    net.backward()
    out["dx"] = net.grad(i_x).copy()
    net.close()
    return out


# --------------------------------------------------------------------------------------------
# global avgpool                                 src/layers/bcnn_avgpool_layer.c:82-125
# --------------------------------------------------------------------------------------------
def make_avgpool(seed, n, c, h, w, name=None):
    rs = np.random.RandomState(seed)
    case = dict(op="avgpool", n=n, c=c, h=h, w=w, name=name or "avgpool")
    case["x"] = _u(rs, (n, c, h, w))
    case["dy"] = (_u(rs, (n, c, 1, 1)) * 0.1).astype(F32)
    case["dx0"] = (_u(rs, (n, c, h, w)) * 0.1).astype(F32)
    return case


def ref_avgpool(case):
    cs = case
    net = rb.RefNet(mode=rb.MODE_TRAIN, w=cs["w"], h=cs["h"], c=cs["c"], n=cs["n"], input_grad=True)
    node = net.avgpool("input", "out")
    net.compile()
    i_x, i_y = net.node_src(node, 0), net.node_dst(node)
    net.data(i_x)[...] = cs["x"]
    net.forward()
    out = {"y": net.data(i_y).copy()}
    net.grad(i_y)[...] = cs["dy"]
    net.grad(i_x)[...] = cs["dx0"]
    net.backward()
    out["dx"] = net.grad(i_x).copy()
    net.close()
    return out


# --------------------------------------------------------------------------------------------
# activation map                                 src/layers/bcnn_activation_layer.c:90-146, 165-226
# (called directly: the stand-alone node segfaults on CPU for non-PReLU)
# --------------------------------------------------------------------------------------------
def make_act(seed, act, n=2, c=3, hw=37, name=None):
    rs = np.random.RandomState(seed)
    case = dict(op="act", act=act, n=n, c=c, hw=hw, name=name or "act")
    x = (_u(rs, (n, c, hw)) * 3.0).astype(F32)
    x.reshape(-1)[:6] = [0.0, -0.0, 1.0, -1.0, 0.5, 2.0]
    case["x"] = x
    case["dy"] = _u(rs, (n, c, hw))
    case["slopes"] = _u(rs, (c,), 0.05, 0.5)
    case["dslopes0"] = (_u(rs, (c,)) * 0.1).astype(F32)
    return case


def ref_act(case):
    cs = case
    L = rb.lib()
    y = cs["x"].copy()
    sz = y.size
    L.bcnn_forward_activation_cpu(rb.fptr(y), sz, rb.fptr(cs["slopes"]), cs["hw"], cs["c"], cs["act"])
    dx = cs["dy"].copy()
    ds = cs["dslopes0"].copy()
    L.bcnn_backward_activation_cpu(rb.fptr(y), rb.fptr(dx), sz, rb.fptr(cs["slopes"]), rb.fptr(ds),
                                   cs["hw"], cs["c"], cs["act"])
    return {"y": y, "dx": dx, "dslopes": ds}


# --------------------------------------------------------------------------------------------
# depthwise conv                                 src/layers/bcnn_depthwise_conv_layer.c:165-547
# --------------------------------------------------------------------------------------------
def make_dw(seed, n, c, h, w, k, s, p, act=rb.ACT_NONE, input_grad=True, bias_one=False, name=None):
    rs = np.random.RandomState(seed)
    oh, ow = conv_out_hw(h, w, k, s, p)
    a = np.sqrt(3.0 / (k * k))
    case = dict(op="dw", n=n, c=c, h=h, w=w, k=k, s=s, p=p, act=act, input_grad=int(input_grad),
                name=name or "dw")
    case["x"] = _u(rs, (n, c, h, w))
    case["wt"] = _u(rs, (c, k, k), -a, a)
    case["bias"] = _u(rs, (c,), -0.5, 0.5)
    if bias_one:
        case["bias"][0] = 1.0
    case["dy"] = (_u(rs, (n, c, oh, ow)) * 0.1).astype(F32)
    case["dw0"] = (_u(rs, (c, k, k)) * 0.05).astype(F32)
    case["db0"] = (_u(rs, (c,)) * 0.05).astype(F32)
    case["dx0"] = (_u(rs, (n, c, h, w)) * 0.1).astype(F32)  # dX accumulates (no zeroing)
    return case


def ref_dw(case):
    cs = case
    net = rb.RefNet(mode=rb.MODE_TRAIN, w=cs["w"], h=cs["h"], c=cs["c"], n=cs["n"],
                    input_grad=bool(cs["input_grad"]))
    node = net.depthwise(cs["k"], cs["s"], cs["p"], cs["act"], "input", "out")
    net.compile()
    i_x, i_w, i_b = net.node_src(node, 0), net.node_src(node, 1), net.node_src(node, 2)
    i_y = net.node_dst(node)
    net.data(i_x)[...] = cs["x"]
    net.data(i_w).reshape(-1)[...] = cs["wt"].reshape(-1)
    net.data(i_b).reshape(-1)[...] = cs["bias"]
    net.forward()
    out = {"y": net.data(i_y).copy()}
    net.grad(i_y)[...] = cs["dy"]
    net.grad(i_w).reshape(-1)[...] = cs["dw0"].reshape(-1)
    net.grad(i_b).reshape(-1)[...] = cs["db0"]
    if cs["input_grad"]:
        net.grad(i_x)[...] = cs["dx0"]
    net.backward()
    out["dy_out"] = net.grad(i_y).copy()
    out["dw"] = net.grad(i_w).reshape(cs["wt"].shape).copy()
    out["db"] = net.grad(i_b).reshape(-1).copy()
    if cs["input_grad"]:
        out["dx"] = net.grad(i_x).copy()
    net.close()
    return out


# --------------------------------------------------------------------------------------------
# raw kernels: im2col / col2im / gemm            src/kernels/bcnn_mat.c:817-970, 2627-2650
# --------------------------------------------------------------------------------------------
def make_im2col(seed, c, h, w, k, s, p, name=None):
    rs = np.random.RandomState(seed)
    oh, ow = conv_out_hw(h, w, k, s, p)
    case = dict(op="im2col", c=c, h=h, w=w, k=k, s=s, p=p, name=name or "im2col")
    case["x"] = _u(rs, (c, h, w))
    case["col_in"] = _u(rs, (c * k * k, oh * ow))
    return case


def ref_im2col(case):
    cs = case
    L = rb.lib()
    oh, ow = conv_out_hw(cs["h"], cs["w"], cs["k"], cs["s"], cs["p"])
    col = np.full((cs["c"] * cs["k"] * cs["k"], oh * ow), 9.0, F32)
    L.bcnn_im2col(rb.fptr(cs["x"]), cs["c"], cs["h"], cs["w"], cs["k"], cs["p"], cs["s"], rb.fptr(col))
    im = np.full((cs["c"], cs["h"], cs["w"]), 9.0, F32)
    L.bcnn_col2im(rb.fptr(cs["col_in"]), cs["c"], cs["h"], cs["w"], cs["k"], cs["p"], cs["s"], rb.fptr(im))
    return {"col": col, "im": im}


def make_gemm(seed, ta, tb, m, n, k, alpha=1.0, beta=1.0, name=None):
    rs = np.random.RandomState(seed)
    case = dict(op="gemm", ta=ta, tb=tb, m=m, n=n, k=k, alpha=float(alpha), beta=float(beta),
                name=name or "gemm")
    case["A"] = _u(rs, (k, m) if ta else (m, k))
    case["B"] = _u(rs, (n, k) if tb else (k, n))
    case["C0"] = _u(rs, (m, n))
    return case


def ref_gemm(case):
    cs = case
    net = rb.RefNet(mode=rb.MODE_TRAIN, w=4, h=4, c=1, n=1)
    Cm = cs["C0"].copy()
    lda = cs["A"].shape[1]
    ldb = cs["B"].shape[1]
    net.L.ref_gemm(net.net, cs["ta"], cs["tb"], cs["m"], cs["n"], cs["k"], cs["alpha"],
                   rb.fptr(cs["A"]), lda, rb.fptr(cs["B"]), ldb, cs["beta"], rb.fptr(Cm), cs["n"])
    net.close()
    return {"C": Cm}


# --------------------------------------------------------------------------------------------
# optimizer steps                                 src/bcnn_learner.c:67-83 (SGD), :106-131 (Adam)
# --------------------------------------------------------------------------------------------
def make_optim(seed, kind, wsize, bsize, steps=3, batch=4, lr=0.01, momentum=0.9, decay=5e-4, beta1=0.9,
               beta2=0.999, name=None):
    """`steps` updates; before step t the fresh gradients dw_t / db_t are ADDED to the gradient buffers (what a
    backward pass does), so the momentum carry (SGD, biases) and the zeroing (Adam weights) are both exercised.
    `iter` follows the call sites: learner->seen = (t+1)*batch."""
    rs = np.random.RandomState(seed)
    case = dict(op="optim", kind=kind, wsize=wsize, bsize=bsize, steps=steps, batch=batch, lr=float(lr),
                momentum=float(momentum), decay=float(decay), beta1=float(beta1), beta2=float(beta2),
                name=name or "optim_" + kind)
    case["w0"] = _u(rs, (wsize,))
    case["b0"] = _u(rs, (bsize,))
    case["dw_steps"] = _u(rs, (steps, wsize)) * np.float32(batch)
    case["db_steps"] = _u(rs, (steps, bsize)) * np.float32(batch)
    # a few exact zeros: Adam's m/(sqrt(v)+1e-7) at v == 0
    case["dw_steps"][:, :3] = 0.0
    return case


def run_optim(case, sgd, adam):
    """shared by the reference, the oracle and the HIP runner: sgd/adam are callables on numpy-like buffers"""
    cs = case
    n = int(cs["steps"])
    w, b = cs["w0"].copy(), cs["b0"].copy()
    dw, db = np.zeros_like(w), np.zeros_like(b)
    m, v = np.zeros_like(w), np.zeros_like(w)
    for t in range(n):
        dw += cs["dw_steps"][t]
        db += cs["db_steps"][t]
        if str(cs["kind"]) == "adam":
            adam(w, b, dw, db, m, v, int(cs["batch"]), (t + 1) * int(cs["batch"]), float(cs["beta1"]),
                 float(cs["beta2"]), float(cs["lr"]), float(cs["momentum"]), float(cs["decay"]))
        else:
            sgd(w, b, dw, db, int(cs["batch"]), float(cs["lr"]), float(cs["momentum"]), float(cs["decay"]))
    out = {"w": w, "b": b, "dw": dw, "db": db}
    if str(cs["kind"]) == "adam":
        out.update(adam_m=m, adam_v=v)
    return out


def ref_optim(case):
    L = rb.lib()
    fp, i, f = C.POINTER(C.c_float), C.c_int, C.c_float
    L.bcnn_sgd_update_cpu.argtypes = [fp, fp, fp, fp, i, i, i, f, f, f]
    L.bcnn_sgd_update_cpu.restype = None
    L.bcnn_adam_update_cpu.argtypes = [fp, fp, fp, fp, fp, fp, i, i, i, i, f, f, f, f, f]
    L.bcnn_adam_update_cpu.restype = None

    def sgd(w, b, dw, db, batch, lr, mom, decay):
        L.bcnn_sgd_update_cpu(rb.fptr(w), rb.fptr(b), rb.fptr(dw), rb.fptr(db), w.size, b.size, batch, lr, mom, decay)

    def adam(w, b, dw, db, m, v, batch, it, b1, b2, lr, mom, decay):
        L.bcnn_adam_update_cpu(rb.fptr(w), rb.fptr(b), rb.fptr(dw), rb.fptr(db), rb.fptr(m), rb.fptr(v), w.size,
                               b.size, batch, it, b1, b2, lr, mom, decay)
    return run_optim(case, sgd, adam)


RUNNERS = {"optim": ref_optim, "conv": ref_conv, "bn": ref_bn, "maxpool": ref_maxpool, "avgpool": ref_avgpool,
           "act": ref_act, "dw": ref_dw, "im2col": ref_im2col, "gemm": ref_gemm}


def run_ref(case):
    return RUNNERS[case["op"]](case)
