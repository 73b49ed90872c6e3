"""Run a parity case (same dicts as oracle/ref_cases.py / tests/golden) through the HIP C-ABI and
return the same output keys. Needs a GPU; imported only by -m gpu tests."""
import numpy as np
import torch

from bcnn_amd import ops

DEV = "cuda:0"
MODE_TRAIN = 1


def D(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def H(t):
    return t.detach().cpu().numpy()


def Z(*shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype, device=DEV)


def hip_conv(cs):
    n, c, h, w, f, k, s, p, g = (int(cs[q]) for q in ("n", "c", "h", "w", "f", "k", "s", "p", "g"))
    bn, act, mode = int(cs["bn"]), int(cs["act"]), int(cs["mode"])
    oh, ow = ops.conv_out_hw(h, w, k, s, p)
    x, wt, bias = D(cs["x"]), D(cs["wt"]), D(cs["bias"])
    y = torch.full((n, f, oh, ow), 3.0, device=DEV)  # garbage: forward must overwrite
    b = None
    if bn:
        b = dict(run_mean=D(cs["run_mean0"]), run_var=D(cs["run_var0"]), scales=D(cs["scales"]),
                 saved_mean=Z(f), saved_var=Z(f), workspace=torch.full((n, f, oh, ow), 5.0, device=DEV))
    slopes = D(cs["slopes"]) if "slopes" in cs else None
    ops.conv_forward(x, wt, bias, y, k, s, p, g, act, slopes, b, mode)
    out = {"y": H(y)}
    if bn:
        out["run_mean"], out["run_var"] = H(b["run_mean"]), H(b["run_var"])
        if mode == MODE_TRAIN:
            out["saved_mean"], out["saved_var"] = H(b["saved_mean"]), H(b["saved_var"])
    if mode == MODE_TRAIN and not int(cs.get("forward_only", 0)):
        dy = D(cs["dy"])
        dw = D(cs["dw0"]) if "dw0" in cs else Z(*cs["wt"].shape)
        db = D(cs["db0"]) if "db0" in cs else Z(f)
        dx = torch.full((n, c, h, w), 7.0, device=DEV) if int(cs["input_grad"]) else None
        if bn:
            b.update(dscales=D(cs["dscales0"]) if "dscales0" in cs else Z(f), dmean=Z(f), dvar=Z(f))
        ws = Z(max(1, ops.conv_workspace_size(n, c, h, w, f, k, s, p, g)))
        dsl = D(cs["dslopes0"]) if "dslopes0" in cs else None
        ops.conv_backward(x, wt, y, dy, dx, dw, db, k, s, p, g, act, ws, slopes, dsl, b, bias)
        out.update(dy_out=H(dy), dw=H(dw), db=H(db))
        if dsl is not None:
            out["dslopes"] = H(dsl)
        if dx is not None:
            out["dx"] = H(dx)
        if bn:
            out.update(dscales=H(b["dscales"]), dmean=H(b["dmean"]), dvar=H(b["dvar"]))
    return out


def hip_bn(cs):
    n, c, h, w, mode = (int(cs[q]) for q in ("n", "c", "h", "w", "mode"))
    x = D(cs["x"])
    y = torch.full_like(x, 3.0)
    rm, rv, sc, bias = D(cs["run_mean0"]), D(cs["run_var0"]), D(cs["scales"]), D(cs["bias"])
    sm, sv, ws = Z(c), Z(c), torch.full_like(x, 5.0)
    ops.batchnorm_forward(x, y, rm, rv, sc, bias, sm, sv, ws, mode)
    out = {"y": H(y), "run_mean": H(rm), "run_var": H(rv)}
    if mode == MODE_TRAIN:
        out["saved_mean"], out["saved_var"] = H(sm), H(sv)
        dy, dx = D(cs["dy"]), torch.full_like(x, 7.0)
        db = D(cs["db0"]) if "db0" in cs else Z(c)
        dsc = D(cs["dscales0"]) if "dscales0" in cs else Z(c)
        dm, dv = Z(c), Z(c)
        ops.batchnorm_backward(dy, dx, sc, dsc, db, sm, sv, dm, dv, ws)
        out.update(dy_out=H(dy), dx=H(dx), db=H(db), dscales=H(dsc), dmean=H(dm), dvar=H(dv))
    return out


def hip_maxpool(cs, exp):
    x = D(cs["x"])
    n, c = x.shape[:2]
    oh, ow = exp["y"].shape[2:]
    y = torch.full((n, c, oh, ow), 3.0, device=DEV)
    idx = torch.full((n, c, oh, ow), -7, dtype=torch.int32, device=DEV)
    ops.maxpool_forward(x, y, idx, int(cs["k"]), int(cs["s"]))
    dx = D(cs["dx0"])
    ops.maxpool_backward(D(exp["dy"]), idx, dx, int(cs["k"]), int(cs["s"]))
    return {"y": H(y), "indexes": H(idx), "dx": H(dx)}


def hip_avgpool(cs):
    x = D(cs["x"])
    y = torch.full((x.shape[0], x.shape[1], 1, 1), 3.0, device=DEV)
    ops.avgpool_forward(x, y)
    dx = D(cs["dx0"])
    ops.avgpool_backward(D(cs["dy"]), dx)
    return {"y": H(y), "dx": H(dx)}


def hip_act(cs):
    y = D(cs["x"])
    act, hw, c = int(cs["act"]), int(cs["hw"]), int(cs["c"])
    sl = D(cs["slopes"])
    ops.activation_forward(y, act, sl, hw, c)
    dx, ds = D(cs["dy"]), D(cs["dslopes0"])
    ops.activation_backward(y, dx, act, sl, ds, hw, c)
    return {"y": H(y), "dx": H(dx), "dslopes": H(ds)}


def hip_dw(cs):
    n, c, h, w, k, s, p, act = (int(cs[q]) for q in ("n", "c", "h", "w", "k", "s", "p", "act"))
    oh, ow = ops.conv_out_hw(h, w, k, s, p)
    x, wt, bias = D(cs["x"]), D(cs["wt"]), D(cs["bias"])
    y = torch.full((n, c, oh, ow), 3.0, device=DEV)
    ops.depthwise_forward(x, wt, bias, y, k, s, p, act)
    out = {"y": H(y)}
    dy, dw, db = D(cs["dy"]), D(cs["dw0"]), D(cs["db0"])
    dx = D(cs["dx0"]) if int(cs["input_grad"]) else None
    ops.depthwise_backward(x, wt, y, dy, dx, dw, db, k, s, p, act)
    out.update(dy_out=H(dy), dw=H(dw), db=H(db))
    if dx is not None:
        out["dx"] = H(dx)
    return out


def hip_im2col(cs):
    c, h, w, k, s, p = (int(cs[q]) for q in ("c", "h", "w", "k", "s", "p"))
    oh, ow = ops.conv_out_hw(h, w, k, s, p)
    col = torch.full((c * k * k, oh * ow), 9.0, device=DEV)
    ops.im2col(D(cs["x"]), k, p, s, col)
    im = torch.full((c, h, w), 9.0, device=DEV)
    ops.col2im(D(cs["col_in"]), k, p, s, im)
    return {"col": H(col), "im": H(im)}


def hip_gemm(cs):
    Cm = D(cs["C0"])
    ops.gemm(int(cs["ta"]), int(cs["tb"]), int(cs["m"]), int(cs["n"]), int(cs["k"]), float(cs["alpha"]),
             D(cs["A"]), cs["A"].shape[1], D(cs["B"]), cs["B"].shape[1], float(cs["beta"]), Cm, int(cs["n"]))
    return {"C": H(Cm)}


def hip_optim(cs):
    """same driver loop as the reference run (oracle/ref_cases.run_optim) on device buffers"""
    n = int(cs["steps"])
    w, b = D(cs["w0"]), D(cs["b0"])
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    adam = str(cs["kind"]) == "adam"
    for t in range(n):
        dw += D(cs["dw_steps"][t])
        db += D(cs["db_steps"][t])
        if adam:
            ops.adam_update(w, b, dw, db, m, v, int(cs["batch"]), (t + 1) * int(cs["batch"]), float(cs["beta1"]),
                            float(cs["beta2"]), float(cs["lr"]), float(cs["momentum"]), float(cs["decay"]))
        else:
            ops.sgd_update(w, b, dw, db, int(cs["batch"]), float(cs["lr"]), float(cs["momentum"]), float(cs["decay"]))
    out = {"w": H(w), "b": H(b), "dw": H(dw), "db": H(db)}
    if adam:
        out.update(adam_m=H(m), adam_v=H(v))
    return out


def run_hip(case, exp=None):
    op = str(case["op"])
    if op == "maxpool":
        return hip_maxpool(case, exp)
    out = {"conv": hip_conv, "bn": hip_bn, "avgpool": hip_avgpool, "act": hip_act, "dw": hip_dw,
           "im2col": hip_im2col, "gemm": hip_gemm, "optim": hip_optim}[op](case)
    torch.cuda.synchronize()
    return out
