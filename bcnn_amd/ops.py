"""Host-side mirror of the reference's per-layer workers on torch device tensors.

Each function takes/returns torch CUDA(=HIP) float32/int32 tensors and forwards to the C-ABI in
lib/libbcnn_hip.so with raw device pointers; argument meaning follows the reference layer code
(file:line cited per function). torch is plumbing only (device memory + streams); no torch op computes
anything on this path, and a missing extension raises (bcnn_amd/_lib.py).
"""
import torch

from . import _lib

ACT = dict(none=0, tanh=1, relu=2, ramp=3, softplus=4, lrelu=5, abs=6, clamp=7, prelu=8, logistic=9)
MODE_PREDICT, MODE_TRAIN, MODE_VALID = 0, 1, 2


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
    return t.data_ptr()


def _f32(t):
    assert t is None or t.dtype == torch.float32
    return _p(t)


def conv_out_hw(h, w, k, s, p):
    """bcnn_conv_layer.c:126-134"""
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


def conv_workspace_size(n, c, h, w, f, k, s, p, g):
    return int(_lib.load().bcnn_hip_conv_workspace_size(n, c, h, w, f, k, s, p, g))


def conv_forward(x, wt, bias, y, k, stride, pad, groups=1, act=0, slopes=None, bn=None, mode=MODE_TRAIN):
    """bcnn_forward_conv_layer (bcnn_conv_layer.c:367-485). bn: dict(run_mean, run_var, scales,
    saved_mean, saved_var, x_norm (optional), workspace) for a fused batch-norm node, else None."""
    n, c, h, w = x.shape
    f = wt.shape[0]
    L = _lib.load()
    b = bn or {}
    L.bcnn_hip_conv_forward(_f32(x), _f32(wt), _f32(bias), _f32(y), n, c, h, w, f, k, stride, pad, groups,
                            act, _f32(slopes), 1 if bn else 0, _f32(b.get("run_mean")), _f32(b.get("run_var")),
                            _f32(b.get("scales")), _f32(b.get("saved_mean")), _f32(b.get("saved_var")),
                            _f32(b.get("x_norm")), _f32(b.get("workspace")), mode)


def conv_backward(x, wt, y, dy, dx, dw, dbias, k, stride, pad, groups, act, workspace, slopes=None,
                  dslopes=None, bn=None, bias=None):
    """bcnn_backward_conv_layer (bcnn_conv_layer.c:487-587). dy is updated in place; dx may be None.
    bias (optional): the forward bias; lets the fused batch-norm backward recompute y instead of reading it."""
    n, c, h, w = x.shape
    f = wt.shape[0]
    L = _lib.load()
    b = bn or {}
    L.bcnn_hip_conv_backward(_f32(x), _f32(wt), _f32(bias), _f32(y), _f32(dy), _f32(dx), _f32(dw), _f32(dbias), n, c, h, w,
                             f, k, stride, pad, groups, act, _f32(slopes), _f32(dslopes), 1 if bn else 0,
                             _f32(b.get("scales")), _f32(b.get("dscales")), _f32(b.get("saved_mean")),
                             _f32(b.get("saved_var")), _f32(b.get("dmean")), _f32(b.get("dvar")),
                             _f32(b.get("x_norm")), _f32(b.get("workspace")), _f32(workspace),
                             workspace.numel() if workspace is not None else 0)


def batchnorm_forward(x, y, run_mean, run_var, scales, bias, saved_mean, saved_var, workspace, mode,
                      x_norm=None, act=0):
    """bcnn_forward_batchnorm_cpu (bcnn_batchnorm_layer.c:196-242)"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_batchnorm_forward(_f32(x), _f32(y), _f32(run_mean), _f32(run_var), _f32(scales),
                                           _f32(bias), _f32(saved_mean), _f32(saved_var), _f32(x_norm),
                                           _f32(workspace), n, c, h * w, mode, act)


def batchnorm_backward(dy, dx, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar, workspace,
                       y=None, act=0, x_norm=None):
    """bcnn_backward_batchnorm_cpu (bcnn_batchnorm_layer.c:301-332)"""
    n, c, h, w = dy.shape
    _lib.load().bcnn_hip_batchnorm_backward(_f32(dy), _f32(dx), _f32(y), act, _f32(scales), _f32(dscales),
                                            _f32(dbias), _f32(saved_mean), _f32(saved_var), _f32(dmean),
                                            _f32(dvar), _f32(x_norm), _f32(workspace), n, c, h * w)


def maxpool_forward(x, y, indexes, size, stride):
    """bcnn_forward_maxpool_layer_cpu (bcnn_maxpool_layer.c:145-191)"""
    n, c, h, w = x.shape
    assert indexes.dtype == torch.int32
    _lib.load().bcnn_hip_maxpool_forward(_f32(x), _f32(y), _p(indexes), n, c, h, w, y.shape[2], y.shape[3],
                                         size, stride)


def maxpool_backward(dy, indexes, dx, size, stride, overwrite=False):
    """bcnn_backward_maxpool_layer_cpu (bcnn_maxpool_layer.c:258-273). overwrite: dx := 0 + sums (the caller skipped
    the zero fill) instead of dx += sums."""
    n, c, h, w = dx.shape
    _lib.load().bcnn_hip_maxpool_backward(_f32(dy), _p(indexes), _f32(dx), n, c, h, w, dy.shape[2], dy.shape[3],
                                          size, stride, 1 if overwrite else 0)


def avgpool_forward(x, y):
    """bcnn_forward_avgpool_layer_cpu (bcnn_avgpool_layer.c:82-99)"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_avgpool_forward(_f32(x), _f32(y), n, c, h, w)


def avgpool_backward(dy, dx):
    """bcnn_backward_avgpool_layer_cpu (bcnn_avgpool_layer.c:109-125)"""
    n, c, h, w = dx.shape
    _lib.load().bcnn_hip_avgpool_backward(_f32(dy), _f32(dx), n, c, h, w)


def activation_forward(x, act, slopes=None, spatial=1, channels=1):
    """bcnn_forward_activation_cpu (bcnn_activation_layer.c:90-146), in place"""
    _lib.load().bcnn_hip_activation_forward(_f32(x), x.numel(), act, _f32(slopes), spatial, channels)


def activation_backward(x, dx, act, slopes=None, dslopes=None, spatial=1, channels=1):
    """bcnn_backward_activation_cpu (bcnn_activation_layer.c:165-226), dx in place"""
    _lib.load().bcnn_hip_activation_backward(_f32(x), _f32(dx), x.numel(), act, _f32(slopes), _f32(dslopes),
                                             spatial, channels)


def depthwise_forward(x, wt, bias, y, k, stride, pad, act=0):
    """bcnn_forward_depthwise_conv_layer_cpu (bcnn_depthwise_conv_layer.c:165-293)"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_depthwise_forward(_f32(x), _f32(wt), _f32(bias), _f32(y), n, c, h, w, k, stride, pad, act)


def depthwise_backward(x, wt, y, dy, dx, dw, dbias, k, stride, pad, act=0, overwrite=False):
    """bcnn_backward_depthwise_conv_layer_cpu (bcnn_depthwise_conv_layer.c:295-547); overwrite: dx = 0 + sums
    (the executor's no-fill mode for a sole gradient writer) instead of dx += sums"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_depthwise_backward(_f32(x), _f32(wt), _f32(y), _f32(dy), _f32(dx), _f32(dw),
                                            _f32(dbias), n, c, h, w, k, stride, pad, act, 1 if overwrite else 0)


def batchnorm_apply(x, y, scales, bias, mean, var, act=0):
    """y = act((x - mean) / sqrtf(var + 1e-6) * scale + bias) with given statistics (bcnn_batchnorm_layer.c:226-241)"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_batchnorm_apply(_f32(x), _f32(y), _f32(scales), _f32(bias), _f32(mean), _f32(var), n, c, h * w, act)


def maxpool_bn_fusable(x, out_h, out_w, size, stride, act):
    n, c, h, w = x.shape
    return bool(_lib.load().bcnn_hip_maxpool_bn_fusable(n, c, h, w, out_h, out_w, size, stride, act, _f32(x)))


def maxpool_forward_bn(x, y, indexes, size, stride, scales, bias, mean, var, act):
    """max-pooling over act(batch-norm(x)) normalised on the fly: bcnn_forward_maxpool_layer_cpu
    (bcnn_maxpool_layer.c:145-191) on the tensor bcnn_batchnorm_layer.c:226-241 would have written"""
    n, c, h, w = x.shape
    assert indexes.dtype == torch.int32
    _lib.load().bcnn_hip_maxpool_forward_bn(_f32(x), _f32(y), _p(indexes), n, c, h, w, y.shape[2], y.shape[3], size, stride,
                                            _f32(scales), _f32(bias), _f32(mean), _f32(var), act)


def depthwise_forward_stats(x, wt, bias, y, k, stride, pad, act, stats):
    """bcnn_hip_depthwise_forward that also leaves per-channel (sum, sum of squares) partials of y in `stats`;
    returns the number of partials per channel (0: none, run the plain batch-norm forward)"""
    n, c, h, w = x.shape
    return _lib.load().bcnn_hip_depthwise_forward_stats(_f32(x), _f32(wt), _f32(bias), _f32(y), n, c, h, w, k, stride,
                                                        pad, act, _f32(stats), stats.numel())


def depthwise_stats_size(n, c, h, w, k, stride, pad):
    return _lib.load().bcnn_hip_depthwise_stats_size(n, c, h, w, k, stride, pad)


def batchnorm_forward_stats(x, y, run_mean, run_var, scales, bias, saved_mean, saved_var, workspace, mode, stats, splits):
    """bcnn_forward_batchnorm_cpu (bcnn_batchnorm_layer.c:196-242) with the statistics partials of the producer"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_batchnorm_forward_stats(_f32(x), _f32(y), _f32(run_mean), _f32(run_var), _f32(scales), _f32(bias),
                                                 _f32(saved_mean), _f32(saved_var), None, _f32(workspace), n, c, h * w,
                                                 mode, 0, _f32(stats), splits)


def depthwise_bn_fusable(n, c, h, w, k, stride, pad, act):
    return bool(_lib.load().bcnn_hip_depthwise_bn_fusable(n, c, h, w, k, stride, pad, act))


def batchnorm_backward_sums(dy, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar, x):
    """first sweep of bcnn_backward_batchnorm_cpu (bcnn_batchnorm_layer.c:301-332): the per-channel sums only"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_batchnorm_backward_sums(_f32(dy), _f32(scales), _f32(dscales), _f32(dbias), _f32(saved_mean),
                                                 _f32(saved_var), _f32(dmean), _f32(dvar), _f32(x), n, c, h * w)


def depthwise_backward_bn(x, wt, y, dz, dx, dw, dbias, k, stride, pad, act, overwrite, mean, var, scales, dmean, dvar):
    """depthwise backward on the gradient a following stand-alone batch-norm node would hand down, applied on the fly"""
    n, c, h, w = x.shape
    _lib.load().bcnn_hip_depthwise_backward_bn(_f32(x), _f32(wt), _f32(y), _f32(dz), _f32(dx), _f32(dw), _f32(dbias), n, c,
                                               h, w, k, stride, pad, act, 1 if overwrite else 0, _f32(mean), _f32(var),
                                               _f32(scales), _f32(dmean), _f32(dvar))


def gemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """bcnn_gemm (bcnn_mat.c:2627-2650)"""
    _lib.load().bcnn_hip_gemm(ta, tb, m, n, k, alpha, _f32(a), lda, _f32(b), ldb, beta, _f32(c), ldc)


def im2col(im, k, pad, stride, col):
    c, h, w = im.shape
    _lib.load().bcnn_hip_im2col(_f32(im), c, h, w, k, pad, stride, _f32(col))


def col2im(col, k, pad, stride, im):
    c, h, w = im.shape
    _lib.load().bcnn_hip_col2im(_f32(col), c, h, w, k, pad, stride, _f32(im))


def adam_update(w, b, dw, db, adam_m, adam_v, batch_size, it, beta1, beta2, lr, momentum, decay):
    """bcnn_adam_update_cpu (bcnn_learner.c:106-131)"""
    _lib.load().bcnn_hip_adam_update(_f32(w), _f32(b), _f32(dw), _f32(db), _f32(adam_m), _f32(adam_v),
                                     w.numel() if w is not None else 0, b.numel() if b is not None else 0,
                                     batch_size, it, beta1, beta2, lr, momentum, decay)


def sgd_update(w, b, dw, db, batch_size, lr, momentum, decay):
    """bcnn_sgd_update_cpu (bcnn_learner.c:67-83)"""
    _lib.load().bcnn_hip_sgd_update(_f32(w), _f32(b), _f32(dw), _f32(db), w.numel() if w is not None else 0,
                                    b.numel() if b is not None else 0, batch_size, lr, momentum, decay)
