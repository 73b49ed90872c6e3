/*
 * oracle/bcnn_oracle.c -- plain-C restatement of the reference's conv/GEMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see bcnn_oracle.h). Parity status: PINNED -- every function here is
 * checked in tests/test_oracle.py against tests/golden/<case>.npz, which are outputs of the unmodified
 * reference (AVX2 + OpenMP build, in-tree bcnn_gemm) produced by tests/golden/make_golden.py.
 *
 * The restatement keeps the reference's summation ORDER (4-lane partial sums of the SSE helpers,
 * KC=384 GEMM panels, separate multiply and add -- the reference build has no FMA) so that it is
 * bit-comparable with the reference, which makes it a sharp checker for the GPU path.
 * Build with -ffp-contract=off (oracle/Makefile).
 *
 * Citations: file:line in the reference tree (jnbraun/bcnn).
 */
#include "bcnn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* BLAS-1 helpers with the reference's lane order                                              */
/* ------------------------------------------------------------------------------------------ */

/* bcnn_vsum, src/kernels/bcnn_mat.c:447-475: four running lanes fed 8 floats per step. */
static float lane_sum(const float *x, int n) {
    float l[4] = {0.f, 0.f, 0.f, 0.f};
    int nd = n / 8 * 8;
    for (int i = 0; i < nd; i += 8) {
        for (int j = 0; j < 4; ++j) l[j] = l[j] + x[i + j];
        for (int j = 0; j < 4; ++j) l[j] = l[j] + x[i + 4 + j];
    }
    float s = 0.f;
    s += l[0] + l[1] + l[2] + l[3];
    for (int i = nd; i < n; ++i) s += x[i];
    return s;
}

/* bcnn_shiftdot, src/kernels/bcnn_mat.c:654-690; bcnn_dot (:417-445) is the a = b = 0 case
 * without the subtractions (x - 0 == x exactly, so one routine serves both). */
static float lane_shiftdot(const float *x, float a, const float *y, float b, int n) {
    float l[4] = {0.f, 0.f, 0.f, 0.f};
    int nd = n / 8 * 8;
    for (int i = 0; i < nd; i += 8) {
        for (int j = 0; j < 4; ++j) {
            float p = (x[i + j] - a) * (y[i + j] - b);
            l[j] = l[j] + p;
        }
        for (int j = 0; j < 4; ++j) {
            float p = (x[i + 4 + j] - a) * (y[i + 4 + j] - b);
            l[j] = l[j] + p;
        }
    }
    float s = 0.f;
    s += l[0] + l[1] + l[2] + l[3];
    for (int i = nd; i < n; ++i) s += (x[i] - a) * (y[i] - b);
    return s;
}

/* bcnn_scal, src/kernels/bcnn_mat.c:319-364: a == 0 -> memset, a == 1 -> untouched. */
static void scal(int n, float a, float *x) {
    if (a == 0.0f) {
        memset(x, 0, (size_t)n * sizeof(float));
    } else if (a != 1.0f) {
        for (int i = 0; i < n; ++i) x[i] *= a;
    }
}

/* bcnn_add_scalar (AVX build), src/kernels/bcnn_mat.c:366-412: a == 0 returns, and a == 1.0f
 * falls through BOTH branches, i.e. nothing is added. */
static void add_scalar(int n, float a, float *x) {
    if (a == 0.0f) return;
    if (a != 1.0f) {
        for (int i = 0; i < n; ++i) x[i] += a;
    }
}

/* bcnn_axpy, src/kernels/bcnn_mat.c:52-115: y = a*x + y, multiply then add. */
static void axpy(int n, float a, const float *x, float *y) {
    for (int i = 0; i < n; ++i) {
        float p = x[i] * a;
        y[i] = p + y[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* im2col / col2im                                                                             */
/* ------------------------------------------------------------------------------------------ */

void orc_im2col(const float *im, int c, int h, int w, int k, int pad, int stride, float *col) {
    const int oh = (h + 2 * pad - k) / stride + 1;
    const int ow = (w + 2 * pad - k) / stride + 1;
#pragma omp parallel for
    for (int row = 0; row < c * k * k; ++row) {
        const int ch = row / (k * k), kr = (row / k) % k, kc = row % k;
        float *dst = col + (size_t)row * oh * ow;
        for (int y = 0; y < oh; ++y) {
            const int iy = y * stride - pad + kr;
            for (int x = 0; x < ow; ++x) {
                const int ix = x * stride - pad + kc;
                const int in = (iy >= 0 && iy < h && ix >= 0 && ix < w);
                dst[y * ow + x] = in ? im[((size_t)ch * h + iy) * w + ix] : 0.f;
            }
        }
    }
}

/* zero-fills `im`, then scatter-adds in (channel, kr, kc, oh, ow) order: bcnn_mat.c:935-970 */
void orc_col2im(const float *col, int c, int h, int w, int k, int pad, int stride, float *im) {
    const int oh = (h + 2 * pad - k) / stride + 1;
    const int ow = (w + 2 * pad - k) / stride + 1;
    memset(im, 0, (size_t)c * h * w * sizeof(float));
#pragma omp parallel for
    for (int ch = 0; ch < c; ++ch) {
        float *plane = im + (size_t)ch * h * w;
        for (int kr = 0; kr < k; ++kr)
            for (int kc = 0; kc < k; ++kc) {
                const float *src = col + (size_t)((ch * k + kr) * k + kc) * oh * ow;
                for (int y = 0; y < oh; ++y) {
                    const int iy = y * stride - pad + kr;
                    if (iy < 0 || iy >= h) continue;
                    for (int x = 0; x < ow; ++x) {
                        const int ix = x * stride - pad + kc;
                        if (ix >= 0 && ix < w) plane[iy * w + ix] += src[y * ow + x];
                    }
                }
            }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* GEMM                                                                                        */
/* ------------------------------------------------------------------------------------------ */

static int near(float a, float b) { return fabsf(a - b) < 1e-5f; } /* `equal`, bcnn_mat.c:2156 */

/* Row-major sgemm with the reference's blocking visible in the numerics: the K dimension is cut
 * into KC = 384 panels (bcnn_mat.h:84-88); inside a panel each C element is a k-ascending chain
 * of separately rounded multiplies and adds starting from 0 (sgemm_ukernel, bcnn_mat.c:2311-2352),
 * then C = beta_panel*C + alpha*acc with beta_panel = beta for the first panel and 1 afterwards
 * (sgemm_nn, bcnn_mat.c:2527-2560). */
void orc_gemm(int ta, int tb, int m, int n, int k, float alpha, const float *A, int lda,
              const float *B, int ldb, float beta, float *C, int ldc) {
    const int KC = 384;
    const size_t ars = ta ? 1 : (size_t)lda, acs = ta ? (size_t)lda : 1;
    const size_t brs = tb ? 1 : (size_t)ldb, bcs = tb ? (size_t)ldb : 1;
    if (near(alpha, 0.f) || k == 0) {
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) {
                if (near(beta, 0.f)) C[(size_t)i * ldc + j] = 0.f;
                else C[(size_t)i * ldc + j] *= beta;
            }
        return;
    }
    for (int l0 = 0; l0 < k; l0 += KC) {
        const int kc = (k - l0 < KC) ? (k - l0) : KC;
        const float bp = (l0 == 0) ? beta : 1.0f;
#pragma omp parallel for
        for (int i = 0; i < m; ++i) {
            float *acc = (float *)malloc((size_t)n * sizeof(float));
            for (int j = 0; j < n; ++j) acc[j] = 0.f;
            for (int l = 0; l < kc; ++l) {
                const float a = A[(size_t)i * ars + (size_t)(l0 + l) * acs];
                const float *brow = B + (size_t)(l0 + l) * brs;
                for (int j = 0; j < n; ++j) {
                    float p = brow[(size_t)j * bcs] * a;
                    acc[j] = acc[j] + p;
                }
            }
            float *crow = C + (size_t)i * ldc;
            for (int j = 0; j < n; ++j) {
                float cv = crow[j];
                if (near(bp, 0.f)) cv = 0.f;
                else if (!near(bp, 1.f)) cv *= bp;
                if (!near(alpha, 1.f)) cv += alpha * acc[j];
                else cv += acc[j];
                crow[j] = cv;
            }
            free(acc);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* per-channel helpers                                                                         */
/* ------------------------------------------------------------------------------------------ */

void orc_add_bias(float *y, const float *bias, int n, int c, int hw) {
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < c; ++i) add_scalar(hw, bias[i], y + ((size_t)b * c + i) * hw);
}

void orc_scales(float *y, const float *scales, int n, int c, int hw) {
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < c; ++i) scal(hw, scales[i], y + ((size_t)b * c + i) * hw);
}

/* sequential `+=` straight into gb[i], images outermost: bcnn_mat.c:798-811 */
void orc_grad_bias(float *gb, const float *g, int n, int c, int hw) {
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < c; ++i) {
            const float *p = g + ((size_t)b * c + i) * hw;
            for (int j = 0; j < hw; ++j) gb[i] += p[j];
        }
}

/* local sequential sum per channel, then one `+=`: bcnn_mat.c:783-796 */
void orc_grad_scales(const float *x_norm, const float *g, int n, int c, int hw, float *gs) {
    for (int f = 0; f < c; ++f) {
        float sum = 0.f;
        for (int b = 0; b < n; ++b)
            for (int i = 0; i < hw; ++i) {
                size_t idx = (size_t)i + (size_t)hw * (f + (size_t)c * b);
                sum += g[idx] * x_norm[idx];
            }
        gs[f] += sum;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* activation map                                                                              */
/* ------------------------------------------------------------------------------------------ */

void orc_act_forward(float *x, int sz, const float *slope, int hw, int c, int act) {
    switch (act) {
        case ORC_ACT_TANH:
            for (int i = 0; i < sz; ++i)
                x[i] = (float)(exp(2 * x[i]) - 1) / ((float)exp(2 * x[i]) + 1);
            break;
        case ORC_ACT_RELU: /* a multiply: negatives give -0.0f, NaN/Inf propagate */
            for (int i = 0; i < sz; ++i) x[i] = x[i] * (x[i] > 0);
            break;
        case ORC_ACT_LRELU: /* slope 0.1 (the header comment says 0.01) */
            for (int i = 0; i < sz; ++i) x[i] = (x[i] > 0 ? x[i] : 0.1f * x[i]);
            break;
        case ORC_ACT_RAMP:
            for (int i = 0; i < sz; ++i) x[i] = x[i] * (x[i] > 0) + 0.1f * x[i];
            break;
        case ORC_ACT_SOFTPLUS:
            for (int i = 0; i < sz; ++i) x[i] = (float)log(1.0f + (float)exp(x[i]));
            break;
        case ORC_ACT_ABS:
            for (int i = 0; i < sz; ++i) x[i] = (float)fabs(x[i]);
            break;
        case ORC_ACT_CLAMP:
            for (int i = 0; i < sz; ++i) x[i] = (x[i] < 0) ? 0 : ((x[i] > 1) ? 1 : x[i]);
            break;
        case ORC_ACT_LOGISTIC:
            for (int i = 0; i < sz; ++i) x[i] = 1.0f / (1.0f + (float)exp(-x[i]));
            break;
        case ORC_ACT_PRELU:
            for (int i = 0; i < sz; ++i) {
                int ch = (i / hw) % c;
                x[i] = (x[i] > 0 ? x[i] : slope[ch] * x[i]);
            }
            break;
        default:
            break;
    }
}

/* `x` is the POST-activation value. */
void orc_act_backward(const float *x, float *dx, int sz, const float *slope, float *dslope, int hw,
                      int c, int act) {
    switch (act) {
        case ORC_ACT_TANH:
            for (int i = 0; i < sz; ++i) dx[i] *= (1 - x[i] * x[i]);
            break;
        case ORC_ACT_RELU:
            for (int i = 0; i < sz; ++i) dx[i] *= ((float)(x[i] > 0));
            break;
        case ORC_ACT_LRELU:
            for (int i = 0; i < sz; ++i) dx[i] *= (x[i] > 0 ? 1.0f : 0.1f);
            break;
        case ORC_ACT_RAMP:
            for (int i = 0; i < sz; ++i) dx[i] *= ((float)(x[i] > 0) + 0.1f);
            break;
        case ORC_ACT_SOFTPLUS:
            for (int i = 0; i < sz; ++i) dx[i] *= 1.0f / (1.0f + (float)exp(-x[i]));
            break;
        case ORC_ACT_ABS:
            for (int i = 0; i < sz; ++i) dx[i] *= (x[i] >= 0 ? 1.0f : -1.0f);
            break;
        case ORC_ACT_CLAMP:
            for (int i = 0; i < sz; ++i) dx[i] *= ((float)(x[i] > 0.0f && x[i] < 1.0f));
            break;
        case ORC_ACT_LOGISTIC:
            for (int i = 0; i < sz; ++i) dx[i] *= (1 - x[i]) * x[i];
            break;
        case ORC_ACT_PRELU:
            for (int i = 0; i < sz; ++i) {
                int ch = (i / hw) % c;
                dslope[ch] += dx[i] * x[i] * (x[i] < 0);
            }
            for (int i = 0; i < sz; ++i) {
                int ch = (i / hw) % c;
                dx[i] *= (x[i] > 0 ? 1.0f : slope[ch]);
            }
            break;
        default:
            break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* batch normalisation                                                                         */
/* ------------------------------------------------------------------------------------------ */

void orc_bn_forward(const float *x, float *y, float *run_mean, float *run_var, const float *scales,
                    const float *bias, float *saved_mean, float *saved_var, float *x_norm,
                    float *workspace, int n, int c, int hw, int mode) {
    const size_t total = (size_t)n * c * hw;
    if (x != y) memcpy(y, x, total * sizeof(float));
    if (mode == ORC_MODE_PREDICT) { /* scale_and_add_bias, bcnn_batchnorm_layer.c:183-194 */
        for (int b = 0; b < n; ++b)
            for (int j = 0; j < c; ++j) {
                float *p = y + ((size_t)b * c + j) * hw;
                for (int i = 0; i < hw; ++i) p[i] = p[i] * scales[j] + bias[j];
            }
        return;
    }
    memcpy(workspace, y, total * sizeof(float));
    const float *mean = run_mean, *var = run_var;
    if (mode == ORC_MODE_TRAIN) {
        /* _mean_variance_forward, :147-168: biased one-pass variance E[x^2] - E[x]^2 */
        const float inv = 1.0f / (n * hw);
        for (int i = 0; i < c; ++i) {
            float m = 0.f, v = 0.f;
            for (int b = 0; b < n; ++b) {
                const float *p = y + ((size_t)b * c + i) * hw;
                m += lane_sum(p, hw);
                v += lane_shiftdot(p, 0.f, p, 0.f, hw);
            }
            saved_mean[i] = m;
            saved_var[i] = v;
        }
        scal(c, inv, saved_mean);
        for (int i = 0; i < c; ++i) { /* bcnn_varmean, bcnn_mat.c:729-759 */
            float a = saved_var[i] * inv, b2 = saved_mean[i] * saved_mean[i];
            saved_var[i] = a - b2;
        }
        scal(c, 0.9f, run_mean); axpy(c, 0.1f, saved_mean, run_mean); /* :220-223 */
        scal(c, 0.9f, run_var);  axpy(c, 0.1f, saved_var, run_var);
        mean = saved_mean; var = saved_var;
    }
    /* _norm_forward, :170-181: eps = 1e-6 */
    for (int b = 0; b < n; ++b)
        for (int j = 0; j < c; ++j) {
            float *p = y + ((size_t)b * c + j) * hw;
            for (int i = 0; i < hw; ++i) p[i] = (p[i] - mean[j]) / (sqrtf(var[j] + 0.000001f));
        }
    if (mode == ORC_MODE_TRAIN && x_norm) memcpy(x_norm, y, total * sizeof(float));
    orc_scales(y, scales, n, c, hw);
    orc_add_bias(y, bias, n, c, hw);
}

void orc_bn_backward(float *dy, float *dx, const float *scales, float *dscales, float *dbias,
                     const float *mean, const float *var, float *dmean, float *dvar,
                     const float *x_norm, const float *workspace, int n, int c, int hw) {
    orc_grad_bias(dbias, dy, n, c, hw);
    orc_grad_scales(x_norm, dy, n, c, hw, dscales);
    orc_scales(dy, scales, n, c, hw);
    /* _mean_variance_backward, :263-281: eps = 1e-5 here (1e-6 in forward) */
    for (int i = 0; i < c; ++i) {
        float md = 0.f, vd = 0.f;
        for (int b = 0; b < n; ++b) {
            size_t off = ((size_t)b * c + i) * hw;
            md += lane_sum(dy + off, hw);
            vd += lane_shiftdot(workspace + off, mean[i], dy + off, 0.0f, hw);
        }
        md *= (-1.0f / sqrtf(var[i] + 0.00001f));
        dmean[i] = md;
        dvar[i] = vd;
    }
    for (int i = 0; i < c; ++i) /* bcnn_varnorm(c, var, -0.5f, var_diff), bcnn_mat.c:692-727 */
        dvar[i] *= -0.5f / (var[i] * sqrtf(var[i]) + 0.00001f);
    /* _normalize_backward, :283-299 */
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < c; ++i) {
            size_t off = ((size_t)b * c + i) * hw;
            for (int k = 0; k < hw; ++k) {
                size_t ind = off + k;
                dy[ind] = dy[ind] * 1.0f / (sqrtf(var[i] + 0.00001f)) +
                          dvar[i] * 2.0f * (workspace[ind] - mean[i]) / (hw * n) +
                          dmean[i] / (hw * n);
            }
        }
    if (dx && dx != dy) memcpy(dx, dy, (size_t)n * c * hw * sizeof(float));
}

/* ------------------------------------------------------------------------------------------ */
/* convolution                                                                                 */
/* ------------------------------------------------------------------------------------------ */

void orc_conv_forward(const float *x, const float *wt, const float *bias, float *y, int n, int c,
                      int h, int w, int f, int k, int stride, int pad, int groups, int act,
                      const float *slopes, int bn, float *run_mean, float *run_var,
                      const float *scales, float *saved_mean, float *saved_var, float *x_norm,
                      float *bn_ws, int mode, float *col_ws) {
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    const int m = f / groups, kk = k * k * c / groups, nn = oh * ow;
    const size_t img = (size_t)c * h * w, wsz = (size_t)f * (c / groups) * k * k;
    memset(y, 0, (size_t)n * f * nn * sizeof(float));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < groups; ++j) {
            const float *a = wt + (size_t)j * wsz / groups;
            float *cm = y + ((size_t)i * groups + j) * nn * m;
            const float *src = x + ((size_t)i * groups + j) * img / groups;
            const float *b = src; /* 1x1: the raw buffer IS the K x N matrix (quirk 1, :445-446) */
            if (k != 1) {
                orc_im2col(src, c / groups, h, w, k, pad, stride, col_ws);
                b = col_ws;
            }
            orc_gemm(0, 0, m, nn, kk, 1.0f, a, kk, b, nn, 1.0f, cm, nn);
        }
    if (bn)
        orc_bn_forward(y, y, run_mean, run_var, scales, bias, saved_mean, saved_var, x_norm, bn_ws,
                       n, f, nn, mode);
    else
        orc_add_bias(y, bias, n, f, nn);
    orc_act_forward(y, n * f * nn, slopes, nn, f, act);
}

void orc_conv_backward(const float *x, const float *wt, const float *y, float *dy, float *dx,
                       float *dwt, float *dbias, int n, int c, int h, int w, int f, int k,
                       int stride, int pad, int groups, int act, const float *slopes,
                       float *dslopes, int bn, const float *scales, float *dscales,
                       const float *saved_mean, const float *saved_var, float *dmean, float *dvar,
                       const float *x_norm, const float *bn_ws, float *col_ws) {
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    const int m = f / groups, nn = k * k * c / groups, kk = oh * ow;
    const size_t img = (size_t)c * h * w, wsz = (size_t)f * (c / groups) * k * k;
    orc_act_backward(y, dy, n * f * kk, slopes, dslopes, kk, f, act);
    if (bn)
        orc_bn_backward(dy, NULL, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar,
                        x_norm, bn_ws, n, f, kk);
    else
        orc_grad_bias(dbias, dy, n, f, kk);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < groups; ++j) {
            float *a = dy + ((size_t)i * groups + j) * m * kk;
            const float *src = x + ((size_t)i * groups + j) * img / groups;
            const float *b = src;
            if (k != 1) {
                orc_im2col(src, c / groups, h, w, k, pad, stride, col_ws);
                b = col_ws;
            }
            /* dW_g += G_i * col^T  (beta = 1: on top of whatever the buffer holds, :547-553) */
            orc_gemm(0, 1, m, nn, kk, 1.0f, a, kk, b, kk, 1.0f, dwt + (size_t)j * wsz / groups, nn);
            if (dx) {
                const float *wa = wt + (size_t)j * wsz / groups;
                float *sg = dx + ((size_t)i * groups + j) * img / groups;
                if (k == 1) { /* beta = 0 straight into src.grad viewed as [K][OH*OW] (:562-569) */
                    orc_gemm(1, 0, nn, kk, m, 1.0f, wa, nn, a, kk, 0.0f, sg, kk);
                } else { /* col2im zero-fills => OVERWRITES src.grad (:571-581) */
                    orc_gemm(1, 0, nn, kk, m, 1.0f, wa, nn, a, kk, 0.0f, col_ws, kk);
                    orc_col2im(col_ws, c / groups, h, w, k, pad, stride, sg);
                }
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* pooling                                                                                     */
/* ------------------------------------------------------------------------------------------ */

void orc_maxpool_forward(const float *x, float *y, int *indexes, int n, int c, int h, int w, int oh,
                         int ow, int k, int stride) {
#pragma omp parallel for collapse(2)
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch)
            for (int i = 0; i < oh; ++i)
                for (int j = 0; j < ow; ++j) {
                    float best = -FLT_MAX;
                    int bi = -1;
                    for (int r = 0; r < k; ++r)       /* rows outer, cols inner */
                        for (int q = 0; q < k; ++q) { /* window starts at (i*s, j*s): no top/left pad */
                            int yy = i * stride + r, xx = j * stride + q;
                            int si = xx + w * (yy + h * (ch + b * c));
                            int ok = (yy >= 0 && yy < h && xx >= 0 && xx < w);
                            float v = ok ? x[si] : -FLT_MAX;
                            if (v > best) { best = v; bi = si; } /* first strict max; NaN never wins */
                        }
                    int di = j + ow * (i + oh * (ch + b * c));
                    y[di] = best;
                    indexes[di] = bi;
                }
}

void orc_maxpool_backward(const float *dy, const int *indexes, float *dx, int out_size) {
    for (int i = 0; i < out_size; ++i) dx[indexes[i]] += dy[i];
}

void orc_avgpool_forward(const float *x, float *y, int n, int c, int h, int w) {
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch) {
            int idx = ch + b * c;
            y[idx] = 0;
            for (int i = 0; i < h * w; ++i) y[idx] += x[(size_t)h * w * idx + i];
            y[idx] /= h * w;
        }
}

void orc_avgpool_backward(const float *dy, float *dx, int n, int c, int h, int w) {
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch) {
            int idx = ch + b * c;
            for (int i = 0; i < h * w; ++i) dx[(size_t)h * w * idx + i] += dy[idx] / (h * w);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* depthwise convolution                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* The four border/interior branches of the reference (:190-283) compute the same zero-padded sum
 * in the same (kh outer, kw inner) order; out-of-image taps are skipped, not added as zeros. */
void orc_dw_forward(const float *x, const float *wt, const float *bias, float *y, int n, int c,
                    int h, int w, int k, int stride, int pad, int act) {
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch)
            for (int i = 0; i < oh; ++i)
                for (int j = 0; j < ow; ++j) {
                    float val = 0;
                    for (int kh = 0; kh < k; ++kh)
                        for (int kw = 0; kw < k; ++kw) {
                            int yy = -pad + i * stride + kh, xx = -pad + j * stride + kw;
                            if (yy >= 0 && yy < h && xx >= 0 && xx < w)
                                val += wt[(ch * k + kh) * k + kw] *
                                       x[(((size_t)b * c + ch) * h + yy) * w + xx];
                        }
                    y[(((size_t)b * c + ch) * oh + i) * ow + j] = val;
                }
    orc_add_bias(y, bias, n, c, oh * ow);
    orc_act_forward(y, n * c * oh * ow, NULL, oh * ow, c, act);
}

void orc_dw_backward(const float *x, const float *wt, const float *y, float *dy, float *dx,
                     float *dwt, float *dbias, int n, int c, int h, int w, int k, int stride,
                     int pad, int act) {
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    orc_act_backward(y, dy, n * c * oh * ow, NULL, NULL, oh * ow, c, act);
    orc_grad_bias(dbias, dy, n, c, oh * ow);
    if (!dx) return; /* both dW and dX are skipped when src has no gradient (:318, :432) */
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch)
            for (int i = 0; i < oh; ++i)
                for (int j = 0; j < ow; ++j) {
                    float g = dy[(((size_t)b * c + ch) * oh + i) * ow + j];
                    for (int kh = 0; kh < k; ++kh)
                        for (int kw = 0; kw < k; ++kw) {
                            int yy = -pad + i * stride + kh, xx = -pad + j * stride + kw;
                            if (yy >= 0 && yy < h && xx >= 0 && xx < w)
                                dwt[(ch * k + kh) * k + kw] +=
                                    x[(((size_t)b * c + ch) * h + yy) * w + xx] * g;
                        }
                }
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch)
            for (int i = 0; i < oh; ++i)
                for (int j = 0; j < ow; ++j) {
                    float g = dy[(((size_t)b * c + ch) * oh + i) * ow + j];
                    for (int kh = 0; kh < k; ++kh)
                        for (int kw = 0; kw < k; ++kw) {
                            int yy = -pad + i * stride + kh, xx = -pad + j * stride + kw;
                            if (yy >= 0 && yy < h && xx >= 0 && xx < w)
                                dx[(((size_t)b * c + ch) * h + yy) * w + xx] +=
                                    wt[(ch * k + kh) * k + kw] * g; /* accumulates, no zeroing */
                        }
                }
}

/* ------------------------------------------------------------------------------------------ */
/* SGD step ("next" row f-1)                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* momentum lives inside the gradient buffers: bcnn_learner.c:67-83 */
void orc_sgd_update(float *weights, float *biases, float *dweights, float *dbiases, int wsize,
                    int bsize, int batch, float lr, float momentum, float decay) {
    if (biases && dbiases) {
        axpy(bsize, -lr / batch, dbiases, biases);
        scal(bsize, momentum, dbiases);
    }
    if (weights && dweights) {
        axpy(wsize, decay * batch, weights, dweights);
        axpy(wsize, -lr / batch, dweights, weights);
        scal(wsize, momentum, dweights);
    }
}

/* Adam: bcnn_learner.c:106-131. The nine BLAS-1 sweeps of the reference written per element with the same
 * separately rounded operations (bcnn_axpby, bcnn_vmul, bcnn_pow(.,0.5) = powf, bcnn_add_scalar(1e-7),
 * bcnn_vdiv -- AVX build: a plain division, bcnn_mat.c:293-306). `iter` is learner->seen at the call sites
 * (bcnn_conv_layer.c:823), i.e. samples seen. */
void orc_adam_update(float *weights, float *biases, float *dweights, float *dbiases, float *adam_m,
                     float *adam_v, int wsize, int bsize, int batch, int iter, float beta1, float beta2,
                     float lr, float momentum, float decay) {
    const float mu = sqrtf(1.0f - powf(beta2, (float)iter + 1)) / (1.0f - powf(beta1, (float)iter + 1));
    if (biases && dbiases) {
        axpy(bsize, -lr / batch, dbiases, biases);
        scal(bsize, momentum, dbiases);
    }
    if (weights && dweights) {
        const float a1 = 1.0f - beta1, a2 = 1.0f - beta2, step = -lr / batch * mu;
        axpy(wsize, decay * batch, weights, dweights);
        for (int i = 0; i < wsize; ++i) {
            volatile float t0 = a1 * dweights[i], t1 = beta1 * adam_m[i];
            adam_m[i] = t0 + t1;
            volatile float g2 = dweights[i] * dweights[i];
            volatile float t2 = a2 * g2, t3 = beta2 * adam_v[i];
            adam_v[i] = t2 + t3;
            volatile float d = powf(adam_v[i], 0.5f);
            d = d + 0.0000001f;
            volatile float q = adam_m[i] / d;
            volatile float upd = step * q;
            weights[i] = upd + weights[i];
            dweights[i] = 0.0f;
        }
    }
}
