// depthwise.hip -- depthwise convolution forward / backward (HBM-bound direct kernels).
//
// Reference semantics: src/layers/bcnn_depthwise_conv_layer.c:165-293 (forward), :295-547 (backward):
// weights [C][k][k]; zero padding (out-of-image taps are skipped); taps accumulated kh outer / kw
// inner; + bias (bcnn_add_bias quirk) ; activation. Backward: dy *= act'(y) in place, dbias += sum,
// and -- only when the source carries a gradient -- dw += sum x*g and dx += w*g (accumulating).
// All reductions are two-level and deterministic (the reference CUDA kernel races at
// bcnn_depthwise_conv_layer.cu:113).
#include "chan_reduce.h"

namespace bcnn_hip {

struct DwShape {
    int N, C, H, W, OH, OW, ksz, stride, pad;
};

__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     const DwShape s, int act, unsigned total) {
    const unsigned gstride = gridDim.x * blockDim.x;
    for (unsigned o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gstride) {
        const unsigned ow = o % (unsigned)s.OW, t = o / (unsigned)s.OW;
        const unsigned oh = t % (unsigned)s.OH, plane = t / (unsigned)s.OH;
        const int c = (int)(plane % (unsigned)s.C);
        const float* src = x + (long long)plane * s.H * s.W;
        const float* wk = w + c * s.ksz * s.ksz;
        const int ih0 = (int)oh * s.stride - s.pad, iw0 = (int)ow * s.stride - s.pad;
        float val = 0.f;
        for (int kh = 0; kh < s.ksz; ++kh) {
            const int ih = ih0 + kh;
            for (int kw = 0; kw < s.ksz; ++kw) {
                const int iw = iw0 + kw;
                if ((unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W)
                    val = __fadd_rn(val, __fmul_rn(wk[kh * s.ksz + kw], src[ih * s.W + iw]));
            }
        }
        const float b = bias[c];
        if (b != 0.0f && b != 1.0f) val += b;
        y[o] = act_fwd_cheap(val, act, 0.f);
    }
}

// dx[n][c][ih][iw] += sum_{kh,kw} w[c][kh][kw] * g[n][c][oh][ow],  oh*stride - pad + kh == ih
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                          float* __restrict__ dx, const DwShape s,
                                                          unsigned total) {
    const unsigned gstride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const unsigned iw = i % (unsigned)s.W, t = i / (unsigned)s.W;
        const unsigned ih = t % (unsigned)s.H, plane = t / (unsigned)s.H;
        const int c = (int)(plane % (unsigned)s.C);
        const float* gp = g + (long long)plane * s.OH * s.OW;
        const float* wk = w + c * s.ksz * s.ksz;
        float acc = dx[i];
        for (int kh = s.ksz - 1; kh >= 0; --kh) {  // ascending oh, like the reference's scatter order
            const int th = (int)ih + s.pad - kh;
            if (th < 0 || th % s.stride) continue;
            const int oh = th / s.stride;
            if (oh >= s.OH) continue;
            for (int kw = s.ksz - 1; kw >= 0; --kw) {
                const int tw = (int)iw + s.pad - kw;
                if (tw < 0 || tw % s.stride) continue;
                const int ow = tw / s.stride;
                if (ow >= s.OW) continue;
                acc = __fadd_rn(acc, __fmul_rn(wk[kh * s.ksz + kw], gp[oh * s.OW + ow]));
            }
        }
        dx[i] = acc;
    }
}

// dw[c][kh][kw] partial sums: grid (C, splits); every thread owns KS*KS accumulators.
template <int KS>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const DwShape s, int splits,
                                                            float* __restrict__ partials) {
    constexpr int NT = KS * KS;
    __shared__ float red[4][NT];
    const int c = blockIdx.x, sp = blockIdx.y;
    const int OHOW = s.OH * s.OW, M = s.N * OHOW;
    const int per = (M + splits - 1) / splits;
    const int lo = sp * per;
    int hi = lo + per;
    if (hi > M) hi = M;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    for (int idx = lo + threadIdx.x; idx < hi; idx += 256) {
        const int n = idx / OHOW, pix = idx - n * OHOW;
        const int oh = pix / s.OW, ow = pix - oh * s.OW;
        const long long plane = (long long)n * s.C + c;
        const float gv = g[plane * OHOW + pix];
        const float* src = x + plane * s.H * s.W;
        const int ih0 = oh * s.stride - s.pad, iw0 = ow * s.stride - s.pad;
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int ih = ih0 + kh, iw = iw0 + kw;
                if ((unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W)
                    acc[kh * KS + kw] += src[ih * s.W + iw] * gv;
            }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float v = wave_sum(acc[t]);
        if (lane == 0) red[wid][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < NT)
        partials[((long long)c * splits + sp) * NT + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// generic kernel size: one tap per blockIdx.z
__global__ __launch_bounds__(256) void dw_bwd_weight_tap_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                const DwShape s, int splits,
                                                                float* __restrict__ partials) {
    __shared__ float red[4];
    const int c = blockIdx.x, sp = blockIdx.y, tap = blockIdx.z;
    const int kh = tap / s.ksz, kw = tap % s.ksz, NT = s.ksz * s.ksz;
    const int OHOW = s.OH * s.OW, M = s.N * OHOW;
    const int per = (M + splits - 1) / splits;
    const int lo = sp * per;
    int hi = lo + per;
    if (hi > M) hi = M;
    float acc = 0.f;
    for (int idx = lo + threadIdx.x; idx < hi; idx += 256) {
        const int n = idx / OHOW, pix = idx - n * OHOW;
        const int oh = pix / s.OW, ow = pix - oh * s.OW;
        const long long plane = (long long)n * s.C + c;
        const int ih = oh * s.stride - s.pad + kh, iw = ow * s.stride - s.pad + kw;
        if ((unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W)
            acc += x[plane * s.H * s.W + ih * s.W + iw] * g[plane * OHOW + pix];
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) partials[((long long)c * splits + sp) * NT + tap] = t;
}

__global__ void dw_weight_accumulate_kernel(const float* __restrict__ partials, int C, int NT, int splits,
                                            float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * NT) return;
    const int c = i / NT, t = i - c * NT;
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += (double)partials[((long long)c * splits + k) * NT + t];
    dw[i] += (float)s;
}

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_depthwise_forward(const float* x, const float* w, const float* bias, float* y, int n, int c,
                                int h, int wd, int k, int stride, int pad, int act) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total = (long long)n * c * s.OH * s.OW;
    if (total <= 0) return;
    const int fused = act_is_cheap(act) ? act : BCNN_HIP_ACT_NONE;
    dw_fwd_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(x, w, bias, y, s, fused,
                                                                               (unsigned)total);
    KERNEL_CHECK();
    if (fused != act) bcnn_hip_activation_forward(y, (size_t)total, act, nullptr, s.OH * s.OW, c);
}

void bcnn_hip_depthwise_backward(const float* x, const float* w, const float* y, float* dy, float* dx,
                                 float* dw, float* dbias, int n, int c, int h, int wd, int k, int stride,
                                 int pad, int act) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const int ohow = s.OH * s.OW;
    const long long total_o = (long long)n * c * ohow;
    if (total_o <= 0) return;
    bcnn_hip_activation_backward(y, dy, (size_t)total_o, act, nullptr, nullptr, ohow, c);
    bcnn_hip_grad_bias(dbias, dy, n, c, ohow);
    if (!dx) return;  // reference: dW and dX are both skipped when the source has no gradient (:318, :432)
    const int NT = k * k;
    const long long M = (long long)n * ohow;
    const int splits = chan_splits(c, M);
    float* part = reduce_scratch((size_t)c * splits * NT);
    dim3 grid((unsigned)c, (unsigned)splits);
    if (k == 3) dw_bwd_weight_kernel<3><<<grid, 256, 0, current_stream()>>>(x, dy, s, splits, part);
    else if (k == 5) dw_bwd_weight_kernel<5><<<grid, 256, 0, current_stream()>>>(x, dy, s, splits, part);
    else {
        dim3 g3((unsigned)c, (unsigned)splits, (unsigned)NT);
        dw_bwd_weight_tap_kernel<<<g3, 256, 0, current_stream()>>>(x, dy, s, splits, part);
    }
    KERNEL_CHECK();
    dw_weight_accumulate_kernel<<<ceil_div(c * NT, 256), 256, 0, current_stream()>>>(part, c, NT, splits, dw);
    KERNEL_CHECK();
    const long long total_i = (long long)n * c * h * wd;
    dw_bwd_data_kernel<<<stream_grid((size_t)total_i, 256), 256, 0, current_stream()>>>(dy, w, dx, s,
                                                                                     (unsigned)total_i);
    KERNEL_CHECK();
}

}  // extern "C"
