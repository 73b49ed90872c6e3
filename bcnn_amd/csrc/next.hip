// next.hip -- device kernels for the nodes next to the hot path: eltwise residual
// add with different operand shapes, the full-connected bias add, softmax. All HBM/latency-bound and
// tiny next to the convolutions; they exist so a training step never leaves the device.
#include <cfloat>

#include "chan_reduce.h"
#include "common.h"

namespace bcnn_hip {

// bcnn_axpy_strided, reference src/kernels/bcnn_mat.c:159-177
__global__ __launch_bounds__(256) void axpy_strided_kernel(float a, const float* __restrict__ x, float* __restrict__ y,
                                                           int sy, int sx, int xc, int xh, int xw, int yc, int yh,
                                                           int yw, int mc, int mh, int mw, unsigned total) {
    const unsigned gs = gridDim.x * blockDim.x;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gs) {
        const unsigned i = t % (unsigned)mw, t1 = t / (unsigned)mw;
        const unsigned j = t1 % (unsigned)mh, t2 = t1 / (unsigned)mh;
        const unsigned k = t2 % (unsigned)mc, n = t2 / (unsigned)mc;
        const size_t di = (size_t)i * sy + (size_t)yw * ((size_t)j * sy + (size_t)yh * ((size_t)yc * n + k));
        const size_t si = (size_t)i * sx + (size_t)xw * ((size_t)j * sx + (size_t)xh * ((size_t)xc * n + k));
        y[di] += a * x[si];
    }
}

__global__ __launch_bounds__(256) void add_rowvec_kernel(float* __restrict__ y, const float* __restrict__ v, int cols,
                                                         unsigned total) {
    const unsigned gs = gridDim.x * blockDim.x;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gs) y[t] += v[t % (unsigned)cols];
}

// eltwise node, same-shape path, one pass each way (16-byte accesses on the aligned body)
__global__ __launch_bounds__(256) void eltwise_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ y, size_t n, size_t b_count, int act) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 v = *reinterpret_cast<const float4*>(a + i);
            if (i + 4 <= b_count) {
                const float4 w = *reinterpret_cast<const float4*>(b + i);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            } else if (i < b_count) {
                if (i < b_count) v.x += b[i];
                if (i + 1 < b_count) v.y += b[i + 1];
                if (i + 2 < b_count) v.z += b[i + 2];
            }
            v.x = act_fwd_cheap(v.x, act, 0.f); v.y = act_fwd_cheap(v.y, act, 0.f);
            v.z = act_fwd_cheap(v.z, act, 0.f); v.w = act_fwd_cheap(v.w, act, 0.f);
            *reinterpret_cast<float4*>(y + i) = v;
        } else {
            for (size_t k = i; k < n; ++k) y[k] = act_fwd_cheap(a[k] + (k < b_count ? b[k] : 0.f), act, 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void eltwise_bwd_kernel(const float* __restrict__ y, float* __restrict__ dy,
                                                          float* __restrict__ da, float* __restrict__ db, size_t n,
                                                          size_t b_count, int act, int overwrite_a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 g = *reinterpret_cast<const float4*>(dy + i);
            if (act != BCNN_HIP_ACT_NONE) {
                const float4 yv = *reinterpret_cast<const float4*>(y + i);
                g.x *= act_bwd_cheap(yv.x, act, 0.f); g.y *= act_bwd_cheap(yv.y, act, 0.f);
                g.z *= act_bwd_cheap(yv.z, act, 0.f); g.w *= act_bwd_cheap(yv.w, act, 0.f);
                *reinterpret_cast<float4*>(dy + i) = g;
            }
            if (da) {
                // overwrite_a: da holds zeros semantically (its fill was skipped); 0.f + g keeps the sign of zero
                float4 t = overwrite_a ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(da + i);
                t.x += g.x; t.y += g.y; t.z += g.z; t.w += g.w;
                *reinterpret_cast<float4*>(da + i) = t;
            }
            if (db && i < b_count) {
                if (i + 4 <= b_count) {
                    float4 t = *reinterpret_cast<const float4*>(db + i);
                    t.x += g.x; t.y += g.y; t.z += g.z; t.w += g.w;
                    *reinterpret_cast<float4*>(db + i) = t;
                } else {
                    db[i] += g.x;
                    if (i + 1 < b_count) db[i + 1] += g.y;
                    if (i + 2 < b_count) db[i + 2] += g.z;
                }
            }
        } else {
            for (size_t k = i; k < n; ++k) {
                float gk = dy[k];
                if (act != BCNN_HIP_ACT_NONE) { gk *= act_bwd_cheap(y[k], act, 0.f); dy[k] = gk; }
                if (da) da[k] = (overwrite_a ? 0.f : da[k]) + gk;
                if (db && k < b_count) db[k] += gk;
            }
        }
    }
}

// one wave per (n, spatial position): max, log-sum-exp, exp(x - lse); exp/log in double like the reference
// (src/layers/bcnn_softmax_layer.c:95-123). The reference adds the exponentials sequentially in float; here
// the 64 lane partials are accumulated in double and rounded once, which differs from it by a few float ulps.
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW,
                                                      unsigned total) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    for (unsigned t = wave; t < total; t += nwaves) {
        const unsigned i = t % (unsigned)HW, n = t / (unsigned)HW;
        const float* px = x + (size_t)n * C * HW + i;
        float* py = y + (size_t)n * C * HW + i;
        float vmax = -FLT_MAX;
        for (int c = lane; c < C; c += 64) vmax = fmaxf(vmax, px[(size_t)c * HW]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        double part = 0.0;
        for (int c = lane; c < C; c += 64) part += (double)(float)exp((double)(px[(size_t)c * HW] - vmax));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        float sum = (float)part;
        sum = (sum != 0.f) ? vmax + (float)log((double)sum) : vmax - 100.0f;
        for (int c = lane; c < C; c += 64) py[(size_t)c * HW] = (float)exp((double)(px[(size_t)c * HW] - sum));
    }
}

// ---- cost metric on the device -----------------------------------------------------------------------
// The scalar the cost node reports (bcnn_compute_error, bcnn_cost_layer.c:142-244). The reference's GPU build
// copies the whole prediction and gradient to the host for it (three blocking transfers, ~0.26 ms of idle GPU per
// ResNet-18 step here); one 1024-thread workgroup computes it in place and only the 4-byte result travels.
// Sums are accumulated in double like the host loops; per-row work is done by one wave with the reference's
// "first strict maximum above FLT_MIN wins" rule.
// Grid of `gridDim.x` workgroups, each leaves one double in partials[blockIdx.x]; cost_metric_final_kernel adds them in
// index order (a single workgroup over 256 x 1000 values took 97 us of the MobileNet step).
__global__ __launch_bounds__(1024) void cost_metric_kernel(int metric, const float* __restrict__ pred,
                                                           const float* __restrict__ label,
                                                           const float* __restrict__ grad, int B, int per,
                                                           double* __restrict__ partials) {
    __shared__ double red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrow = 16 * gridDim.x, row0 = blockIdx.x * 16 + wave;
    double acc = 0.0;
    if (metric == 0) {  // BCNN_METRIC_ERROR_RATE
        for (int i = row0; i < B; i += nrow) {
            const float* row = pred + (size_t)i * per;
            float pm = FLT_MIN;
            int best = 0;
            for (int j = lane; j < per; j += 64)
                if (row[j] > pm) { pm = row[j]; best = j; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float om = __shfl_xor(pm, off);
                const int ob = __shfl_xor(best, off);
                if (om > pm || (om == pm && ob < best)) { pm = om; best = ob; }
            }
            if (lane == 0 && label[(size_t)i * per + best] == 0) acc += 1.0;
        }
    } else if (metric == 5) {  // BCNN_METRIC_DICE
        for (int i = row0; i < B; i += nrow) {
            int n = 0, d = 0;
            for (int j = lane; j < per; j += 64) {
                const float l = label[(size_t)i * per + j];
                const float t = pred[(size_t)i * per + j] > 0.5f ? 1.f : 0.f;
                n += (int)(l * t);
                d += (int)(l + t);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { n += __shfl_xor(n, off); d += __shfl_xor(d, off); }
            if (lane == 0) acc += (2.0f * n + 1.0f) / (d + 1.0f);
        }
    } else {
        const size_t sz = (size_t)B * per;
        for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < sz; i += (size_t)gridDim.x * 1024) {
            if (metric == 1) {  // BCNN_METRIC_LOGLOSS
                if (label[i] > 0.0f) {
                    float q = pred[i];
                    q = q < 1e-8f ? 1e-8f : (q > 1.0f - 1e-8f ? 1.0f - 1e-8f : q);
                    acc += -log((double)q);
                }
            } else {            // SSE / MSE / CRPS: sum of squared errors held in the node's gradient
                acc += (double)grad[i] * grad[i];
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        partials[blockIdx.x] = t;
    }
}
__global__ void cost_metric_final_kernel(int metric, const double* __restrict__ partials, int n, int per,
                                         float* __restrict__ out) {
    double t = 0.0;
    for (int i = 0; i < n; ++i) t += partials[i];
    out[0] = (float)(metric == 3 ? t / per : t);  // BCNN_METRIC_MSE divides by the per-sample size
}

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_axpy_strided(int num_batches, float a, const float* x, float* y, int stride_y, int stride_x, int x_c,
                           int x_h, int x_w, int y_c, int y_h, int y_w, int min_c, int min_h, int min_w) {
    const long long total = (long long)num_batches * min_c * min_h * min_w;
    if (total <= 0) return;
    axpy_strided_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(
        a, x, y, stride_y, stride_x, x_c, x_h, x_w, y_c, y_h, y_w, min_c, min_h, min_w, (unsigned)total);
    KERNEL_CHECK();
}

void bcnn_hip_add_rowvec(float* y, const float* v, int rows, int cols) {
    const long long total = (long long)rows * cols;
    if (total <= 0) return;
    add_rowvec_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(y, v, cols, (unsigned)total);
    KERNEL_CHECK();
}

void bcnn_hip_softmax_forward(const float* x, float* y, int n, int c, int hw) {
    const long long total = (long long)n * hw;
    if (total <= 0) return;
    softmax_kernel<<<stream_grid((size_t)total * 64, 256), 256, 0, current_stream()>>>(x, y, c, hw, (unsigned)total);
    KERNEL_CHECK();
}

static bool aligned16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void bcnn_hip_eltwise_forward(const float* a, const float* b, float* y, size_t n, size_t b_count, int act) {
    if (!n) return;
    if (!act_is_cheap(act) || !aligned16(a) || !aligned16(b) || !aligned16(y)) {  // three-pass form of the reference
        bcnn_hip_copy_f32(n, a, y);
        bcnn_hip_axpy(b_count, 1.0f, b, y);
        bcnn_hip_activation_forward(y, n, act, nullptr, 1, 1);
        return;
    }
    eltwise_fwd_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, current_stream()>>>(a, b, y, n, b_count, act);
    KERNEL_CHECK();
}

void bcnn_hip_eltwise_backward(const float* y, float* dy, float* da, float* db, size_t n, size_t b_count, int act,
                               int overwrite_a) {
    if (!n) return;
    if (!act_bwd_is_cheap(act) || !aligned16(y) || !aligned16(dy) || !aligned16(da) || !aligned16(db)) {
        bcnn_hip_activation_backward(y, dy, n, act, nullptr, nullptr, 1, 1);
        if (da && overwrite_a) bcnn_hip_fill_f32(da, n, 0.f);
        if (da) bcnn_hip_axpy(n, 1.0f, dy, da);
        if (db) bcnn_hip_axpy(b_count, 1.0f, dy, db);
        return;
    }
    eltwise_bwd_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, current_stream()>>>(y, dy, da, db, n, b_count, act,
                                                                                   overwrite_a);
    KERNEL_CHECK();
}

void bcnn_hip_cost_metric(int metric, const float* pred, const float* label, const float* grad, int batch, int per,
                          float* out) {
    const long long sz = (long long)batch * per;
    int blocks = (int)((sz + 8191) / 8192);  // >= 8 values per thread before a second workgroup pays
    if (blocks > 64) blocks = 64;
    if (blocks < 1) blocks = 1;
    double* partials = reinterpret_cast<double*>(reduce_scratch(2 * 64));
    cost_metric_kernel<<<blocks, 1024, 0, current_stream()>>>(metric, pred, label, grad, batch, per, partials);
    KERNEL_CHECK();
    cost_metric_final_kernel<<<1, 1, 0, current_stream()>>>(metric, partials, blocks, per, out);
    KERNEL_CHECK();
}

}  // extern "C"
