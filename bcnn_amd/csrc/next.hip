// next.hip -- device kernels for the nodes next to the hot path: eltwise residual
// add with different operand shapes, the full-connected bias add, softmax. All HBM/latency-bound and
// tiny next to the convolutions; they exist so a training step never leaves the device.
#include <cfloat>

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

}  // extern "C"
