// gemm.hip -- row-major SGEMM with the four transpose combinations on the fp32 matrix cores, plus
// stand-alone im2col / col2im. These are the C-ABI counterparts of bcnn_gemm (reference
// src/kernels/bcnn_mat.c:2627-2650), bcnn_im2col (:817-854) and bcnn_col2im (:935-970); the conv node
// itself never calls them (its im2col is fused into the implicit GEMM), the full-connected node does.
#include "conv_common.h"
#include "chan_reduce.h"

namespace bcnn_hip {

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int m, n, k;
    long long ars, acs, brs, bcs;  // element (i,l) of op(A) at A[i*ars + l*acs]; (l,j) of op(B) at B[l*brs + j*bcs]
    int ldc;
    float alpha, beta;
    // split-K (gridDim.z > 1): workgroup z takes k-tiles [z * tiles_per_split, ...) and writes its raw partial tile to
    // partials[z][m][n]; gemm_splitk_finalize_kernel combines them in order (deterministic) and applies alpha / beta
    float* partials;
    int tiles_per_split;
};

// 64x64 tile per workgroup, 2x2 waves of one 32x32 accumulator, BK = 16, register-prefetched staging.
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs a) {
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ float As[2][BK][BM + 1];
    __shared__ float Bs[2][BK][BN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
    const int nk_all = (a.k + BK - 1) / BK;
    const int kt0 = (int)blockIdx.z * a.tiles_per_split;
    const int nk = min(a.tiles_per_split, nk_all - kt0);
    // staging map, per operand by which of its strides is 1 (the lanes of a wave run along the contiguous direction):
    //   rows contiguous (stride along k is the leading dimension): thread -> (row tid % 64, k column tid / 64 + 4 i)
    //   k contiguous:                                              thread -> (k column tid % 16, row tid / 16 + 16 i)
    const bool a_kc = a.acs == 1 && a.ars != 1, b_kc = a.brs == 1 && a.bcs != 1;
    const int ar = a_kc ? (tid >> 4) : (tid & 63), ak = a_kc ? (tid & 15) : (tid >> 6);
    const int br = b_kc ? (tid >> 4) : (tid & 63), bk = b_kc ? (tid & 15) : (tid >> 6);
    const int ars_i = a_kc ? 16 : 0, aks_i = a_kc ? 0 : 4, brs_i = b_kc ? 16 : 0, bks_i = b_kc ? 0 : 4;
    float ra[4], rb[4];
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int la = (kt0 + kt) * BK + ak + aks_i * i, rowa = i0 + ar + ars_i * i;
            const int lb = (kt0 + kt) * BK + bk + bks_i * i, rowb = j0 + br + brs_i * i;
            ra[i] = (rowa < a.m && la < a.k) ? a.A[(long long)rowa * a.ars + (long long)la * a.acs] : 0.f;
            rb[i] = (rowb < a.n && lb < a.k) ? a.B[(long long)lb * a.brs + (long long)rowb * a.bcs] : 0.f;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            As[buf][ak + aks_i * i][ar + ars_i * i] = ra[i];
            Bs[buf][bk + bks_i * i][br + brs_i * i] = rb[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    load(0);
    store(0);
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks)
            acc = mfma32(As[cur][2 * ks + lhi][wm * 32 + l31], Bs[cur][2 * ks + lhi][wn * 32 + l31], acc);
        if (kt + 1 < nk) store(cur ^ 1);
        __syncthreads();
    }
    const int j = j0 + wn * 32 + l31;
    if (j < a.n) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = i0 + wm * 32 + mfma_row(q, lane);
            if (i >= a.m) continue;
            if (gridDim.z > 1) {  // uniform
                a.partials[((long long)blockIdx.z * a.m + i) * a.n + j] = acc[q];
                continue;
            }
            float* cp = a.C + (long long)i * a.ldc + j;
            float v = a.alpha * acc[q];
            if (a.beta != 0.0f) v += a.beta * (*cp);
            *cp = v;
        }
    }
}

// C = alpha * (partials[0] + partials[1] + ...) + beta * C, the partial tiles added in split order
__global__ __launch_bounds__(256) void gemm_splitk_finalize_kernel(const float* __restrict__ partials, int splits, float* C,
                                                                   int m, int n, int ldc, float alpha, float beta) {
    const long long total = (long long)m * n, stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        float sum = partials[e];
        for (int sp = 1; sp < splits; ++sp) sum += partials[(long long)sp * total + e];
        float* cp = C + (e / n) * ldc + (e % n);
        float v = alpha * sum;
        if (beta != 0.0f) v += beta * (*cp);
        *cp = v;
    }
}

__global__ __launch_bounds__(256) void gemm_scale_kernel(float* C, int m, int n, int ldc, float beta) {
    const long long total = (long long)m * n, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float* p = C + (i / n) * ldc + (i % n);
        *p = (beta == 0.0f) ? 0.f : (*p) * beta;
    }
}

// ---- im2col / col2im (one image) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ im, float* __restrict__ col, int C,
                                                     int H, int W, int ksz, int pad, int stride, int OH,
                                                     int OW, unsigned total) {
    const unsigned gs = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const unsigned ow = i % (unsigned)OW, t = i / (unsigned)OW;
        const unsigned oh = t % (unsigned)OH, row = t / (unsigned)OH;
        const int kc = (int)(row % (unsigned)ksz), kr = (int)((row / (unsigned)ksz) % (unsigned)ksz);
        const int c = (int)(row / (unsigned)(ksz * ksz));
        const int ih = (int)oh * stride - pad + kr, iw = (int)ow * stride - pad + kc;
        col[i] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) ? im[(c * H + ih) * W + iw] : 0.f;
    }
}

// gather form: im[c][ih][iw] = sum over (kr,kc) of col[(c,kr,kc)][oh][ow]; overwrites (zero-fill semantics)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ col, float* __restrict__ im, int C,
                                                     int H, int W, int ksz, int pad, int stride, int OH,
                                                     int OW, unsigned total) {
    const unsigned gs = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int iw = (int)(i % (unsigned)W), t = (int)(i / (unsigned)W);
        const int ih = t % H, c = t / H;
        float acc = 0.f;
        for (int kr = 0; kr < ksz; ++kr) {
            const int th = ih + pad - kr;
            if (th < 0 || th % stride) continue;
            const int oh = th / stride;
            if (oh >= OH) continue;
            for (int kc = 0; kc < ksz; ++kc) {
                const int tw = iw + pad - kc;
                if (tw < 0 || tw % stride) continue;
                const int ow = tw / stride;
                if (ow >= OW) continue;
                acc += col[(((long long)c * ksz + kr) * ksz + kc) * OH * OW + oh * OW + ow];
            }
        }
        im[i] = acc;
    }
}

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_gemm(int ta, int tb, int m, int n, int k, float alpha, const float* A, int lda, const float* B,
                   int ldb, float beta, float* C, int ldc) {
    if (m <= 0 || n <= 0) return;
    if (k <= 0 || alpha == 0.0f) {
        if (beta != 1.0f) {
            gemm_scale_kernel<<<stream_grid((size_t)m * n, 256), 256, 0, current_stream()>>>(C, m, n, ldc, beta);
            KERNEL_CHECK();
        }
        return;
    }
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.m = m; a.n = n; a.k = k; a.ldc = ldc; a.alpha = alpha; a.beta = beta;
    a.ars = ta ? 1 : lda; a.acs = ta ? lda : 1;
    a.brs = tb ? 1 : ldb; a.bcs = tb ? ldb : 1;
    // few output tiles and a long reduction (the full-connected head: 128 x 1000 x 512 and its two backward products): the
    // k range is cut so that ~two workgroups per CU exist, each writing a partial tile
    const int tiles = ceil_div(n, 64) * ceil_div(m, 64), ktiles = ceil_div(k, 16);
    int splits = 1;
    if (tiles < kCUs / 2 && ktiles >= 16) {
        splits = (2 * kCUs) / tiles;
        if (splits > ktiles / 4) splits = ktiles / 4;  // at least four k-tiles per split
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
    }
    a.tiles_per_split = ceil_div(ktiles, splits);
    splits = ceil_div(ktiles, a.tiles_per_split);
    a.partials = splits > 1 ? reduce_scratch((size_t)splits * m * n) : nullptr;
    dim3 grid((unsigned)ceil_div(n, 64), (unsigned)ceil_div(m, 64), (unsigned)splits);
    gemm_kernel<<<grid, 256, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    if (splits > 1) {
        gemm_splitk_finalize_kernel<<<stream_grid((size_t)m * n, 256), 256, 0, current_stream()>>>(a.partials, splits, C, m, n, ldc,
                                                                                                alpha, beta);
        KERNEL_CHECK();
    }
}

void bcnn_hip_im2col(const float* im, int channels, int height, int width, int ksize, int pad, int stride,
                     float* col) {
    const int oh = (height + 2 * pad - ksize) / stride + 1, ow = (width + 2 * pad - ksize) / stride + 1;
    const long long total = (long long)channels * ksize * ksize * oh * ow;
    if (total <= 0) return;
    im2col_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(
        im, col, channels, height, width, ksize, pad, stride, oh, ow, (unsigned)total);
    KERNEL_CHECK();
}

void bcnn_hip_col2im(const float* col, int channels, int height, int width, int ksize, int pad, int stride,
                     float* im) {
    const int oh = (height + 2 * pad - ksize) / stride + 1, ow = (width + 2 * pad - ksize) / stride + 1;
    const long long total = (long long)channels * height * width;
    if (total <= 0) return;
    col2im_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(
        col, im, channels, height, width, ksize, pad, stride, oh, ow, (unsigned)total);
    KERNEL_CHECK();
}

}  // extern "C"
