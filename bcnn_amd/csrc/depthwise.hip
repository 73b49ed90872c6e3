// depthwise.hip -- depthwise convolution forward / backward (HBM-bound direct kernels).
//
// Reference semantics: src/layers/bcnn_depthwise_conv_layer.c:165-293 (forward), :295-547 (backward):
// weights [C][k][k]; zero padding (out-of-image taps are skipped); taps accumulated kh outer / kw
// inner; + bias (bcnn_add_bias quirk) ; activation. Backward: dy *= act'(y) in place, dbias += sum,
// and -- only when the source carries a gradient -- dw += sum x*g and dx += w*g (accumulating).
// All reductions are two-level and deterministic (the reference CUDA kernel races at
// bcnn_depthwise_conv_layer.cu:113).
#include "chan_reduce.h"
#include "depthwise.h"

namespace bcnn_hip {

void activation_backward_grad_bias(const float* y, float* dy, float* dbias, int n, int c, int hw, int act);  // blas1.hip


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
                                                          unsigned total, int overwrite) {
    const unsigned gstride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const unsigned iw = i % (unsigned)s.W, t = i / (unsigned)s.W;
        const unsigned ih = t % (unsigned)s.H, plane = t / (unsigned)s.H;
        const int c = (int)(plane % (unsigned)s.C);
        const float* gp = g + (long long)plane * s.OH * s.OW;
        const float* wk = w + c * s.ksz * s.ksz;
        float acc = overwrite ? 0.f : dx[i];  // overwrite: the caller skipped the zero fill (sole writer)
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

// ================================================================================================
// 3x3 fast paths: one thread per FOUR consecutive output columns of a row. The (plane, row, column group)
// index is split with two multiply-high divisions per thread instead of three hardware divisions per
// element; the 3 x (3*S + 3) input window is fetched once with zero fill (adding w * 0 is what skipping an
// out-of-image tap amounts to) and reused by the four outputs; outputs leave as one 16-byte store when the
// row allows it. Same tap order as the reference (kh outer, kw inner, separate multiply and add).
// ================================================================================================
#ifndef DW_VR
#define DW_VR 2  // output rows per thread of the 3x3 forward / weight-gradient kernels
#endif
struct Dw3Args {
    DwShape s;
    unsigned groups_per_row;   // ceil(OW / 4)
    unsigned row_groups;       // ceil(OH / VR): output rows are handled VR at a time
    unsigned gpr_magic, oh_magic;  // magic of groups_per_row / row_groups
    unsigned total_groups;     // planes * row_groups * groups_per_row
};
__device__ __forceinline__ unsigned dw_div(unsigned n, unsigned d, unsigned magic) {
    // magic = ceil(2^32 / d); one correction step makes it exact for every n < 2^32 / 2
    if (d == 1) return n;
    unsigned q = __umulhi(n, magic);
    if (q * d > n) --q;
    return q;
}

// zero-filled window of WIN consecutive floats starting at column iw0 of `row` (row == NULL: all zero).
// `fast` (wave-uniform): rows are 16-byte aligned and iw0 + 1 is a multiple of 4, so the body of the window
// is one or two aligned 16-byte loads and only the first / last column are scalar.
template <int WIN>
__device__ __forceinline__ void dw_window(const float* row, int iw0, int W, bool fast, float (&xv)[WIN]) {
    if (row == nullptr) {
#pragma unroll
        for (int i = 0; i < WIN; ++i) xv[i] = 0.f;
        return;
    }
    if (fast) {
        xv[0] = iw0 >= 0 ? row[iw0] : 0.f;
#pragma unroll
        for (int v = 0; v + 4 < WIN; v += 4) {
            const int col = iw0 + 1 + v;
            if (col + 3 < W) {
                const float4 t = *reinterpret_cast<const float4*>(row + col);
                xv[1 + v] = t.x; xv[2 + v] = t.y; xv[3 + v] = t.z; xv[4 + v] = t.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[1 + v + i] = (col + i < W) ? row[col + i] : 0.f;
            }
        }
#pragma unroll
        for (int i = 1 + ((WIN - 1) / 4) * 4; i < WIN; ++i) xv[i] = (iw0 + i < W) ? row[iw0 + i] : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            const int iw = iw0 + i;
            xv[i] = (unsigned)iw < (unsigned)W ? row[iw] : 0.f;
        }
    }
}

template <int S, int VR>  // VR output rows per thread: the (VR - 1) * S + 3 input rows are fetched once
__global__ __launch_bounds__(256) void dw3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      const Dw3Args a, int act) {
    constexpr int WIN = 3 * S + 3;  // input columns feeding 4 outputs
    constexpr int NR = (VR - 1) * S + 3;
    const DwShape& s = a.s;
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < a.total_groups; t += gridDim.x * 256u) {
    const unsigned rowid = dw_div(t, a.groups_per_row, a.gpr_magic), q = t - rowid * a.groups_per_row;
    const unsigned plane = dw_div(rowid, a.row_groups, a.oh_magic), og = rowid - plane * a.row_groups;
    const int c = (int)(plane % (unsigned)s.C);
    const float* src = x + (size_t)plane * s.H * s.W;
    const float* wk = w + c * 9;
    float wv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wv[i] = wk[i];
    const int oh0 = (int)og * VR, ow0 = (int)q * 4, ih0 = oh0 * S - s.pad, iw0 = ow0 * S - s.pad;
    float acc[VR][4];
#pragma unroll
    for (int r = 0; r < VR; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
    const bool fast = s.pad == 1 && (s.W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int ih = ih0 + i;
        const bool rv = (unsigned)ih < (unsigned)s.H;
        float xv[WIN];
        dw_window<WIN>(rv ? src + ih * s.W : nullptr, iw0, s.W, fast, xv);
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int kh = i - r * S;  // compile-time after unrolling; ascending i == ascending kh per output row
            if (kh < 0 || kh > 2) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[r][j] = __fadd_rn(acc[r][j], __fmul_rn(wv[kh * 3 + kw], xv[j * S + kw]));
        }
    }
    const float b = bias[c];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        const int oh = oh0 + r;
        if (oh >= s.OH) break;
        float* dst = y + ((size_t)plane * s.OH + oh) * s.OW + ow0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (b != 0.0f && b != 1.0f) acc[r][j] += b;
            acc[r][j] = act_fwd_cheap(acc[r][j], act, 0.f);
        }
        if (ow0 + 4 <= s.OW && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ow0 + j < s.OW) dst[j] = acc[r][j];
        }
    }
    }
}

// dw partial sums: grid-stride over the same 4-column groups, restricted to ONE channel per workgroup row
// (blockIdx.y = channel) so the nine accumulators reduce without atomics; blockIdx.x = split of the channel's
// (image, row, group) space.
template <int S, int VR>
__global__ __launch_bounds__(256) void dw3_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const Dw3Args a, int splits,
                                                             float* __restrict__ partials) {
    constexpr int WIN = 3 * S + 3;
    constexpr int NR = (VR - 1) * S + 3;
    __shared__ float red[4][9];
    const DwShape& s = a.s;
    const int c = blockIdx.y, sp = blockIdx.x;
    const unsigned per_img = a.row_groups * a.groups_per_row;          // thread items per (image, channel) plane
    const unsigned M = (unsigned)s.N * per_img;
    const unsigned per = (M + splits - 1) / splits;
    const unsigned lo = sp * per;
    unsigned hi = lo + per;
    if (hi > M) hi = M;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    const bool fast = s.pad == 1 && (s.W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    for (unsigned idx = lo + threadIdx.x; idx < hi; idx += 256) {
        const unsigned rowid = dw_div(idx, a.groups_per_row, a.gpr_magic), q = idx - rowid * a.groups_per_row;
        const unsigned n = dw_div(rowid, a.row_groups, a.oh_magic), og = rowid - n * a.row_groups;
        const size_t plane = (size_t)n * s.C + c;
        const float* src = x + plane * s.H * s.W;
        const int oh0 = (int)og * VR, ow0 = (int)q * 4, ih0 = oh0 * S - s.pad, iw0 = ow0 * S - s.pad;
        float gv[VR][4];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const float* gp = g + (plane * s.OH + oh0 + r) * s.OW;
            const bool rok = oh0 + r < s.OH;
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[r][j] = (rok && ow0 + j < s.OW) ? gp[ow0 + j] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int ih = ih0 + i;
            const bool rv = (unsigned)ih < (unsigned)s.H;
            float xv[WIN];
            dw_window<WIN>(rv ? src + ih * s.W : nullptr, iw0, s.W, fast, xv);
#pragma unroll
            for (int r = 0; r < VR; ++r) {
                const int kh = i - r * S;
                if (kh < 0 || kh > 2) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[kh * 3 + kw] += xv[j * S + kw] * gv[r][j];
            }
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wid][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)
        partials[((long long)c * splits + sp) * 9 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dx += gather, one thread per four consecutive INPUT columns; any stride (tap parity tested per pixel).
// Tap order per pixel as in dw_bwd_data_kernel (descending kh, kw = ascending output position).
struct Dw3DxArgs {
    DwShape s;
    unsigned groups_per_row;   // ceil(W / 4)
    unsigned row_groups;       // input rows (stride-1 kernel: ceil(H / VR) row groups)
    unsigned gpr_magic, h_magic;  // magic of groups_per_row / row_groups
    unsigned total_groups;     // planes * row_groups * groups_per_row
    int overwrite;             // dx is assigned 0 + sum instead of accumulated onto (no read of dx)
};
template <int S>  // S = compile-time stride (1, 2), 0 = runtime stride
__global__ __launch_bounds__(256) void dw3_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                           float* __restrict__ dx, const Dw3DxArgs a) {
    const DwShape& s = a.s;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= a.total_groups) return;
    const unsigned rowid = dw_div(t, a.groups_per_row, a.gpr_magic), q = t - rowid * a.groups_per_row;
    const unsigned plane = dw_div(rowid, a.row_groups, a.h_magic), ih = rowid - plane * a.row_groups;
    const int c = (int)(plane % (unsigned)s.C);
    const float* gp = g + (size_t)plane * s.OH * s.OW;
    const float* wk = w + c * 9;
    float wv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wv[i] = wk[i];
    const int iw0 = (int)q * 4;
    float* dst = dx + ((size_t)plane * s.H + ih) * s.W + iw0;
    const bool vec = iw0 + 4 <= s.W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    float acc[4];
    if (a.overwrite) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = 0.f;
    } else if (vec) {
        const float4 v = *reinterpret_cast<const float4*>(dst);
        acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (iw0 + j < s.W) ? dst[j] : 0.f;
    }
    if (S == 1 || S == 2) {
        // tw = e + j - kw with e = iw0 + pad; the dy columns the four pixels can meet form a short window that
        // is fetched once per row with zero fill: S == 1: ow = e - 2 .. e + 3; S == 2: ow = eb - 1 .. eb + 2
        // where e = 2 * eb + par (iw0 is a multiple of 4, so par is the parity of pad: wave-uniform)
        constexpr int WIN = (S == 1) ? 6 : 4;
        const int e = iw0 + s.pad;
        const int par = (S == 2) ? (e & 1) : 0;
        const int wbase = (S == 1) ? e - 2 : ((e - par) >> 1) - 1;
#pragma unroll
        for (int kh = 2; kh >= 0; --kh) {
            const int th = (int)ih + s.pad - kh;
            if (th < 0 || (S == 2 && (th & 1))) continue;
            const int oh = (S == 2) ? th >> 1 : th;
            if (oh >= s.OH) continue;
            const float* grow = gp + oh * s.OW;
            float gw[WIN];
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int ow = wbase + i;
                gw[i] = (unsigned)ow < (unsigned)s.OW ? grow[ow] : 0.f;
            }
            if (S == 1) {
#pragma unroll
                for (int kw = 2; kw >= 0; --kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __fadd_rn(acc[j], __fmul_rn(wv[kh * 3 + kw], gw[j + 2 - kw]));
            } else if (par == 0) {
#pragma unroll
                for (int kw = 2; kw >= 0; --kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (((j - kw) & 1) == 0)
                            acc[j] = __fadd_rn(acc[j], __fmul_rn(wv[kh * 3 + kw], gw[(j - kw + 2) / 2]));
            } else {
#pragma unroll
                for (int kw = 2; kw >= 0; --kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (((1 + j - kw) & 1) == 0)
                            acc[j] = __fadd_rn(acc[j], __fmul_rn(wv[kh * 3 + kw], gw[(1 + j - kw + 2) / 2]));
            }
        }
    } else {
#pragma unroll
        for (int kh = 2; kh >= 0; --kh) {
            const int stride = s.stride;
            const int th = (int)ih + s.pad - kh;
            if (th < 0) continue;
            const int oh = th / stride;
            if (oh * stride != th || oh >= s.OH) continue;
            const float* grow = gp + oh * s.OW;
#pragma unroll
            for (int kw = 2; kw >= 0; --kw)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int tw = iw0 + j + s.pad - kw;
                    const int ow = tw / stride;
                    if (tw >= 0 && ow * stride == tw && ow < s.OW)
                        acc[j] = __fadd_rn(acc[j], __fmul_rn(wv[kh * 3 + kw], grow[ow]));
                }
        }
    }
    if (vec) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (iw0 + j < s.W) dst[j] = acc[j];
    }
}

// stride 1: VR consecutive input rows per thread; the VR + 2 dy rows they meet are fetched once each
// (16-byte window loads when the rows allow). Per pixel the taps still arrive in descending kh, kw.
template <int VR>
__global__ __launch_bounds__(256) void dw3_bwd_data_s1_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                              float* __restrict__ dx, const Dw3DxArgs a) {
    const DwShape& s = a.s;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= a.total_groups) return;
    const unsigned rowid = dw_div(t, a.groups_per_row, a.gpr_magic), q = t - rowid * a.groups_per_row;
    const unsigned plane = dw_div(rowid, a.row_groups, a.h_magic), rg = rowid - plane * a.row_groups;
    const int c = (int)(plane % (unsigned)s.C);
    const float* gp = g + (size_t)plane * s.OH * s.OW;
    const float* wk = w + c * 9;
    float wv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wv[i] = wk[i];
    const int ih0 = (int)rg * VR, iw0 = (int)q * 4;
    float* dst0 = dx + ((size_t)plane * s.H + ih0) * s.W + iw0;
    const bool vec = iw0 + 4 <= s.W && (s.W & 3) == 0 && ((reinterpret_cast<uintptr_t>(dx) & 15) == 0);
    float acc[VR][4];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        if (ih0 + r < s.H && !a.overwrite) {
            if (vec) {
                const float4 v = *reinterpret_cast<const float4*>(dst0 + r * s.W);
                acc[r][0] = v.x; acc[r][1] = v.y; acc[r][2] = v.z; acc[r][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = (iw0 + j < s.W) ? dst0[r * s.W + j] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
        }
    }
    const int wbase = iw0 + s.pad - 2;  // first dy column the four pixels can meet
    const bool fast = s.pad == 1 && (s.OW & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0;
#pragma unroll
    for (int tt = 0; tt < VR + 2; ++tt) {
        const int oh = ih0 + s.pad - 2 + tt;
        float gw[6];
        dw_window<6>((unsigned)oh < (unsigned)s.OH ? gp + oh * s.OW : nullptr, wbase, s.OW, fast, gw);
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int kh = r + 2 - tt;  // oh = ih + pad - kh; ascending tt == descending kh per input row
            if (kh < 0 || kh > 2) continue;
#pragma unroll
            for (int kw = 2; kw >= 0; --kw)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[r][j] = __fadd_rn(acc[r][j], __fmul_rn(wv[kh * 3 + kw], gw[j + 2 - kw]));
        }
    }
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        if (ih0 + r >= s.H) break;
        if (vec) {
            *reinterpret_cast<float4*>(dst0 + r * s.W) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (iw0 + j < s.W) dst0[r * s.W + j] = acc[r][j];
        }
    }
}

static unsigned dw_magic(unsigned d) { return d > 1 ? (unsigned)((0x100000000ULL + d - 1) / d) : 0u; }

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_depthwise_forward(const float* x, const float* w, const float* bias, float* y, int n, int c,
                                int h, int wd, int k, int stride, int pad, int act) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total = (long long)n * c * s.OH * s.OW;
    if (total <= 0) return;
    KTimer kt(K_DEPTHWISE_FWD, 2.0 * (double)total * k * k, 4.0 * ((double)n * c * h * wd + (double)total));
    const int fused = act_is_cheap(act) ? act : BCNN_HIP_ACT_NONE;
    if (depthwise_forward_lds(x, w, bias, y, s, fused, nullptr)) {
        if (fused != act) bcnn_hip_activation_forward(y, (size_t)total, act, nullptr, s.OH * s.OW, c);
        return;
    }
    const unsigned gpr = (unsigned)ceil_div(s.OW, 4);
    constexpr int VR = DW_VR;
    const unsigned rgs = (unsigned)ceil_div(s.OH, VR);
    const long long groups = (long long)n * c * rgs * gpr;
    if (k == 3 && (stride == 1 || stride == 2) && groups < 0x7fffffffLL) {
        Dw3Args a;
        a.s = s; a.groups_per_row = gpr; a.row_groups = rgs; a.gpr_magic = dw_magic(gpr); a.oh_magic = dw_magic(rgs);
        a.total_groups = (unsigned)groups;
        unsigned blocks = (unsigned)((groups + 255) / 256);
#ifdef DW_PERSIST
        if (blocks > (unsigned)(kCUs * DW_PERSIST)) blocks = (unsigned)(kCUs * DW_PERSIST);
#endif
        if (stride == 1) dw3_fwd_kernel<1, VR><<<blocks, 256, 0, current_stream()>>>(x, w, bias, y, a, fused);
        else dw3_fwd_kernel<2, VR><<<blocks, 256, 0, current_stream()>>>(x, w, bias, y, a, fused);
    } else {
        dw_fwd_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(x, w, bias, y, s, fused,
                                                                                   (unsigned)total);
    }
    KERNEL_CHECK();
    if (fused != act) bcnn_hip_activation_forward(y, (size_t)total, act, nullptr, s.OH * s.OW, c);
}

size_t bcnn_hip_depthwise_stats_size(int n, int c, int h, int wd, int k, int stride, int pad) {
    if (stride < 1 || k < 1) return 0;
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    return depthwise_lds_stats_floats(s);
}

int bcnn_hip_depthwise_forward_stats(const float* x, const float* w, const float* bias, float* y, int n, int c, int h,
                                     int wd, int k, int stride, int pad, int act, float* stats, size_t stats_floats) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total = (long long)n * c * s.OH * s.OW;
    if (total > 0 && stats && act_is_cheap(act)) {
        // the statistics are those of the STORED output, so the activation has to be fused in the kernel
        KTimer kt(K_DEPTHWISE_FWD, 2.0 * (double)total * k * k, 4.0 * ((double)n * c * h * wd + (double)total));
        ConvStats st;
        st.partials = stats; st.capacity = stats_floats; st.splits = 0;
        if (depthwise_forward_lds(x, w, bias, y, s, act, &st)) return st.splits;
    }
    bcnn_hip_depthwise_forward(x, w, bias, y, n, c, h, wd, k, stride, pad, act);
    return 0;
}

int bcnn_hip_depthwise_bn_fusable(int n, int c, int h, int wd, int k, int stride, int pad, int act) {
    if (stride < 1 || k < 1) return 0;
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    return depthwise_lds_ok(s) && act_bwd_is_cheap(act) && act != BCNN_HIP_ACT_PRELU;
}

void bcnn_hip_depthwise_backward_bn(const float* x, const float* w, const float* y, const float* dz, float* dx, float* dw,
                                    float* dbias, int n, int c, int h, int wd, int k, int stride, int pad, int act,
                                    int overwrite, const float* bn_mean, const float* bn_var, const float* bn_scales,
                                    const float* bn_dmean, const float* bn_dvar) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total_o = (long long)n * c * s.OH * s.OW;
    if (total_o <= 0) return;
    // algorithmic bytes: dz, y, x read; dx written (read too when it accumulates)
    KTimer kt(K_DEPTHWISE_BWD, 4.0 * (double)total_o * k * k,
              4.0 * (2.0 * (double)total_o + (overwrite ? 2.0 : 3.0) * (double)n * c * h * wd));
    DwBnBwd bn{dz, bn_mean, bn_var, bn_scales, bn_dmean, bn_dvar};
    if (!dx || !depthwise_backward_lds(x, w, y, nullptr, dx, dw, dbias, s, act, overwrite, 0, &bn)) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_depthwise_backward_bn: shape not fusable (ask bcnn_hip_depthwise_bn_fusable)\n");
        exit(1);
    }
}

// ---- the same three with the producing convolution node's batch-norm applied to the input on the fly ----------------
int bcnn_hip_depthwise_bnin_fusable(int n, int c, int h, int wd, int k, int stride, int pad, int act, int in_act) {
    if (stride < 1 || k < 1) return 0;
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    return depthwise_lds_ok(s) && act_is_cheap(act) && act_bwd_is_cheap(act) && act != BCNN_HIP_ACT_PRELU &&
           act_is_cheap(in_act) && in_act != BCNN_HIP_ACT_PRELU;
}

static void dw_bnin_refused(const char* who) {
    fprintf(stderr, "[bcnn_hip] %s: shape / activation not fusable (ask bcnn_hip_depthwise_bnin_fusable)\n", who);
    exit(1);
}

int bcnn_hip_depthwise_forward_bnin(const float* x_raw, const float* w, const float* bias, float* y, int n, int c, int h,
                                    int wd, int k, int stride, int pad, int act, float* stats, size_t stats_floats,
                                    const float* in_mean, const float* in_var, const float* in_scale, const float* in_bias,
                                    int in_act) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total = (long long)n * c * s.OH * s.OW;
    if (total <= 0) return 0;
    KTimer kt(K_DEPTHWISE_FWD, 2.0 * (double)total * k * k, 4.0 * ((double)n * c * h * wd + (double)total));
    ConvStats st;
    st.partials = stats; st.capacity = stats_floats; st.splits = 0;
    DwBnIn in{in_mean, in_var, in_scale, in_bias, in_act};
    if (!depthwise_forward_lds(x_raw, w, bias, y, s, act, stats ? &st : nullptr, &in)) dw_bnin_refused("bcnn_hip_depthwise_forward_bnin");
    return st.splits;
}

void bcnn_hip_depthwise_backward_bnin(const float* x_raw, const float* w, const float* y, float* dy, float* dx, float* dw,
                                      float* dbias, int n, int c, int h, int wd, int k, int stride, int pad, int act,
                                      int overwrite, const float* in_mean, const float* in_var, const float* in_scale,
                                      const float* in_bias, int in_act) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total_o = (long long)n * c * s.OH * s.OW;
    if (total_o <= 0) return;
    KTimer kt(K_DEPTHWISE_BWD, 4.0 * (double)total_o * k * k,
              4.0 * (3.0 * (double)total_o + (overwrite ? 2.0 : 3.0) * (double)n * c * h * wd));
    DwBnIn in{in_mean, in_var, in_scale, in_bias, in_act};
    if (!dx || !depthwise_backward_lds(x_raw, w, y, dy, dx, dw, dbias, s, act, overwrite, /*write_back=*/1, nullptr, &in))
        dw_bnin_refused("bcnn_hip_depthwise_backward_bnin");
}

void bcnn_hip_depthwise_backward_bn_bnin(const float* x_raw, const float* w, const float* y, const float* dz, float* dx,
                                         float* dw, float* dbias, int n, int c, int h, int wd, int k, int stride, int pad,
                                         int act, int overwrite, const float* bn_mean, const float* bn_var,
                                         const float* bn_scales, const float* bn_dmean, const float* bn_dvar,
                                         const float* in_mean, const float* in_var, const float* in_scale,
                                         const float* in_bias, int in_act) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total_o = (long long)n * c * s.OH * s.OW;
    if (total_o <= 0) return;
    KTimer kt(K_DEPTHWISE_BWD, 4.0 * (double)total_o * k * k,
              4.0 * (2.0 * (double)total_o + (overwrite ? 2.0 : 3.0) * (double)n * c * h * wd));
    DwBnBwd bn{dz, bn_mean, bn_var, bn_scales, bn_dmean, bn_dvar};
    DwBnIn in{in_mean, in_var, in_scale, in_bias, in_act};
    if (!dx || !depthwise_backward_lds(x_raw, w, y, nullptr, dx, dw, dbias, s, act, overwrite, 0, &bn, &in))
        dw_bnin_refused("bcnn_hip_depthwise_backward_bn_bnin");
}

size_t bcnn_hip_depthwise_insums_size(int n, int c, int h, int wd, int k, int stride, int pad) {
    if (stride < 1 || k < 1) return 0;
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    return depthwise_lds_in_sums_floats(s);
}

// bcnn_hip_depthwise_backward_bnin (bn_mean == NULL) / _bn_bnin whose kernel also leaves the backward sums of the PRODUCER's
// batch-norm in in_sums; returns the partials per channel (0: not emitted -- dx accumulates, or the buffer is too small)
int bcnn_hip_depthwise_backward_bnin_sums(const float* x_raw, const float* w, const float* y, float* dy, float* dx, float* dw,
                                          float* dbias, int n, int c, int h, int wd, int k, int stride, int pad, int act,
                                          int overwrite, const float* bn_mean, const float* bn_var, const float* bn_scales,
                                          const float* bn_dmean, const float* bn_dvar, const float* in_mean,
                                          const float* in_var, const float* in_scale, const float* in_bias, int in_act,
                                          float* in_sums, size_t in_sums_floats) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const long long total_o = (long long)n * c * s.OH * s.OW;
    if (total_o <= 0) return 0;
    KTimer kt(K_DEPTHWISE_BWD, 4.0 * (double)total_o * k * k,
              4.0 * ((bn_mean ? 2.0 : 3.0) * (double)total_o + (overwrite ? 2.0 : 3.0) * (double)n * c * h * wd));
    DwBnBwd bn{dy, bn_mean, bn_var, bn_scales, bn_dmean, bn_dvar};
    DwBnIn in{in_mean, in_var, in_scale, in_bias, in_act};
    ConvStats st;
    st.partials = in_sums; st.capacity = in_sums_floats; st.splits = 0;
    if (!dx || !depthwise_backward_lds(x_raw, w, y, bn_mean ? nullptr : dy, dx, dw, dbias, s, act, overwrite,
                                       /*write_back=*/bn_mean ? 0 : 1, bn_mean ? &bn : nullptr, &in, in_sums ? &st : nullptr))
        dw_bnin_refused("bcnn_hip_depthwise_backward_bnin_sums");
    return st.splits;
}

void bcnn_hip_depthwise_backward(const float* x, const float* w, const float* y, float* dy, float* dx,
                                 float* dw, float* dbias, int n, int c, int h, int wd, int k, int stride,
                                 int pad, int act, int overwrite) {
    DwShape s{n, c, h, wd, (h + 2 * pad - k) / stride + 1, (wd + 2 * pad - k) / stride + 1, k, stride, pad};
    const int ohow = s.OH * s.OW;
    const long long total_o = (long long)n * c * ohow;
    if (total_o <= 0) return;
    // algorithmic bytes: activation backward (y, dy r/w), bias gradient (dy), dW (x, dy), dX (dy, dx r/w)
    KTimer kt(K_DEPTHWISE_BWD, 4.0 * (double)total_o * k * k,
              4.0 * (((act != BCNN_HIP_ACT_NONE) ? 3.0 : 0.0) * (double)total_o + (double)total_o +
                     (dx ? ((double)n * c * h * wd + (double)total_o) + ((double)total_o + (overwrite ? 1.0 : 2.0) * (double)n * c * h * wd)
                         : 0.0)));
    if (dx && depthwise_lds_ok(s) && act != BCNN_HIP_ACT_PRELU) {
        // one pass: dy *= act'(y) written back, dbias, dW, dX (softplus: its derivative needs exp() -> own pass first)
        int a = act;
        if (!act_bwd_is_cheap(act)) {
            bcnn_hip_activation_backward(y, dy, (size_t)total_o, act, nullptr, nullptr, ohow, c);
            a = BCNN_HIP_ACT_NONE;
        }
        // a refusal (an LDS image above 64 KB on a shape the marching kernels do not take either) leaves dy as it was handed
        // over: depthwise_backward_lds decides before it prepares anything
        if (depthwise_backward_lds(x, w, y, dy, dx, dw, dbias, s, a, overwrite, /*write_back=*/1, nullptr)) return;
        if (a != act) {  // the expensive derivative has been applied above: the generic path continues without it
            act = BCNN_HIP_ACT_NONE;
        }
    }
    activation_backward_grad_bias(y, dy, dbias, n, c, ohow, act);  // one sweep: dy *= act'(y), dbias += sum
    if (!dx) return;  // reference: dW and dX are both skipped when the source has no gradient (:318, :432)
    const int NT = k * k;
    const long long M = (long long)n * ohow;
    const int splits = chan_splits(c, M);
    float* part = reduce_scratch((size_t)c * splits * NT);
    dim3 grid((unsigned)c, (unsigned)splits);
    const unsigned gpr = (unsigned)ceil_div(s.OW, 4);
    constexpr int VR = DW_VR;
    const unsigned rgs = (unsigned)ceil_div(s.OH, VR);
    if (k == 3 && (stride == 1 || stride == 2) && (long long)n * rgs * gpr < 0x7fffffffLL && c <= 65535) {
        Dw3Args a;
        a.s = s; a.groups_per_row = gpr; a.row_groups = rgs; a.gpr_magic = dw_magic(gpr); a.oh_magic = dw_magic(rgs);
        a.total_groups = 0;
        dim3 g2((unsigned)splits, (unsigned)c);
        if (stride == 1) dw3_bwd_weight_kernel<1, VR><<<g2, 256, 0, current_stream()>>>(x, dy, a, splits, part);
        else dw3_bwd_weight_kernel<2, VR><<<g2, 256, 0, current_stream()>>>(x, dy, a, splits, part);
    } else if (k == 3) dw_bwd_weight_kernel<3><<<grid, 256, 0, current_stream()>>>(x, dy, s, splits, part);
    else if (k == 5) dw_bwd_weight_kernel<5><<<grid, 256, 0, current_stream()>>>(x, dy, s, splits, part);
    else {
        dim3 g3((unsigned)c, (unsigned)splits, (unsigned)NT);
        dw_bwd_weight_tap_kernel<<<g3, 256, 0, current_stream()>>>(x, dy, s, splits, part);
    }
    KERNEL_CHECK();
    dw_weight_accumulate_kernel<<<ceil_div(c * NT, 256), 256, 0, current_stream()>>>(part, c, NT, splits, dw);
    KERNEL_CHECK();
    const long long total_i = (long long)n * c * h * wd;
    const unsigned gpr_i = (unsigned)ceil_div(wd, 4);
    constexpr int VRX = DW_VR;
    const unsigned rgs_i = stride == 1 ? (unsigned)ceil_div(h, VRX) : (unsigned)h;
    const long long groups_i = (long long)n * c * rgs_i * gpr_i;
    if (k == 3 && groups_i < 0x7fffffffLL) {
        Dw3DxArgs a;
        a.s = s; a.groups_per_row = gpr_i; a.row_groups = rgs_i; a.gpr_magic = dw_magic(gpr_i); a.h_magic = dw_magic(rgs_i);
        a.total_groups = (unsigned)groups_i;
        a.overwrite = overwrite;
        const unsigned blocks = (unsigned)((groups_i + 255) / 256);
        if (stride == 1) dw3_bwd_data_s1_kernel<VRX><<<blocks, 256, 0, current_stream()>>>(dy, w, dx, a);
        else if (stride == 2) dw3_bwd_data_kernel<2><<<blocks, 256, 0, current_stream()>>>(dy, w, dx, a);
        else dw3_bwd_data_kernel<0><<<blocks, 256, 0, current_stream()>>>(dy, w, dx, a);
    } else {
        dw_bwd_data_kernel<<<stream_grid((size_t)total_i, 256), 256, 0, current_stream()>>>(dy, w, dx, s,
                                                                                         (unsigned)total_i, overwrite);
    }
    KERNEL_CHECK();
}

}  // extern "C"
