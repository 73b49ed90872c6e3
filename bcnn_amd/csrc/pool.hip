// pool.hip -- max pooling (bit-exact indices) and global average pooling. HBM-bound.
//
// Reference semantics: src/layers/bcnn_maxpool_layer.c:145-191 (forward), :258-273 (backward),
// src/layers/bcnn_avgpool_layer.c:82-99, 109-125.
#include <cfloat>

#include "bn_math.h"
#include "common.h"

namespace bcnn_hip {

// One thread per output element; consecutive lanes take consecutive output columns, so the window
// rows they read are contiguous (stride `stride` floats) and the value/index stores are coalesced.
// Window origin (i*stride, j*stride): padding exists only at the bottom/right; out-of-range taps
// read as -FLT_MAX; scan rows outer / cols inner; replace only on `>`: the first maximum wins and a
// NaN never wins. The index is the flat offset into the WHOLE source tensor (int32), as stored by
// the reference -- compared bit-for-bit in tests.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int* __restrict__ idx, int planes, int H, int W,
                                                          int OH, int OW, int size, int stride,
                                                          unsigned total) {
    const unsigned gstride = gridDim.x * blockDim.x;
    for (unsigned o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gstride) {
        const unsigned j = o % (unsigned)OW, t = o / (unsigned)OW;
        const unsigned i = t % (unsigned)OH, plane = t / (unsigned)OH;
        const int base = (int)plane * H * W;
        float best = -FLT_MAX;
        int bi = -1;
        for (int r = 0; r < size; ++r) {
            const int hh = (int)i * stride + r;
            for (int q = 0; q < size; ++q) {
                const int ww = (int)j * stride + q;
                const int si = base + hh * W + ww;
                const bool ok = hh < H && ww < W;
                const float v = ok ? x[si] : -FLT_MAX;
                if (v > best) { best = v; bi = si; }
            }
        }
        y[o] = best;
        idx[o] = bi;
    }
}

// Stride-2 windows of size 2 or 3 (the ResNet / LeNet pools): one thread per TWO consecutive outputs of a row.
// Their windows span source columns 4q .. 4q+3 (+1 for size 3): one aligned 16-byte load (+ one scalar) per
// window row instead of 2 x SIZE scalar loads, one pair of divisions per two outputs. Scan order, strict
// `>` and the flat int32 index are those of the generic kernel (bit-exact).
template <int SIZE>
__global__ __launch_bounds__(256) void maxpool_fwd_s2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int* __restrict__ idx, int H, int W, int OH, int OW,
                                                             unsigned total_pairs) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total_pairs) return;
    const unsigned ppr = (unsigned)(OW + 1) >> 1;  // output pairs per row
    const unsigned rowid = t / ppr, q = t - rowid * ppr;
    const unsigned plane = rowid / (unsigned)OH, i = rowid - plane * (unsigned)OH;
    const int base = (int)plane * H * W;
    const int w0 = (int)q * 4;  // first source column of the pair (W % 4 == 0 => the 16-byte load is in range)
    float best[2] = {-FLT_MAX, -FLT_MAX};
    int bi[2] = {-1, -1};
#pragma unroll
    for (int r = 0; r < SIZE; ++r) {
        const int hh = (int)i * 2 + r;
        if (hh >= H) continue;  // bottom padding: every tap of the row reads as -FLT_MAX and never wins
        const int rb = base + hh * W + w0;
        const float4 v4 = *reinterpret_cast<const float4*>(x + rb);
        const float v[5] = {v4.x, v4.y, v4.z, v4.w, (SIZE == 3 && w0 + 4 < W) ? x[rb + 4] : -FLT_MAX};
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int c = 0; c < SIZE; ++c) {
                const float val = v[2 * o + c];
                if (val > best[o]) { best[o] = val; bi[o] = rb + 2 * o + c; }
            }
    }
    const unsigned out = rowid * (unsigned)OW + q * 2;
    y[out] = best[0];
    idx[out] = bi[0];
    if ((int)(q * 2 + 1) < OW) {
        y[out + 1] = best[1];
        idx[out + 1] = bi[1];
    }
}

// The same windows over act(batch-norm(x)) computed on the fly: the pooling node behind a convolution node with batch-norm
// whose pre-normalisation output x is kept anyway (the ResNet stem). The normalised tensor -- four times the size of the
// pooled one -- is then never written nor read back. Values, scan order and indexes are those of bn apply + the kernel
// above (same bn_one arithmetic per element).
template <int SIZE, int R>  // R consecutive output rows per thread: their 2R + SIZE - 2 source rows are normalised once each
__global__ __launch_bounds__(256) void maxpool_fwd_s2_bn_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int* __restrict__ idx, int C, int H, int W, int OH, int OW,
                                                                unsigned total_items, const float* __restrict__ mean,
                                                                const float* __restrict__ var, const float* __restrict__ scale,
                                                                const float* __restrict__ bias, int act,
                                                                float* __restrict__ raw_at_max) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total_items) return;
    const unsigned ppr = (unsigned)(OW + 1) >> 1;       // output pairs per row
    const unsigned rgs = (unsigned)(OH + R - 1) / R;    // row groups per plane
    const unsigned rowid = t / ppr, q = t - rowid * ppr;
    const unsigned plane = rowid / rgs, i0 = (rowid - plane * rgs) * R;
    const int ch = (int)(plane % (unsigned)C);
    const float m = mean[ch], sc = scale[ch], b = bias[ch];
    const BnDiv rs = bn_divisor(sqrtf(var[ch] + 0.000001f));
    const int base = (int)plane * H * W;
    const int w0 = (int)q * 4;
    float best[R][2], braw[R][2];
    int bi[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r][0] = best[r][1] = -FLT_MAX; bi[r][0] = bi[r][1] = -1; braw[r][0] = braw[r][1] = 0.f; }
    float dummy;
    const bool tail = SIZE == 3 && w0 + 4 < W;
    const bool keep = raw_at_max != nullptr;  // uniform
#pragma unroll
    for (int rr = 0; rr < 2 * R + SIZE - 2; ++rr) {
        const int hh = (int)i0 * 2 + rr;
        if (hh >= H) continue;  // bottom padding: the row never wins
        const int rb = base + hh * W + w0;
        const float4 v4 = *reinterpret_cast<const float4*>(x + rb);
        const float raw[5] = {v4.x, v4.y, v4.z, v4.w, tail ? x[rb + 4] : 0.f};
        const float v[5] = {bn_one(raw[0], m, rs, sc, b, 0, act, &dummy), bn_one(raw[1], m, rs, sc, b, 0, act, &dummy),
                            bn_one(raw[2], m, rs, sc, b, 0, act, &dummy), bn_one(raw[3], m, rs, sc, b, 0, act, &dummy),
                            tail ? bn_one(raw[4], m, rs, sc, b, 0, act, &dummy) : -FLT_MAX};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = rr - 2 * r;  // window row of output row i0 + r; ascending rr == ascending k: the scan order
            if (k < 0 || k >= SIZE) continue;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int c = 0; c < SIZE; ++c) {
                    const float val = v[2 * o + c];
                    if (val > best[r][o]) {
                        best[r][o] = val;
                        bi[r][o] = rb + 2 * o + c;
                        if (keep) braw[r][o] = raw[2 * o + c];
                    }
                }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if ((int)(i0 + r) >= OH) break;
        const unsigned out = (plane * (unsigned)OH + i0 + r) * (unsigned)OW + q * 2;
        y[out] = best[r][0];
        idx[out] = bi[r][0];
        // the pre-normalisation value that won: what the backward pass needs of x at the only places where the pooled
        // gradient lands (bcnn_hip_maxpool_bn_backward)
        if (keep) raw_at_max[out] = braw[r][0];
        if ((int)(q * 2 + 1) < OW) {
            y[out + 1] = best[r][1];
            idx[out + 1] = bi[r][1];
            if (keep) raw_at_max[out + 1] = braw[r][1];
        }
    }
}

// Gather form of `dx[idx[o]] += dy[o]`: one thread per SOURCE element visits the (at most
// ceil(size/stride)^2) outputs whose window covers it, in ascending output order, and adds those
// that selected it. Same per-element addition order as the reference's ascending-o loop, no atomics.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                          float* __restrict__ dx, int H, int W, int OH, int OW,
                                                          int size, int stride, unsigned total) {
    const unsigned gstride = gridDim.x * blockDim.x;
    for (unsigned s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gstride) {
        const unsigned w = s % (unsigned)W, t = s / (unsigned)W;
        const unsigned h = t % (unsigned)H, plane = t / (unsigned)H;
        // outputs i with i*stride <= h <= i*stride + size - 1
        int i0 = ((int)h - size + stride) / stride; if ((int)h - size + 1 <= 0) i0 = 0;
        int j0 = ((int)w - size + stride) / stride; if ((int)w - size + 1 <= 0) j0 = 0;
        int i1 = (int)h / stride; if (i1 > OH - 1) i1 = OH - 1;
        int j1 = (int)w / stride; if (j1 > OW - 1) j1 = OW - 1;
        float acc = dx[s];
        bool hit = false;
        const int obase = (int)plane * OH * OW;
        for (int i = i0; i <= i1; ++i)
            for (int j = j0; j <= j1; ++j) {
                const int o = obase + i * OW + j;
                if (idx[o] == (int)s) { acc += dy[o]; hit = true; }
            }
        if (hit) dx[s] = acc;
    }
}

// Same gather, vectorised: one thread per 4 consecutive source pixels of a row (16-byte read-modify-write of
// dx, one division per 4 elements). The windows covering the 4 pixels are visited
// in ascending output order, so every dx element still sees its additions in the reference's order.
// OVERWRITE: dx is known to hold zeros semantically (the executor proved this node is its only writer and skipped
// the zero fill): start from 0.f instead of loading, store always -- same sums bit for bit, a third of the traffic.
template <bool OVERWRITE>
__global__ __launch_bounds__(256) void maxpool_bwd_vec4_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                               float* __restrict__ dx, int H, int W, int OH, int OW,
                                                               int size, int stride) {
    const int W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;  // (row, group of 4) within the plane
    const int plane = blockIdx.y;
    if (t >= H * W4) return;
    const int h = t / W4, w0 = (t - h * W4) * 4;
    const int s0 = (plane * H + h) * W + w0;
    int i0 = (h - size + stride) / stride; if (h - size + 1 <= 0) i0 = 0;
    int j0 = (w0 - size + stride) / stride; if (w0 - size + 1 <= 0) j0 = 0;
    int i1 = h / stride; if (i1 > OH - 1) i1 = OH - 1;
    int j1 = (w0 + 3) / stride; if (j1 > OW - 1) j1 = OW - 1;
    float4 v = OVERWRITE ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(dx + s0);
    bool hit = OVERWRITE;
    for (int i = i0; i <= i1; ++i) {
        const int o = (plane * OH + i) * OW;
        for (int j = j0; j <= j1; ++j) {
            const unsigned d = (unsigned)(idx[o + j] - s0);
            if (d < 4u) {
                const float g = dy[o + j];
                if (d == 0) v.x += g; else if (d == 1) v.y += g; else if (d == 2) v.z += g; else v.w += g;
                hit = true;
            }
        }
    }
    if (hit) *reinterpret_cast<float4*>(dx + s0) = v;
}

// 3x3 / stride 2 (the ResNet stem pool): at most 2 window rows x 3 window columns cover a group of 4 pixels.
// All six (index, gradient) pairs are loaded up front from clamped addresses -- twelve independent loads in flight
// instead of a load -> compare -> load chain per window -- and applied in the same (row, column) order as above,
// so the sums are bit-identical.
template <bool OVERWRITE>
__global__ __launch_bounds__(256) void maxpool_bwd_vec4_k3s2_kernel(const float* __restrict__ dy,
                                                                    const int* __restrict__ idx,
                                                                    float* __restrict__ dx, int H, int W, int OH,
                                                                    int OW) {
    const int W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int plane = blockIdx.y;
    if (t >= H * W4) return;
    const int h = t / W4, w0 = (t - h * W4) * 4;
    const int s0 = (plane * H + h) * W + w0;
    const int i0 = h >= 2 ? (h - 1) >> 1 : 0, j0 = w0 >= 2 ? (w0 - 1) >> 1 : 0;
    int i1 = h >> 1; if (i1 > OH - 1) i1 = OH - 1;
    int j1 = (w0 + 3) >> 1; if (j1 > OW - 1) j1 = OW - 1;
    int id[2][3];
    float g[2][3];
    bool ok[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int i = i0 + a, j = j0 + b;
            ok[a][b] = i <= i1 && j <= j1;
            const int o = (plane * OH + (i <= i1 ? i : i1)) * OW + (j <= j1 ? j : j1);
            id[a][b] = idx[o];
            g[a][b] = dy[o];
        }
    float4 v = OVERWRITE ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(dx + s0);
    bool hit = OVERWRITE;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const unsigned d = (unsigned)(id[a][b] - s0);
            if (ok[a][b] && d < 4u) {
                const float gg = g[a][b];
                if (d == 0) v.x += gg; else if (d == 1) v.y += gg; else if (d == 2) v.z += gg; else v.w += gg;
                hit = true;
            }
        }
    if (hit) *reinterpret_cast<float4*>(dx + s0) = v;
}

// The same for OW == W / 2 (the "same"-padded stem pool: 112 -> 56): the three window columns of thread k are 2k-1, 2k and
// 2k+1, so the pair (2k, 2k+1) is ONE aligned 8-byte load per row and tensor -- contiguous over the wave -- and column
// 2k-1 is the previous lane's second element (one DPP move; lane 0 of a wave fetches it itself). 4 + 4 memory
// instructions instead of 12 stride-2 gathers; same additions in the same order.
template <bool OVERWRITE>
__global__ __launch_bounds__(256) void maxpool_bwd_vec4_k3s2_pair_kernel(const float* __restrict__ dy,
                                                                         const int* __restrict__ idx,
                                                                         float* __restrict__ dx, int H, int W, int OH,
                                                                         int OW) {
    const int W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int plane = blockIdx.y;
    const bool live = t < H * W4;
    const int tt = live ? t : 0;
    const int h = tt / W4, k = tt - h * W4, w0 = k * 4;
    const int s0 = (plane * H + h) * W + w0;
    const int i0 = h >= 2 ? (h - 1) >> 1 : 0;
    int i1 = h >> 1; if (i1 > OH - 1) i1 = OH - 1;
    const bool need_left = k > 0, fetch_left = need_left && (threadIdx.x & 63) == 0;
    int id[2][3];
    float g[2][3];
    bool ok[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int i = i0 + a;
        ok[a] = live && i <= i1;
        const int o = (plane * OH + (i <= i1 ? i : i1)) * OW + 2 * k;
        const int2 ip = *reinterpret_cast<const int2*>(idx + o);
        const float2 gp = *reinterpret_cast<const float2*>(dy + o);
        const int ie = idx[fetch_left ? o - 1 : o];
        const float ge = dy[fetch_left ? o - 1 : o];
        id[a][0] = __builtin_amdgcn_update_dpp(ie, ip.y, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        g[a][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ge), __builtin_bit_cast(int, gp.y),
                                                                        0x138, 0xf, 0xf, false));
        id[a][1] = ip.x; g[a][1] = gp.x;
        id[a][2] = ip.y; g[a][2] = gp.y;
    }
    if (!live) return;
    float4 v = OVERWRITE ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(dx + s0);
    bool hit = OVERWRITE;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const unsigned d = (unsigned)(id[a][b] - s0);
            if (ok[a] && (b > 0 || need_left) && d < 4u) {
                const float gg = g[a][b];
                if (d == 0) v.x += gg; else if (d == 1) v.y += gg; else if (d == 2) v.z += gg; else v.w += gg;
                hit = true;
            }
        }
    if (hit) *reinterpret_cast<float4*>(dx + s0) = v;
}

// The pair kernel for a pooling node behind a convolution node with batch-norm (the ResNet stem), with that node's
// batch-norm backward apply step (bcnn_batchnorm_layer.c:292-296 behind the activation backward) on the four gradient
// values while they are in registers: dx[s] = bn_bwd(sum of the pooled gradients that selected s) is written straight into
// the convolution node's output-gradient tensor, which is then never written and re-read in its pooled-gradient form. The
// coefficients (dmean, dvar) come from sums taken over the POOLED tensors (bcnn_hip_maxpool_bn_backward below).
__global__ __launch_bounds__(256) void maxpool_bwd_pair_bn_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                                  const float* __restrict__ raw, float* __restrict__ dx,
                                                                  int C, int H, int W, int OH, int OW,
                                                                  const float4* __restrict__ consts, unsigned w4_magic,
                                                                  float fM, float rfM, int act) {
    const int W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int plane = blockIdx.y;
    const bool live = t < H * W4;
    const int tt = live ? t : 0;
    // tt / W4 by multiply-high with ceil(2^32 / W4) (exact: tt * W4 < 2^32, checked by the launcher)
    const int h = W4 > 1 ? (int)__umulhi((unsigned)tt, w4_magic) : tt, k = tt - h * W4, w0 = k * 4;
    const int s0 = (plane * H + h) * W + w0;
    const int i0 = h >= 2 ? (h - 1) >> 1 : 0;
    int i1 = h >> 1; if (i1 > OH - 1) i1 = OH - 1;
    const bool need_left = k > 0, fetch_left = need_left && (threadIdx.x & 63) == 0;
    const float4 xv = *reinterpret_cast<const float4*>(raw + s0);
    int id[2][3];
    float g[2][3];
    bool ok[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int i = i0 + a;
        ok[a] = live && i <= i1;
        const int o = (plane * OH + (i <= i1 ? i : i1)) * OW + 2 * k;
        const int2 ip = *reinterpret_cast<const int2*>(idx + o);
        const float2 gp = *reinterpret_cast<const float2*>(dy + o);
        const int ie = idx[fetch_left ? o - 1 : o];
        const float ge = dy[fetch_left ? o - 1 : o];
        id[a][0] = __builtin_amdgcn_update_dpp(ie, ip.y, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        g[a][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ge), __builtin_bit_cast(int, gp.y),
                                                                        0x138, 0xf, 0xf, false));
        id[a][1] = ip.x; g[a][1] = gp.x;
        id[a][2] = ip.y; g[a][2] = gp.y;
    }
    if (!live) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const unsigned d = (unsigned)(id[a][b] - s0);
            if (ok[a] && (b > 0 || need_left) && d < 4u) {
                const float gg = g[a][b];
                if (d == 0) v.x += gg; else if (d == 1) v.y += gg; else if (d == 2) v.z += gg; else v.w += gg;
            }
        }
    // per-channel constants: the table bn_bwd_finalize_kernel left (wave-uniform: scalar loads). Four IEEE divisions and two
    // square roots per thread -- for four elements -- had made this kernel vector-ALU bound (profiles/r04_sq_step_resnet18.txt)
    const int ch = plane % C;
    const float4 k0 = consts[3 * ch], k1 = consts[3 * ch + 1], k2 = consts[3 * ch + 2];
    const float m = k0.x, sc = k0.y, bb = k1.z, dmm = k1.w, dv = k2.x;
    const BnDiv rs{k0.z, k0.w}, rs_fwd{k1.x, k1.y}, fMd{fM, rfM};
    float dummy;
    float4 o;
    o.x = bn_bwd_one(v.x, bn_one(xv.x, m, rs_fwd, sc, bb, 0, act, &dummy), xv.x, m, rs, sc, dmm, dv, fMd, act);
    o.y = bn_bwd_one(v.y, bn_one(xv.y, m, rs_fwd, sc, bb, 0, act, &dummy), xv.y, m, rs, sc, dmm, dv, fMd, act);
    o.z = bn_bwd_one(v.z, bn_one(xv.z, m, rs_fwd, sc, bb, 0, act, &dummy), xv.z, m, rs, sc, dmm, dv, fMd, act);
    o.w = bn_bwd_one(v.w, bn_one(xv.w, m, rs_fwd, sc, bb, 0, act, &dummy), xv.w, m, rs, sc, dmm, dv, fMd, act);
    *reinterpret_cast<float4*>(dx + s0) = o;
}

// Global average pooling: one wave64 per (n,c) plane, shuffle reduction, then / (H*W).
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int planes, int HW) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < planes; p += nwaves) {
        const float* src = x + (long long)p * HW;
        float s = 0.f;
        for (int i = lane; i < HW; i += 64) s += src[i];
        s = wave_sum(s);
        if (lane == 0) y[p] = s / (float)HW;
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                          int HW, unsigned total) {
    const unsigned gstride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const unsigned p = i / (unsigned)HW;
        dx[i] += dy[p] / (float)HW;  // dst.grad / (h*w), int divisor promoted (bcnn_avgpool_layer.c:118-120)
    }
}

// batchnorm.hip: sums + finalize of a batch-norm backward over (dy, x) with the forward output recomputed from x
void batchnorm_backward_sums(const float* dy, const float* y, int act, const float* scales, float* dscales, float* dbias,
                             const float* saved_mean, const float* saved_var, float* dmean, float* dvar,
                             const float* workspace, int n, int c, int hw, const float* fwd_bias, const float* res,
                             unsigned res_count, float4* consts = nullptr, float consts_fM = 0.f);
float4* bn_consts_scratch(int channels, bool required);  // batchnorm.hip: the per-channel constants its backward finalize kernel leaves

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_maxpool_forward(const float* x, float* y, int* indexes, int n, int c, int h, int w, int out_h,
                              int out_w, int size, int stride) {
    const long long total = (long long)n * c * out_h * out_w;
    if (!total) return;
    // two outputs per thread need both windows inside the 16-byte load: out_w = ceil(w / 2) covers it when w % 4 == 0
    if (stride == 2 && (size == 2 || size == 3) && (w & 3) == 0 && out_w * 2 <= w + 1 &&
        (long long)n * c * h * w < 0x7fffffffLL && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long long pairs = (long long)n * c * out_h * ((out_w + 1) / 2);
        const unsigned blocks = (unsigned)((pairs + 255) / 256);
        if (size == 2) maxpool_fwd_s2_kernel<2><<<blocks, 256, 0, current_stream()>>>(x, y, indexes, h, w, out_h, out_w, (unsigned)pairs);
        else maxpool_fwd_s2_kernel<3><<<blocks, 256, 0, current_stream()>>>(x, y, indexes, h, w, out_h, out_w, (unsigned)pairs);
        KERNEL_CHECK();
        return;
    }
    maxpool_fwd_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(
        x, y, indexes, n * c, h, w, out_h, out_w, size, stride, (unsigned)total);
    KERNEL_CHECK();
}

int bcnn_hip_maxpool_bn_fusable(int n, int c, int h, int w, int out_h, int out_w, int size, int stride, int act,
                                 const float* x) {
    return stride == 2 && (size == 2 || size == 3) && (w & 3) == 0 && out_w * 2 <= w + 1 &&
           (long long)n * c * h * w < 0x7fffffffLL && x && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && act_is_cheap(act) &&
           act != BCNN_HIP_ACT_PRELU;
}

void bcnn_hip_maxpool_forward_bn_keep(const float* x, float* y, int* indexes, int n, int c, int h, int w, int out_h, int out_w,
                                      int size, int stride, const float* scales, const float* bias, const float* mean,
                                      const float* var, int act, float* raw_at_max) {
    const long long total = (long long)n * c * out_h * out_w;
    if (!total) return;
    if (!bcnn_hip_maxpool_bn_fusable(n, c, h, w, out_h, out_w, size, stride, act, x)) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_maxpool_forward_bn: not fusable (ask bcnn_hip_maxpool_bn_fusable)\n");
        exit(1);
    }
    KTimer kt(K_POOL, 0.0, 4.0 * ((double)n * c * h * w + 2.0 * (double)total));
    trace_kernel("maxpool_fwd_s2_bn_kernel");
    constexpr int R = 4;
    const long long items = (long long)n * c * ((out_h + R - 1) / R) * ((out_w + 1) / 2);
    const unsigned blocks = (unsigned)((items + 255) / 256);
    if (size == 2)
        maxpool_fwd_s2_bn_kernel<2, R><<<blocks, 256, 0, current_stream()>>>(x, y, indexes, c, h, w, out_h, out_w,
                                                                             (unsigned)items, mean, var, scales, bias, act,
                                                                             raw_at_max);
    else
        maxpool_fwd_s2_bn_kernel<3, R><<<blocks, 256, 0, current_stream()>>>(x, y, indexes, c, h, w, out_h, out_w,
                                                                             (unsigned)items, mean, var, scales, bias, act,
                                                                             raw_at_max);
    KERNEL_CHECK();
}

void bcnn_hip_maxpool_forward_bn(const float* x, float* y, int* indexes, int n, int c, int h, int w, int out_h, int out_w,
                                 int size, int stride, const float* scales, const float* bias, const float* mean,
                                 const float* var, int act) {
    bcnn_hip_maxpool_forward_bn_keep(x, y, indexes, n, c, h, w, out_h, out_w, size, stride, scales, bias, mean, var, act, nullptr);
}

int bcnn_hip_maxpool_bn_backward_fusable(int n, int c, int h, int w, int out_h, int out_w, int size, int stride, int act,
                                         const float* raw, const float* dpool, const int* indexes, const float* dx) {
    const uintptr_t p16 = reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(dx);
    const uintptr_t p8 = reinterpret_cast<uintptr_t>(dpool) | reinterpret_cast<uintptr_t>(indexes);
    return size == 3 && stride == 2 && (w & 3) == 0 && out_w * 2 == w && out_h * out_w > 0 &&
           (long long)n * c * h * w < 0x7fffffffLL && (long long)n * c <= 65535 && raw && dx && (p16 & 15) == 0 && (p8 & 7) == 0 &&
           (long long)h * (w / 4) * (w / 4) < (1LL << 32) &&  // the kernel's t / (W / 4) by multiply-high is exact only while t * (W / 4) < 2^32
           act_is_cheap(act) && act_bwd_is_cheap(act) && act != BCNN_HIP_ACT_PRELU;
}

void bcnn_hip_maxpool_bn_backward(const float* dpool, const int* indexes, const float* raw_at_max, const float* raw, float* dx,
                                  int n, int c, int h, int w, int out_h, int out_w, int size, int stride, const float* scales,
                                  float* dscales, const float* bias, float* dbias, const float* mean, const float* var,
                                  float* dmean, float* dvar, int act) {
    const long long total = (long long)n * c * h * w, ptotal = (long long)n * c * out_h * out_w;
    if (!total) return;
    if (!bcnn_hip_maxpool_bn_backward_fusable(n, c, h, w, out_h, out_w, size, stride, act, raw, dpool, indexes, dx) || !raw_at_max) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_maxpool_bn_backward: not fusable (ask bcnn_hip_maxpool_bn_backward_fusable)\n");
        exit(1);
    }
    float4* consts = bn_consts_scratch(c, true);
    const float fM = (float)((long long)n * h * w);  // the un-pooled element count divides dmean and the dvar term
    // S1 = sum g act'(y), S2 = sum g act'(y) (x - mean) of the batch-norm backward: the gradient of the un-pooled tensor is
    // zero except where a window's maximum sits, and there it is the sum of the pooled gradients that selected the place --
    // so both sums are sums over the POOLED gradient against the pre-normalisation values that won (a quarter of the data;
    // the divisor of dmean / M and of the dvar term stays the un-pooled element count)
    {
        KTimer kt(K_BN_BWD, 0.0, 4.0 * 2.0 * (double)ptotal);
        batchnorm_backward_sums(dpool, nullptr, act, scales, dscales, dbias, mean, var, dmean, dvar, raw_at_max, n, c,
                                out_h * out_w, act != BCNN_HIP_ACT_NONE ? bias : nullptr, nullptr, 0u, consts, fM);
    }
    KTimer kt(K_POOL, 0.0, 4.0 * (2.0 * (double)total + 2.0 * (double)ptotal));
    dim3 grid((unsigned)ceil_div(h * (w / 4), 256), (unsigned)(n * c));
    const unsigned w4 = (unsigned)(w / 4);
    trace_kernel("maxpool_bwd_pair_bn_kernel");
    maxpool_bwd_pair_bn_kernel<<<grid, 256, 0, current_stream()>>>(dpool, indexes, raw, dx, c, h, w, out_h, out_w, consts,
                                                                   w4 > 1 ? (unsigned)((0x100000000ULL + w4 - 1) / w4) : 0u, fM,
                                                                   1.0f / fM, act);
    KERNEL_CHECK();
}

void bcnn_hip_maxpool_backward(const float* dy, const int* indexes, float* dx, int n, int c, int h, int w,
                               int out_h, int out_w, int size, int stride, int overwrite) {
    const long long total = (long long)n * c * h * w;
    if (!total) return;
    if (!(out_h * out_w)) {
        if (overwrite) bcnn_hip_fill_f32(dx, (size_t)total, 0.f);
        return;
    }
    if ((w & 3) == 0 && total < 0x7fffffffLL && (long long)n * c <= 65535 &&
        (reinterpret_cast<uintptr_t>(dx) & 15) == 0) {
        dim3 grid((unsigned)ceil_div(h * (w / 4), 256), (unsigned)(n * c));
        const bool pairs = size == 3 && stride == 2 && out_w * 2 == w &&
                           ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(indexes)) & 7) == 0;
        if (pairs) {
            if (overwrite) maxpool_bwd_vec4_k3s2_pair_kernel<true><<<grid, 256, 0, current_stream()>>>(dy, indexes, dx, h, w, out_h, out_w);
            else maxpool_bwd_vec4_k3s2_pair_kernel<false><<<grid, 256, 0, current_stream()>>>(dy, indexes, dx, h, w, out_h, out_w);
        } else if (size == 3 && stride == 2) {
            if (overwrite) maxpool_bwd_vec4_k3s2_kernel<true><<<grid, 256, 0, current_stream()>>>(dy, indexes, dx, h, w, out_h, out_w);
            else maxpool_bwd_vec4_k3s2_kernel<false><<<grid, 256, 0, current_stream()>>>(dy, indexes, dx, h, w, out_h, out_w);
        } else if (overwrite)
            maxpool_bwd_vec4_kernel<true><<<grid, 256, 0, current_stream()>>>(dy, indexes, dx, h, w, out_h, out_w, size, stride);
        else
            maxpool_bwd_vec4_kernel<false><<<grid, 256, 0, current_stream()>>>(dy, indexes, dx, h, w, out_h, out_w, size, stride);
        KERNEL_CHECK();
        return;
    }
    if (overwrite) bcnn_hip_fill_f32(dx, (size_t)total, 0.f);
    maxpool_bwd_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(
        dy, indexes, dx, h, w, out_h, out_w, size, stride, (unsigned)total);
    KERNEL_CHECK();
}

void bcnn_hip_avgpool_forward(const float* x, float* y, int n, int c, int h, int w) {
    const int planes = n * c;
    if (!planes) return;
    const int grid = stream_grid((size_t)planes * 64, 256);
    avgpool_fwd_kernel<<<grid, 256, 0, current_stream()>>>(x, y, planes, h * w);
    KERNEL_CHECK();
}

void bcnn_hip_avgpool_backward(const float* dy, float* dx, int n, int c, int h, int w) {
    const long long total = (long long)n * c * h * w;
    if (!total) return;
    avgpool_bwd_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(dy, dx, h * w, (unsigned)total);
    KERNEL_CHECK();
}

}  // extern "C"
