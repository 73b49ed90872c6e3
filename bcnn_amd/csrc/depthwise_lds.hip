// depthwise_lds.hip -- 3x3 depthwise convolution (pad 1, stride 1 or 2) staged through LDS, HBM-bound.
//
// Reference semantics: src/layers/bcnn_depthwise_conv_layer.c:165-293 (forward), :295-547 (backward); the stand-alone
// batch-norm that follows a depthwise layer in MobileNet: src/layers/bcnn_batchnorm_layer.c:196-242, :292-296.
//
// Why a second set of kernels: the register-window kernels of depthwise.hip fetch every thread's 3 x 6 window straight
// from global memory. On 112 x 112 planes that reaches 3.6 TB/s, on 14 x 14 planes 2.3 TB/s (rows are 56 bytes: no
// 16-byte accesses, eight cache lines per load instruction), and the backward pass needs three sweeps (activation
// backward + bias gradient, weight gradient, data gradient). Here a workgroup owns a TILE -- a band of rows of one plane,
// or a run of whole consecutive planes when planes are small -- whose input is one CONTIGUOUS piece of the NCHW tensor:
//   * it is fetched with 16-byte coalesced loads and scattered into a zero-padded image in LDS (pad column / pad rows
//     hold the zeros the reference's "skip the tap" amounts to), so the compute phase has no bounds tests;
//   * threads walk the image with lanes along the row (conflict-free ds_read_b32), same tap order and the same separate
//     multiply / add roundings as the reference loops;
//   * results are collected in LDS and leave as one contiguous 16-byte coalesced piece.
// Forward also emits the per-channel sum / sum of squares of what it stores (the statistics of a following batch-norm
// node); backward does activation backward, bias gradient, weight gradient and data gradient in ONE pass over
// (dy, y, x) -> dx, and with the coefficients of a following batch-norm node it applies that node's backward to the
// incoming gradient on the fly (bcnn_batchnorm_layer.c:292-296), which removes the batch-norm apply sweep as well.
// All reductions are two-level with a fixed order (one partial per tile and channel, combined in double).
#include "bn_math.h"
#include "chan_reduce.h"
#include "depthwise.h"
#include "lds_dma.h"

#include <algorithm>

namespace bcnn_hip {

namespace {

#ifndef DWL_ABL
#define DWL_ABL 0  // ablation bits for tools/exp (wrong results): 1 no weight gradient, 2 no data gradient, 4 no batch-norm math
#endif
#ifndef DWL_TILE
#define DWL_TILE 4096
#endif
#ifndef DWL_IMAGE
#define DWL_IMAGE 3600
#endif
constexpr int kTileFloats = DWL_TILE;  // most a tile reads per stream: 4 x 16 bytes per thread in flight
constexpr int kMaxQ = kTileFloats / 4 / 256;
constexpr int kImageFloats = DWL_IMAGE;  // padded LDS image of a multi-plane tile (x and g each)
constexpr int kSlack = 4;          // LDS rows behind an image that ragged row groups may read (values discarded)
constexpr int kConst = 24;         // floats per plane in the constants table: 9 taps, bias | mean, rs, scale, dmean/M, dvar,
                                   // 1/rs (14); 16..20: mean, rs, scale, bias, 1/rs of the batch-norm applied to the INPUT

struct DwlGeom {
    int P;    // planes per tile: 1, or a multiple of 4 (whole planes)
    int BR;   // output rows per band (P > 1: all of them)
    int NB;   // bands per plane
    int PWX, RPX, rows_x;  // x image: pitch, LDS rows per plane slot (P > 1), rows allocated
    int PWG, RPG, rows_g;  // g image (backward only)
    int stage_floats;      // forward: outputs of a tile; backward: the dx rows a tile owns
};

// n / d without a hardware division and without a branch: multiply-high by ceil(2^32 / d) (exact while n * d < 2^32);
// d == 1 has no 32-bit magic, so it is encoded as magic 0 + an all-ones mask that passes n through
struct DwlDiv {
    unsigned magic, mask;
};
inline DwlDiv dwl_magic(unsigned d) {
    DwlDiv v;
    v.magic = d > 1 ? (unsigned)((0x100000000ULL + d - 1) / d) : 0u;
    v.mask = d > 1 ? 0u : 0xffffffffu;
    return v;
}
#ifdef __HIPCC__
__device__ __forceinline__ unsigned dwl_div(unsigned n, const DwlDiv& d) { return __umulhi(n, d.magic) | (n & d.mask); }
#endif

inline DwlGeom dwl_plan(const DwShape& s) {
    DwlGeom g;
    const int S = s.stride, plane = s.H * s.W;
    if (plane > kTileFloats) {
        g.P = 1;
        const int ir = kTileFloats / s.W;  // input rows a band may stage
        int br = (ir - 3) / S + 1;
        if (br < 1) br = 1;
        if (br > s.OH) br = s.OH;
        g.NB = (s.OH + br - 1) / br;
        g.BR = (s.OH + g.NB - 1) / g.NB;
        g.NB = (s.OH + g.BR - 1) / g.BR;
    } else {
        g.BR = s.OH;
        g.NB = 1;
        g.P = 1;
        if (plane < kTileFloats / 4) {  // several whole planes per tile, sized by their PADDED image (small planes pad a lot)
            g.P = (kImageFloats / ((s.H + 1) * (s.W + 4))) & ~3;
            if (g.P < 4) g.P = 4;
            while (g.P > 4 && g.P * plane > kTileFloats) g.P -= 4;
        }
    }
    const int ir = (g.BR - 1) * S + 3;
    g.PWX = s.W + 4;
    g.PWG = s.OW + 4;
    if (g.P > 1) {
        g.RPX = s.H + 1;
        g.rows_x = g.P * g.RPX + 1 + kSlack;
        g.RPG = s.OH + 1;
        g.rows_g = g.P * g.RPG + 1 + kSlack;
    } else {
        g.RPX = 0;
        g.rows_x = ir + kSlack;
        g.RPG = 0;
        g.rows_g = g.BR + 2 + kSlack;
    }
    g.stage_floats = 0;
    return g;
}

// the padded LDS image of a tile's contiguous global piece: element e of the piece (row-major, `Wd` wide, `rpp` rows per
// plane) lands at LDS row j * RP + lr0 + row, column 4 + col
struct DwlImg {
    int Wd, PW, RP, rpp, lr0;
    DwlDiv wd_magic, rpp_magic;
};
__device__ __forceinline__ int dwl_slot(const DwlImg& m, int e, bool multi, int& j, int& row) {
    const int gr = (int)dwl_div((unsigned)e, m.wd_magic), col = e - gr * m.Wd;
    j = 0;
    row = gr;
    if (multi) {
        j = (int)dwl_div((unsigned)gr, m.rpp_magic);
        row = gr - j * m.rpp;
    }
    return (j * m.RP + m.lr0 + row) * m.PW + 4 + col;
}

// sum over the 16 lanes of a DPP row, valid in every lane of the row (the first four steps of wave_sum_dpp)
__device__ __forceinline__ float row_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

// the slot of the element after the one at (slot, j, row, col): rows that are not a multiple of 16 bytes are scattered
// element by element, and walking beats dividing four times per 16 bytes
__device__ __forceinline__ void dwl_next(const DwlImg& m, bool multi, int& slot, int& j, int& row, int& col) {
    ++slot;
    if (++col == m.Wd) {
        col = 0;
        slot += m.PW - m.Wd;
        if (++row == m.rpp && multi) {
            row = 0;
            ++j;
            slot += (m.RP - m.rpp) * m.PW;
        }
    }
}

__device__ __forceinline__ void dwl_zero(float* img, int floats) {  // floats % 4 == 0, img 16-byte aligned
    float4* p = reinterpret_cast<float4*>(img);
    for (int i = threadIdx.x; i < floats / 4; i += 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int NQ>
__device__ __forceinline__ void dwl_fetch(const float* g, int count, bool vec, float4 (&v)[NQ]) {
    if (!vec) return;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = threadIdx.x + q * 256;
        if (i * 4 < count) v[q] = reinterpret_cast<const float4*>(g)[i];
    }
}

// copy of a contiguous piece into its image (after the zero fill and a barrier)
template <int NQ>
__device__ __forceinline__ void dwl_scatter(const float* g, int count, bool vec, bool rowvec, bool multi, const DwlImg& m,
                                            const float4 (&v)[NQ], float* img) {
    int j, row;
    if (vec) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (threadIdx.x + q * 256) * 4;
            if (e >= count) continue;
            if (rowvec) {
                *reinterpret_cast<float4*>(img + dwl_slot(m, e, multi, j, row)) = v[q];
            } else {
                int slot = dwl_slot(m, e, multi, j, row);
                int col = slot - ((j * m.RP + m.lr0 + row) * m.PW + 4);
                img[slot] = v[q].x;
                dwl_next(m, multi, slot, j, row, col);
                img[slot] = v[q].y;
                dwl_next(m, multi, slot, j, row, col);
                img[slot] = v[q].z;
                dwl_next(m, multi, slot, j, row, col);
                img[slot] = v[q].w;
            }
        }
    } else {
        for (int e = threadIdx.x; e < count; e += 256) img[dwl_slot(m, e, multi, j, row)] = g[e];
    }
}

// slots 16..20 of a plane's constants: mean, sqrt(var + 1e-6), scale, bias, 1 / sqrt(var + 1e-6) of the input's batch-norm
__device__ __forceinline__ float dwl_bnin_const(const DwBnIn& in, int c, int t) {
    if (t == 16) return in.mean[c];
    if (t == 18) return in.scale[c];
    if (t == 19) return in.bias[c];
    const float rs = sqrtf(in.var[c] + 0.000001f);
    return t == 17 ? rs : __fdiv_rn(1.0f, rs);
}

// the same with act(bn(.)) of the producing convolution node applied to every element on its way into the image (the pad
// cells keep their zeros: padding applies to the normalised tensor)
// mbits (optional): bit i of it = the activation's derivative at the thread's i-th element is 1 (not 0) -- element
// (tid + q * 256) * 4 + k <-> bit 4 q + k on the vector path, element tid + 256 i <-> bit i otherwise; meaningful for
// activations whose derivative is 0 or 1 (none, ReLU)
template <int NQ>
__device__ __forceinline__ void dwl_scatter_bn(const float* g, int count, bool vec, bool rowvec, bool multi, const DwlImg& m,
                                               const float4 (&v)[NQ], float* img, const float* wl, int act,
                                               unsigned* mbits = nullptr) {
    int j, row;
    float dummy;
    unsigned bits = 0u;
    int bit = 0;
    auto one = [&](float x, int jj) -> float {
        const float* k = wl + jj * kConst + 16;
        const BnDiv rs{k[1], k[4]};
        const float y = bn_one(x, k[0], rs, k[2], k[3], 0, act, &dummy);
        if (mbits) bits |= (act_bwd_cheap(y, act, 0.f) != 0.f ? 1u : 0u) << bit;
        return y;
    };
    if (vec) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (threadIdx.x + q * 256) * 4;
            if (e >= count) continue;
            bit = 4 * q;
            if (rowvec) {
                const int slot = dwl_slot(m, e, multi, j, row);
                float4 o;
                o.x = one(v[q].x, j); ++bit;
                o.y = one(v[q].y, j); ++bit;
                o.z = one(v[q].z, j); ++bit;
                o.w = one(v[q].w, j);
                *reinterpret_cast<float4*>(img + slot) = o;
            } else {
                int slot = dwl_slot(m, e, multi, j, row);
                int col = slot - ((j * m.RP + m.lr0 + row) * m.PW + 4);
                img[slot] = one(v[q].x, j); ++bit;
                dwl_next(m, multi, slot, j, row, col);
                img[slot] = one(v[q].y, j); ++bit;
                dwl_next(m, multi, slot, j, row, col);
                img[slot] = one(v[q].z, j); ++bit;
                dwl_next(m, multi, slot, j, row, col);
                img[slot] = one(v[q].w, j);
            }
        }
    } else {
        for (int e = threadIdx.x; e < count; e += 256, ++bit) {
            const int slot = dwl_slot(m, e, multi, j, row);
            img[slot] = one(g[e], j);
        }
    }
    if (mbits) *mbits = bits;
}

// contiguous LDS piece -> contiguous global piece
__device__ __forceinline__ void dwl_copy_out(const float* stage, float* dst, int count) {
    if (((reinterpret_cast<uintptr_t>(dst) & 15) == 0) && (count & 3) == 0) {
        for (int i = threadIdx.x; i < count / 4; i += 256)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(stage)[i];
    } else {
        for (int i = threadIdx.x; i < count; i += 256) dst[i] = stage[i];
    }
}

struct DwlFwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    float* stats;  // NULL: none
    DwBnIn in;     // BNIN kernels: the producer's batch-norm, applied while staging x
    int C, H, W, OH, OW, planes, act, splits, RG;
    DwlGeom g;
    DwlDiv w_magic, h_magic, ow_magic, rg_magic;
    int x_floats;  // LDS floats of the x image (multiple of 4)
};

template <int S, int VR, bool BNIN>
__global__ __launch_bounds__(256) void dwl_fwd_kernel(const DwlFwdArgs a) {
    extern __shared__ float4 dwl_smem[];
    __shared__ float red[4][2];
    float* xl = reinterpret_cast<float*>(dwl_smem);
    float* out = xl + a.x_floats;
    float* wl = out + a.g.stage_floats;
    const int tid = threadIdx.x;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int pb = tile / a.g.NB, band = tile - pb * a.g.NB;
    const int p0 = pb * a.g.P;
    const int Pe = min(a.g.P, a.planes - p0);
    const bool multi = a.g.P > 1;
    const int oh0 = band * a.g.BR;
    const int BRt = min(a.g.BR, a.OH - oh0);
    const int ihb = oh0 * S - 1;  // image row of LDS row 0
    const int r0 = max(ihb, 0), r1 = min(ihb + (BRt - 1) * S + 3, a.H);
    const float* gx = a.x + ((size_t)p0 * a.H + r0) * a.W;
    const int count = Pe * (r1 - r0) * a.W;
    const bool rowvec = (a.W & 3) == 0;
    const bool vec = (rowvec || multi) && ((reinterpret_cast<uintptr_t>(gx) & 15) == 0) && (count & 3) == 0;
    DwlImg m;
    m.Wd = a.W; m.PW = a.g.PWX; m.RP = a.g.RPX; m.rpp = r1 - r0; m.lr0 = r0 - ihb; m.wd_magic = a.w_magic; m.rpp_magic = a.h_magic;
    float4 xv[kMaxQ];
    dwl_fetch<kMaxQ>(gx, count, vec, xv);
    dwl_zero(xl, a.x_floats);
    for (int i = tid; i < Pe * kConst; i += 256) {
        const int j = i / kConst, t = i - j * kConst, c = (p0 + j) % a.C;
        float v = t < 9 ? a.w[c * 9 + t] : (t == 9 ? a.bias[c] : 0.f);
        if (BNIN && t >= 16 && t <= 20) v = dwl_bnin_const(a.in, c, t);
        wl[i] = v;
    }
    __syncthreads();
    if (BNIN) dwl_scatter_bn<kMaxQ>(gx, count, vec, rowvec, multi, m, xv, xl, wl, a.in.act);
    else dwl_scatter<kMaxQ>(gx, count, vec, rowvec, multi, m, xv, xl);
    __syncthreads();

    constexpr int NR = (VR - 1) * S + 3;
    const int items = Pe * a.RG * a.OW;
    for (int item = tid; item < items; item += 256) {
        const int t = (int)dwl_div((unsigned)item, a.ow_magic), ow = item - t * a.OW;
        const int j = (int)dwl_div((unsigned)t, a.rg_magic), rl0 = (t - j * a.RG) * VR;
        const float* wp = wl + j * kConst;
        float wv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wv[i] = wp[i];
        const float b = wp[9];
        const float* ip = xl + (j * a.g.RPX + rl0 * S) * a.g.PWX + 3 + ow * S;
        float acc[VR];
#pragma unroll
        for (int r = 0; r < VR; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float x0 = ip[i * a.g.PWX], x1 = ip[i * a.g.PWX + 1], x2 = ip[i * a.g.PWX + 2];
#pragma unroll
            for (int r = 0; r < VR; ++r) {
                const int kh = i - r * S;  // ascending i == ascending kh per output row: the reference's tap order
                if (kh < 0 || kh > 2) continue;
                acc[r] = __fadd_rn(acc[r], __fmul_rn(wv[kh * 3 + 0], x0));
                acc[r] = __fadd_rn(acc[r], __fmul_rn(wv[kh * 3 + 1], x1));
                acc[r] = __fadd_rn(acc[r], __fmul_rn(wv[kh * 3 + 2], x2));
            }
        }
        float* op = out + (j * BRt + rl0) * a.OW + ow;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            if (rl0 + r >= BRt) break;
            float v = acc[r];
            if (b != 0.0f && b != 1.0f) v += b;  // bcnn_add_bias quirk
            op[r * a.OW] = act_fwd_cheap(v, a.act, 0.f);
        }
    }
    __syncthreads();
    const int per_plane = BRt * a.OW;
    dwl_copy_out(out, a.y + ((size_t)p0 * a.OH + oh0) * a.OW, Pe * per_plane);
    if (!a.stats) return;
    const int lane = tid & 63, wid = tid >> 6;
    if (!multi) {
        float s1 = 0.f, s2 = 0.f;
        for (int i = tid; i < per_plane; i += 256) {
            const float v = out[i];
            s1 += v;
            s2 += v * v;
        }
        s1 = wave_sum_dpp(s1);
        s2 = wave_sum_dpp(s2);
        if (lane == 63) { red[wid][0] = s1; red[wid][1] = s2; }
        __syncthreads();
        if (tid < 2) {
            const int n = p0 / a.C, c = p0 - n * a.C;
            a.stats[((size_t)c * a.splits + (size_t)n * a.g.NB + band) * 2 + tid] =
                (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        }
    } else {
        for (int j = wid; j < Pe; j += 4) {
            float s1 = 0.f, s2 = 0.f;
            for (int i = lane; i < per_plane; i += 64) {
                const float v = out[j * per_plane + i];
                s1 += v;
                s2 += v * v;
            }
            s1 = wave_sum_dpp(s1);
            s2 = wave_sum_dpp(s2);
            if (lane == 63) {
                const int p = p0 + j, n = p / a.C, c = p - n * a.C;
                float* dst = a.stats + ((size_t)c * a.splits + n) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
    }
}

// ================================================================================================
// backward
// ================================================================================================
struct DwlBwdArgs {
    const float* x;
    const float* w;
    const float* y;
    float* dy;        // read (no batch-norm), written back when write_back
    float* dx;
    float* partials;  // [C][splits][12]: nine taps, bias sum
    float* in_sums;   // BNIN kernels, optional: [C][splits][2] backward sums of the producer's batch-norm (see the kernel's end)
    DwBnBwd bn;
    DwBnIn in;        // BNIN kernels: the producer's batch-norm, applied while staging x
    float fM, rfM;    // N * OH * OW as float (batch-norm) and its correctly rounded reciprocal
    int C, H, W, OH, OW, planes, act, overwrite, write_back, splits, RG;
    DwlGeom g;
    DwlDiv w_magic, h_magic, ow_magic, oh_magic, rg_magic, rgx_magic, hw2_magic, hw_magic;
    int RGX;          // stride 1: ceil(own rows / 4) of a full band; stride 2: unused
    int x_floats, g_floats;
};

constexpr int kPart = 12;

template <int S, int VR, bool BN, bool BNIN>
__global__ __launch_bounds__(256) void dwl_bwd_kernel(const DwlBwdArgs a) {
    extern __shared__ float4 dwl_smem[];
    __shared__ float red[4][10];
    // the dx rows are collected where the x image was: it is dead once the weight gradient is done
    float* xl = reinterpret_cast<float*>(dwl_smem);
    float* dl = xl;
    float* gl = xl + a.x_floats;  // x_floats covers the larger of the two
    float* wl = gl + a.g_floats;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int pb = tile / a.g.NB, band = tile - pb * a.g.NB;
    const int p0 = pb * a.g.P;
    const int Pe = min(a.g.P, a.planes - p0);
    const bool multi = a.g.P > 1;
    const int oh0 = band * a.g.BR;
    const int BRt = min(a.g.BR, a.OH - oh0);
    // x image: input rows ihb .. ihb + (BRt - 1) * S + 2 (weight gradient of the tile's own output rows)
    const int ihb = oh0 * S - 1;
    const int r0 = max(ihb, 0), r1 = min(ihb + (BRt - 1) * S + 3, a.H);
    const float* gx = a.x + ((size_t)p0 * a.H + r0) * a.W;
    const int xcount = Pe * (r1 - r0) * a.W;
    const bool xrowvec = (a.W & 3) == 0;
    const bool xvec = (xrowvec || multi) && ((reinterpret_cast<uintptr_t>(gx) & 15) == 0) && (xcount & 3) == 0;
    DwlImg mx;
    mx.Wd = a.W; mx.PW = a.g.PWX; mx.RP = a.g.RPX; mx.rpp = r1 - r0; mx.lr0 = r0 - ihb; mx.wd_magic = a.w_magic; mx.rpp_magic = a.h_magic;
    // g image: output rows oh0 - 1 .. oh0 + BRt (data gradient of the tile's own input rows)
    const int ogb = oh0 - 1;
    const int q0 = max(ogb, 0), q1 = min(oh0 + BRt + 1, a.OH);
    const size_t goff = ((size_t)p0 * a.OH + q0) * a.OW;
    const float* gsrc = (BN ? a.bn.dz : a.dy) + goff;
    const float* ysrc = a.y + goff;
    const int gcount = Pe * (q1 - q0) * a.OW;
    const bool growvec = (a.OW & 3) == 0;
    const bool need_y = BN || a.act != BCNN_HIP_ACT_NONE;
    const bool gvec = (growvec || multi) && (((reinterpret_cast<uintptr_t>(gsrc) | reinterpret_cast<uintptr_t>(ysrc)) & 15) == 0) &&
                      (gcount & 3) == 0;
    DwlImg mg;
    mg.Wd = a.OW; mg.PW = a.g.PWG; mg.RP = a.g.RPG; mg.rpp = q1 - q0; mg.lr0 = q0 - ogb; mg.wd_magic = a.ow_magic; mg.rpp_magic = a.oh_magic;
    // the input rows this tile owns (data gradient): [i0, i1)
    const int i0 = oh0 * S, i1 = min((oh0 + BRt) * S, a.H);
    const int own = i1 - i0;
    float* gdx = a.dx + ((size_t)p0 * a.H + i0) * a.W;
    const int dcount = Pe * own * a.W;

    float4 xv[kMaxQ], gv4[kMaxQ], yv4[kMaxQ];
    dwl_fetch<kMaxQ>(gx, xcount, xvec, xv);
    dwl_fetch<kMaxQ>(gsrc, gcount, gvec, gv4);
    if (need_y) dwl_fetch<kMaxQ>(ysrc, gcount, gvec, yv4);
    dwl_zero(xl, a.x_floats + a.g_floats);
    for (int i = tid; i < Pe * kConst; i += 256) {
        const int j = i / kConst, t = i - j * kConst, c = (p0 + j) % a.C;
        float v = 0.f;
        if (t < 9) v = a.w[c * 9 + t];
        else if (BN) {
            if (t == 9) v = a.bn.mean[c];
            else if (t == 10) v = sqrtf(a.bn.var[c] + 0.00001f);
            else if (t == 11) v = a.bn.scale[c];
            else if (t == 12) v = __fdiv_rn(a.bn.dmean[c], a.fM);
            else if (t == 13) v = a.bn.dvar[c];
            else if (t == 14) v = __fdiv_rn(1.0f, sqrtf(a.bn.var[c] + 0.00001f));
        }
        if (BNIN && t >= 16 && t <= 20) v = dwl_bnin_const(a.in, c, t);
        wl[i] = v;
    }
    __syncthreads();
    unsigned in_pass_bits = 0u;  // BNIN with sums: which of this thread's staged elements the producer's activation passes
    if (BNIN) dwl_scatter_bn<kMaxQ>(gx, xcount, xvec, xrowvec, multi, mx, xv, xl, wl, a.in.act, a.in_sums ? &in_pass_bits : nullptr);
    else dwl_scatter<kMaxQ>(gx, xcount, xvec, xrowvec, multi, mx, xv, xl);
    {
        // g = [batch-norm backward of dz] * act'(y), into the image and (own rows, no batch-norm) back over dy
        auto one = [&](float gin, float yv, int j) -> float {
            float g = gin;
            if (BN && !(DWL_ABL & 4)) {
                const float* k = wl + j * kConst;
                const BnDiv rs{k[10], k[14]}, fM{a.fM, a.rfM};
                g = bn_bwd_one(gin, 0.f, yv, k[9], rs, k[11], k[12], k[13], fM, BCNN_HIP_ACT_NONE);
            }
            if (a.act != BCNN_HIP_ACT_NONE) g *= act_bwd_cheap(yv, a.act, 0.f);
            return g;
        };
        const bool wb = !BN && a.write_back && a.act != BCNN_HIP_ACT_NONE;
        float* gdy = a.dy + goff;
        int j, row;
        if (gvec) {
#pragma unroll
            for (int q = 0; q < kMaxQ; ++q) {
                const int e = (tid + q * 256) * 4;
                if (e >= gcount) continue;
                const float4 gi = gv4[q];
                float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (need_y) yv = yv4[q];
                float4 o;
                if (growvec) {
                    const int slot = dwl_slot(mg, e, multi, j, row);
                    o.x = one(gi.x, yv.x, j); o.y = one(gi.y, yv.y, j); o.z = one(gi.z, yv.z, j); o.w = one(gi.w, yv.w, j);
                    *reinterpret_cast<float4*>(gl + slot) = o;
                    const int oh = q0 + row;
                    if (wb && oh >= oh0 && oh < oh0 + BRt) reinterpret_cast<float4*>(gdy)[tid + q * 256] = o;
                } else {  // whole planes: every row is the tile's own
                    int s0 = dwl_slot(mg, e, multi, j, row);
                    int col = s0 - ((j * mg.RP + mg.lr0 + row) * mg.PW + 4);
                    o.x = one(gi.x, yv.x, j); gl[s0] = o.x;
                    dwl_next(mg, multi, s0, j, row, col);
                    o.y = one(gi.y, yv.y, j); gl[s0] = o.y;
                    dwl_next(mg, multi, s0, j, row, col);
                    o.z = one(gi.z, yv.z, j); gl[s0] = o.z;
                    dwl_next(mg, multi, s0, j, row, col);
                    o.w = one(gi.w, yv.w, j); gl[s0] = o.w;
                    if (wb) reinterpret_cast<float4*>(gdy)[tid + q * 256] = o;
                }
            }
        } else {
            for (int e = tid; e < gcount; e += 256) {
                const int slot = dwl_slot(mg, e, multi, j, row);
                const float o = one(gsrc[e], need_y ? ysrc[e] : 0.f, j);
                gl[slot] = o;
                const int oh = q0 + row;
                if (wb && oh >= oh0 && oh < oh0 + BRt) gdy[e] = o;
            }
        }
    }
    __syncthreads();

    // ---- weight gradient + bias gradient over the tile's own output rows: one partial per plane ----
    // One plane per workgroup: all 256 threads share it. Several planes: every 16-lane row of a wave takes a plane of its
    // own (small planes have few items -- 14 on a 7 x 7 plane -- and a row sum costs four DPP steps instead of six).
    if (!(DWL_ABL & 1)) {
        constexpr int NR = (VR - 1) * S + 3;
        const int items = a.RG * a.OW;
        const bool rows = a.g.P >= 8;  // fewer planes than that: a whole wave per plane keeps all four waves busy
        const int jstep = multi ? (rows ? 16 : 4) : 1, jfirst = multi ? (rows ? wid * 4 + (lane >> 4) : wid) : 0;
        const int ifirst = multi ? (rows ? (lane & 15) : lane) : tid, istep = multi ? (rows ? 16 : 64) : 256;
        const int jend = (multi && rows) ? ((Pe + 15) & ~15) : Pe;  // whole waves walk together (DPP needs the lanes on)
        for (int j = jfirst; j < jend; j += jstep) {
            float acc[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) acc[i] = 0.f;
            for (int item = ifirst; item < items && j < Pe; item += istep) {
                const int rg = (int)dwl_div((unsigned)item, a.ow_magic), ow = item - rg * a.OW, rl0 = rg * VR;
                const float* ip = xl + (j * a.g.RPX + rl0 * S) * a.g.PWX + 3 + ow * S;
                const float* gp = gl + (j * a.g.RPG + rl0 + 1) * a.g.PWG + 4 + ow;
                float gvv[VR];
#pragma unroll
                for (int r = 0; r < VR; ++r) {
                    const float v = gp[r * a.g.PWG];
                    gvv[r] = (rl0 + r < BRt) ? v : 0.f;  // the row behind the band belongs to the next tile
                    acc[9] += gvv[r];
                }
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const float x0 = ip[i * a.g.PWX], x1 = ip[i * a.g.PWX + 1], x2 = ip[i * a.g.PWX + 2];
#pragma unroll
                    for (int r = 0; r < VR; ++r) {
                        const int kh = i - r * S;
                        if (kh < 0 || kh > 2) continue;
                        // fused multiply-add: these are partial sums in this kernel's own order (the reference's sum over
                        // the batch is reproduced to 1e-4, not bit for bit) and the kernel is vector-ALU bound
                        acc[kh * 3 + 0] = __fmaf_rn(x0, gvv[r], acc[kh * 3 + 0]);
                        acc[kh * 3 + 1] = __fmaf_rn(x1, gvv[r], acc[kh * 3 + 1]);
                        acc[kh * 3 + 2] = __fmaf_rn(x2, gvv[r], acc[kh * 3 + 2]);
                    }
                }
            }
            if (multi) {
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i] = rows ? row_sum_dpp(acc[i]) : wave_sum_dpp(acc[i]);
                if ((rows ? (lane & 15) == 0 : lane == 63) && j < Pe) {
                    const int p = p0 + j, n = p / a.C, c = p - n * a.C;
                    float* dst = a.partials + ((size_t)c * a.splits + n) * kPart;
#pragma unroll
                    for (int i = 0; i < 10; ++i) dst[i] = acc[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i] = wave_sum_dpp(acc[i]);
                if (lane == 63) {
#pragma unroll
                    for (int i = 0; i < 10; ++i) red[wid][i] = acc[i];
                }
                __syncthreads();
                if (tid < 10) {
                    const int n = p0 / a.C, c = p0 - n * a.C;
                    a.partials[((size_t)c * a.splits + (size_t)n * a.g.NB + band) * kPart + tid] =
                        (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
                }
            }
        }
    }

    __syncthreads();  // every wave is done with the x image: its space now collects dx
    if (!a.overwrite) {  // dx accumulates onto what is there: the sums start from the old values
        if (((reinterpret_cast<uintptr_t>(gdx) & 15) == 0) && (dcount & 3) == 0) {
            for (int i = tid; i < dcount / 4; i += 256) reinterpret_cast<float4*>(dl)[i] = reinterpret_cast<const float4*>(gdx)[i];
        } else {
            for (int i = tid; i < dcount; i += 256) dl[i] = gdx[i];
        }
        __syncthreads();
    }
    // ---- data gradient of the tile's own input rows, per pixel the taps in the reference's scatter order
    //      (descending kh, kw == ascending output position) ----
    if (DWL_ABL & 2) {
    } else if (S == 1) {
        constexpr int VRX = 4;
        const int rgx_n = a.RGX;
        const int items = Pe * rgx_n * a.W;
        for (int item = tid; item < items; item += 256) {
            const int t = (int)dwl_div((unsigned)item, a.w_magic), iw = item - t * a.W;
            const int j = (int)dwl_div((unsigned)t, a.rgx_magic), il0 = (t - j * rgx_n) * VRX;
            const float* wp = wl + j * kConst;
            float wv[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) wv[i] = wp[i];
            float* dp = dl + (j * own + il0) * a.W + iw;
            float acc[VRX];
#pragma unroll
            for (int r = 0; r < VRX; ++r) acc[r] = (!a.overwrite && il0 + r < own) ? dp[r * a.W] : 0.f;
            const float* gp = gl + (j * a.g.RPG + il0) * a.g.PWG + 3 + iw;
#pragma unroll
            for (int tt = 0; tt < VRX + 2; ++tt) {
                const float g0 = gp[tt * a.g.PWG], g1 = gp[tt * a.g.PWG + 1], g2 = gp[tt * a.g.PWG + 2];
#pragma unroll
                for (int r = 0; r < VRX; ++r) {
                    const int kh = r + 2 - tt;  // ascending tt == descending kh per input row
                    if (kh < 0 || kh > 2) continue;
                    acc[r] = __fadd_rn(acc[r], __fmul_rn(wv[kh * 3 + 2], g0));
                    acc[r] = __fadd_rn(acc[r], __fmul_rn(wv[kh * 3 + 1], g1));
                    acc[r] = __fadd_rn(acc[r], __fmul_rn(wv[kh * 3 + 0], g2));
                }
            }
#pragma unroll
            for (int r = 0; r < VRX; ++r)
                if (il0 + r < own) dp[r * a.W] = acc[r];
        }
    } else {
        // stride 2: one thread per 2 x 2 input block (2a, 2b): the four parity classes meet 1, 2, 2 and 4 taps
        const int hw2 = (a.W + 1) >> 1;
        const int items = Pe * BRt * hw2;
        for (int item = tid; item < items; item += 256) {
            const int t = (int)dwl_div((unsigned)item, a.hw2_magic), b = item - t * hw2;
            int j = 0, ar = t;
            if (multi) {
                j = (int)dwl_div((unsigned)t, a.oh_magic);
                ar = t - j * a.OH;
            }
            const float* wp = wl + j * kConst;
            float wv[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) wv[i] = wp[i];
            const float* gp = gl + (j * a.g.RPG + ar + 1) * a.g.PWG + 4 + b;
            const float g00 = gp[0], g01 = gp[1], g10 = gp[a.g.PWG], g11 = gp[a.g.PWG + 1];
            const int il = 2 * ar, iw = 2 * b;
            float* dp = dl + (j * own + il) * a.W + iw;
            const bool c1 = iw + 1 < a.W, r1ok = il + 1 < own;
            float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
            if (!a.overwrite) {
                v00 = dp[0];
                if (c1) v01 = dp[1];
                if (r1ok) v10 = dp[a.W];
                if (r1ok && c1) v11 = dp[a.W + 1];
            }
            v00 = __fadd_rn(v00, __fmul_rn(wv[4], g00));
            v01 = __fadd_rn(v01, __fmul_rn(wv[5], g00));
            v01 = __fadd_rn(v01, __fmul_rn(wv[3], g01));
            v10 = __fadd_rn(v10, __fmul_rn(wv[7], g00));
            v10 = __fadd_rn(v10, __fmul_rn(wv[1], g10));
            v11 = __fadd_rn(v11, __fmul_rn(wv[8], g00));
            v11 = __fadd_rn(v11, __fmul_rn(wv[6], g01));
            v11 = __fadd_rn(v11, __fmul_rn(wv[2], g10));
            v11 = __fadd_rn(v11, __fmul_rn(wv[0], g11));
            dp[0] = v00;
            if (c1) dp[1] = v01;
            if (r1ok) dp[a.W] = v10;
            if (r1ok && c1) dp[a.W + 1] = v11;
        }
    }
    __syncthreads();
    // whole planes with 16-byte pieces: the thread that staged elements 4 i .. 4 i + 3 also copies them out (below)
    const bool sums_fused = BNIN && a.in_sums != nullptr && multi && xvec && dcount == xcount && a.H * a.W >= 4 &&
                            ((reinterpret_cast<uintptr_t>(gdx) & 15) == 0);
    if (!sums_fused) dwl_copy_out(dl, gdx, dcount);
    // ---- the producer's batch-norm backward starts with S1 = sum g, S2 = sum g * (raw - mean) over (n, h, w) per channel,
    //      g = dx * act'(y_in) (bcnn_batchnorm_layer.c:263-281 behind bcnn_backward_activation_cpu). dx is complete here,
    //      raw is the piece this thread staged (still in its registers) and act'(y_in) in {0, 1} was noted while staging it,
    //      so the sweep that would re-read both tensors for these sums is not needed. The kernel is vector-ALU bound:
    //      g and g * (raw - mean) are formed once per element by the thread that staged it (dx piece and the dead g image
    //      as LDS space), then summed per plane. One partial per tile and plane, like the weight gradient above. ----
    if (BNIN && a.in_sums != nullptr) {
        const int per = own * a.W;  // several planes per tile: whole planes, own == H
        float* rl = gl;             // several planes per tile: the host made the g image's space hold a dx piece
        if (!multi) {
            // one plane per tile: every element a thread staged belongs to it -- sums in registers, no LDS arrays
            const int shift = (i0 - r0) * a.W;  // the staged piece starts one halo row above the own rows (not on the first band)
            const float mean = wl[16];
            float s1 = 0.f, s2 = 0.f;
            auto add = [&](int e, float r, unsigned pass) {
                const int o = e - shift;
                if (o < 0 || o >= dcount) return;
                const float g = pass ? dl[o] : 0.f;
                s1 += g;
                s2 += g * (r - mean);
            };
            if (xvec) {
#pragma unroll
                for (int q = 0; q < kMaxQ; ++q) {
                    const int e = (tid + q * 256) * 4;
                    if (e >= xcount) continue;
                    add(e, xv[q].x, in_pass_bits & (1u << (4 * q)));
                    add(e + 1, xv[q].y, in_pass_bits & (2u << (4 * q)));
                    add(e + 2, xv[q].z, in_pass_bits & (4u << (4 * q)));
                    add(e + 3, xv[q].w, in_pass_bits & (8u << (4 * q)));
                }
            } else {
                int bit = 0;
                for (int e = tid; e < xcount; e += 256, ++bit) add(e, gx[e], in_pass_bits & (1u << bit));
            }
            s1 = wave_sum_dpp(s1);
            s2 = wave_sum_dpp(s2);
            if (lane == 63) { red[wid][0] = s1; red[wid][1] = s2; }
            __syncthreads();
            if (tid < 2) {
                const int n = p0 / a.C, c = p0 - n * a.C;
                a.in_sums[((size_t)c * a.splits + (size_t)n * a.g.NB + band) * 2 + tid] =
                    (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            }
            return;
        }
        if (sums_fused) {
#pragma unroll
            for (int q = 0; q < kMaxQ; ++q) {
                const int i = tid + q * 256;
                if (i * 4 >= dcount) continue;
                const float4 d = reinterpret_cast<const float4*>(dl)[i];
                reinterpret_cast<float4*>(gdx)[i] = d;
                const int j0 = (int)dwl_div((unsigned)(4 * i), a.hw_magic), j3 = (int)dwl_div((unsigned)(4 * i + 3), a.hw_magic);
                const float m0 = wl[j0 * kConst + 16], m3 = wl[j3 * kConst + 16];
                const int edge = (j0 + 1) * per - 4 * i;  // elements of the four that still belong to plane j0 (>= 1)
                float4 g, t;
                g.x = (in_pass_bits & (1u << (4 * q))) ? d.x : 0.f;
                g.y = (in_pass_bits & (2u << (4 * q))) ? d.y : 0.f;
                g.z = (in_pass_bits & (4u << (4 * q))) ? d.z : 0.f;
                g.w = (in_pass_bits & (8u << (4 * q))) ? d.w : 0.f;
                t.x = g.x * (xv[q].x - m0);
                t.y = g.y * (xv[q].y - (edge > 1 ? m0 : m3));
                t.z = g.z * (xv[q].z - (edge > 2 ? m0 : m3));
                t.w = g.w * (xv[q].w - m3);
                reinterpret_cast<float4*>(dl)[i] = g;
                reinterpret_cast<float4*>(rl)[i] = t;
            }
        } else {
            __syncthreads();  // the copy above has read every dx value
            const int shift = (i0 - r0) * a.W;  // the staged piece starts one halo row above the own rows (not on the first band)
            auto put = [&](int e, float r, unsigned pass) {
                const int o = e - shift;
                if (o < 0 || o >= dcount) return;
                const int j = multi ? (int)dwl_div((unsigned)o, a.hw_magic) : 0;
                const float g = pass ? dl[o] : 0.f;
                dl[o] = g;
                rl[o] = g * (r - wl[j * kConst + 16]);
            };
            if (xvec) {
#pragma unroll
                for (int q = 0; q < kMaxQ; ++q) {
                    const int e = (tid + q * 256) * 4;
                    if (e >= xcount) continue;
                    put(e, xv[q].x, in_pass_bits & (1u << (4 * q)));
                    put(e + 1, xv[q].y, in_pass_bits & (2u << (4 * q)));
                    put(e + 2, xv[q].z, in_pass_bits & (4u << (4 * q)));
                    put(e + 3, xv[q].w, in_pass_bits & (8u << (4 * q)));
                }
            } else {
                int bit = 0;
                for (int e = tid; e < xcount; e += 256, ++bit) put(e, gx[e], in_pass_bits & (1u << bit));
            }
        }
        __syncthreads();
        const bool rows = a.g.P >= 8;
        const int jstep = multi ? (rows ? 16 : 4) : 1, jfirst = multi ? (rows ? wid * 4 + (lane >> 4) : wid) : 0;
        const int ifirst = multi ? (rows ? (lane & 15) : lane) : tid, istep = multi ? (rows ? 16 : 64) : 256;
        const int jend = (multi && rows) ? ((Pe + 15) & ~15) : Pe;
        for (int j = jfirst; j < jend; j += jstep) {
            float s1 = 0.f, s2 = 0.f;
            if (j < Pe) {
                const float* dj = dl + j * per;
                const float* rj = rl + j * per;
                for (int e = ifirst; e < per; e += istep) {
                    s1 += dj[e];
                    s2 += rj[e];
                }
            }
            if (multi) {
                s1 = rows ? row_sum_dpp(s1) : wave_sum_dpp(s1);
                s2 = rows ? row_sum_dpp(s2) : wave_sum_dpp(s2);
                if ((rows ? (lane & 15) == 0 : lane == 63) && j < Pe) {
                    const int p = p0 + j, n = p / a.C, c = p - n * a.C;
                    float* dst = a.in_sums + ((size_t)c * a.splits + n) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            } else {
                s1 = wave_sum_dpp(s1);
                s2 = wave_sum_dpp(s2);
                if (lane == 63) { red[wid][0] = s1; red[wid][1] = s2; }
                __syncthreads();
                if (tid < 2) {
                    const int n = p0 / a.C, c = p0 - n * a.C;
                    a.in_sums[((size_t)c * a.splits + (size_t)n * a.g.NB + band) * 2 + tid] =
                        (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
                }
            }
        }
    }
}

// dw[c][t] += sum over slots (double, fixed order); dbias[c] += the tenth column when `dbias`
__global__ __launch_bounds__(256) void dwl_finalize_kernel(const float* __restrict__ partials, int splits,
                                                           float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ double red[4][10];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float* p = partials + (size_t)c * splits * kPart;
    double s[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = 0.0;
    for (int k = threadIdx.x; k < splits; k += 256) {
        const float4 a0 = *reinterpret_cast<const float4*>(p + (size_t)k * kPart);
        const float4 a1 = *reinterpret_cast<const float4*>(p + (size_t)k * kPart + 4);
        const float2 a2 = *reinterpret_cast<const float2*>(p + (size_t)k * kPart + 8);
        s[0] += (double)a0.x; s[1] += (double)a0.y; s[2] += (double)a0.z; s[3] += (double)a0.w;
        s[4] += (double)a1.x; s[5] += (double)a1.y; s[6] += (double)a1.z; s[7] += (double)a1.w;
        s[8] += (double)a2.x; s[9] += (double)a2.y;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[i] += __shfl_xor(s[i], o);
        if (lane == 0) red[wid][i] = s[i];
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const int i = threadIdx.x;
        const double t = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (i < 9) dw[c * 9 + i] += (float)t;
        else if (dbias) dbias[c] += (float)t;
    }
}

constexpr int kVR1 = 4, kVR2 = 2;  // output rows per thread, stride 1 / 2

}  // namespace

void dwl_finalize_launch(const float* partials, int splits, int C, float* dw, float* dbias, hipStream_t st) {
    dwl_finalize_kernel<<<C, 256, 0, st>>>(partials, splits, dw, dbias);
    KERNEL_CHECK();
}

bool depthwise_lds_ok(const DwShape& s) {
    static const int on = BCNN_EXP_ENV("BCNN_HIP_NO_DW_LDS") ? 0 : 1;  // A/B switch (experiment build only)
    if (!on) return false;
    if (s.ksz != 3 || s.pad != 1 || (s.stride != 1 && s.stride != 2)) return false;
    if (s.N < 1 || s.C < 1 || s.H < 1 || s.W < 1 || s.OH < 1 || s.OW < 1) return false;
    if (s.W > kTileFloats / 8) return false;  // a band needs at least 3 + stride input rows of <= kTileFloats floats
    const DwlGeom g = dwl_plan(s);
    const long long tiles = (long long)ceil_div((long long)s.N * s.C, g.P) * g.NB;
    return tiles < 0x7fffffffLL && (long long)s.N * s.C * s.H * s.W < 0x7fffffffLL * 2;
}

size_t depthwise_lds_stats_floats(const DwShape& s) {
    if (!depthwise_lds_ok(s)) return 0;
    const DwlGeom g = dwl_plan(s);
    const size_t march = (size_t)s.C * depthwise_march_splits(s) * 2;  // the marching kernels' slots, where they take the shape
    return std::max((size_t)s.C * s.N * g.NB * 2, march);
}

size_t depthwise_lds_partial_floats(const DwShape& s) {
    if (!depthwise_lds_ok(s)) return 0;
    const DwlGeom g = dwl_plan(s);
    return std::max((size_t)s.C * s.N * g.NB * kPart, (size_t)s.C * depthwise_march_splits(s) * kPart);
}

bool depthwise_forward_lds(const float* x, const float* w, const float* bias, float* y, const DwShape& s, int act,
                           ConvStats* stats, const DwBnIn* in) {
    if (!depthwise_lds_ok(s) || !act_is_cheap(act) || act == BCNN_HIP_ACT_PRELU) return false;
    if (in && (!in->mean || !act_is_cheap(in->act) || in->act == BCNN_HIP_ACT_PRELU)) return false;
    if (depthwise_forward_march(x, w, bias, y, s, act, stats, in)) return true;  // rows of whole 16-byte groups
    DwlFwdArgs a;
    a.g = dwl_plan(s);
    const int VR = s.stride == 1 ? kVR1 : kVR2;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = nullptr;
    a.C = s.C; a.H = s.H; a.W = s.W; a.OH = s.OH; a.OW = s.OW; a.planes = s.N * s.C; a.act = act;
    a.splits = s.N * a.g.NB;
    a.RG = ceil_div(a.g.BR, VR);
    a.g.stage_floats = (a.g.P * a.g.BR * s.OW + 3) & ~3;
    a.x_floats = (a.g.rows_x * a.g.PWX + 3) & ~3;
    a.w_magic = dwl_magic((unsigned)s.W); a.h_magic = dwl_magic((unsigned)s.H);
    a.ow_magic = dwl_magic((unsigned)s.OW); a.rg_magic = dwl_magic((unsigned)a.RG);
    if (stats) {
        stats->splits = 0;
        if (stats->partials && stats->capacity >= (size_t)s.C * a.splits * 2) {
            a.stats = stats->partials;
            stats->splits = a.splits;
        }
    }
    const size_t lds = (size_t)(a.x_floats + a.g.stage_floats + a.g.P * kConst) * sizeof(float);
    if (lds > 64 * 1024) return false;
    const unsigned tiles = (unsigned)ceil_div((long long)a.planes, a.g.P) * (unsigned)a.g.NB;
    if (in) {
        a.in = *in;
        if (s.stride == 1) dwl_fwd_kernel<1, kVR1, true><<<tiles, 256, lds, current_stream()>>>(a);
        else dwl_fwd_kernel<2, kVR2, true><<<tiles, 256, lds, current_stream()>>>(a);
    } else {
        a.in = DwBnIn{nullptr, nullptr, nullptr, nullptr, 0};
        if (s.stride == 1) dwl_fwd_kernel<1, kVR1, false><<<tiles, 256, lds, current_stream()>>>(a);
        else dwl_fwd_kernel<2, kVR2, false><<<tiles, 256, lds, current_stream()>>>(a);
    }
    KERNEL_CHECK();
    return true;
}

size_t depthwise_lds_in_sums_floats(const DwShape& s) {
    if (!depthwise_lds_ok(s)) return 0;
    const DwlGeom g = dwl_plan(s);
    return std::max((size_t)s.C * s.N * g.NB * 2, (size_t)s.C * depthwise_march_splits(s) * 2);
}

bool depthwise_backward_lds(const float* x, const float* w, const float* y, float* dy, float* dx, float* dw, float* dbias,
                            const DwShape& s, int act, int overwrite, int write_back, const DwBnBwd* bn,
                            const DwBnIn* in, ConvStats* in_sums) {
    if (in_sums) in_sums->splits = 0;
    if (!depthwise_lds_ok(s) || !act_bwd_is_cheap(act) || act == BCNN_HIP_ACT_PRELU || !dx) return false;
    if (in && (!in->mean || !act_is_cheap(in->act) || in->act == BCNN_HIP_ACT_PRELU)) return false;
    // Everything that can still refuse the layer is decided BEFORE dy is touched (a pre-pass followed by a refusal
    // left the caller to apply the derivative a second time).
    DwlBwdArgs a;
    a.g = dwl_plan(s);
    const int S = s.stride, VR = S == 1 ? kVR1 : kVR2;
    a.splits = s.N * a.g.NB;
    a.RG = ceil_div(a.g.BR, VR);
    a.RGX = ceil_div(a.g.BR, 4);  // stride 1: a tile owns as many input rows as output rows
    const int own_rows = a.g.BR * S < s.H ? a.g.BR * S : s.H;
    a.g.stage_floats = (a.g.P * own_rows * s.W + 3) & ~3;
    a.x_floats = (a.g.rows_x * a.g.PWX + 3) & ~3;
    if (a.x_floats < a.g.stage_floats) a.x_floats = a.g.stage_floats;  // the dx rows reuse the x image's space
    a.g_floats = (a.g.rows_g * a.g.PWG + 3) & ~3;
    // the sums of the producer's batch-norm backward are of the COMPLETE gradient: only when this kernel is its sole writer;
    // for producer activations whose derivative is 0 or 1 (noted as one bit per element while staging); tiles of several
    // planes form g and g * (raw - mean) in LDS -- the dx piece and the g image's space, which then must hold a dx piece
    bool want_sums = in && in_sums && in_sums->partials && overwrite &&
                     (in->act == BCNN_HIP_ACT_NONE || in->act == BCNN_HIP_ACT_RELU) &&
                     in_sums->capacity >= (size_t)s.C * a.splits * 2;
    if (want_sums && a.g.P > 1 && a.g_floats < a.g.stage_floats) {  // stride 2
        if ((size_t)(a.x_floats + a.g.stage_floats + a.g.P * kConst) * sizeof(float) <= 64 * 1024) a.g_floats = a.g.stage_floats;
        else want_sums = false;
    }
    const size_t lds = (size_t)(a.x_floats + a.g_floats + a.g.P * kConst) * sizeof(float);
    const bool lds_takes = lds <= 64 * 1024;
    const bool march_takes = depthwise_backward_march_takes(x, y, dy, dx, s, act, bn, in);
    if (!lds_takes && !march_takes) return false;
    if (!bn && write_back && act != BCNN_HIP_ACT_NONE && act != BCNN_HIP_ACT_RELU && act != BCNN_HIP_ACT_CLAMP) {
        // g = dy * act'(y) is written back over dy by the band that owns the row, while the neighbouring band reads the same
        // row as its halo: harmless when applying the derivative twice changes nothing (a factor 0 or 1), a race otherwise
        // (leaky ReLU: 0.01 instead of 0.1 on a band's edge row, whenever the owner happened to run first). Those
        // activations get their own in-place pass first.
        bcnn_hip_activation_backward(y, dy, (size_t)s.N * s.C * s.OH * s.OW, act, nullptr, nullptr, s.OH * s.OW, s.C);
        act = BCNN_HIP_ACT_NONE;
    }
    if (march_takes) {
        if (depthwise_backward_march(x, w, y, dy, dx, dw, dbias, s, act, overwrite, write_back, bn, in, in_sums)) return true;
        fprintf(stderr, "[bcnn_hip] depthwise_backward_march refused a layer depthwise_backward_march_takes accepted\n");
        exit(1);
    }
    a.x = x; a.w = w; a.y = y; a.dy = dy; a.dx = dx;
    a.C = s.C; a.H = s.H; a.W = s.W; a.OH = s.OH; a.OW = s.OW; a.planes = s.N * s.C; a.act = act;
    a.overwrite = overwrite; a.write_back = write_back;
    a.w_magic = dwl_magic((unsigned)s.W); a.h_magic = dwl_magic((unsigned)s.H);
    a.ow_magic = dwl_magic((unsigned)s.OW); a.oh_magic = dwl_magic((unsigned)s.OH);
    a.rg_magic = dwl_magic((unsigned)a.RG); a.rgx_magic = dwl_magic((unsigned)a.RGX);
    a.hw2_magic = dwl_magic((unsigned)((s.W + 1) >> 1));
    a.hw_magic = dwl_magic((unsigned)(s.H * s.W));
    a.fM = (float)((long long)s.N * s.OH * s.OW);
    a.rfM = 1.0f / a.fM;  // host division: IEEE, round to nearest
    if (bn) a.bn = *bn;
    else a.bn = DwBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    a.partials = reduce_scratch((size_t)s.C * a.splits * kPart);
    a.in_sums = nullptr;
    if (want_sums) {
        a.in_sums = in_sums->partials;
        in_sums->splits = a.splits;
    }
    const unsigned tiles = (unsigned)ceil_div((long long)a.planes, a.g.P) * (unsigned)a.g.NB;
    hipStream_t st = current_stream();
    a.in = in ? *in : DwBnIn{nullptr, nullptr, nullptr, nullptr, 0};
#define DWL_LAUNCH(SV, VRV)                                                                     \
    do {                                                                                        \
        if (bn && in) dwl_bwd_kernel<SV, VRV, true, true><<<tiles, 256, lds, st>>>(a);          \
        else if (bn) dwl_bwd_kernel<SV, VRV, true, false><<<tiles, 256, lds, st>>>(a);          \
        else if (in) dwl_bwd_kernel<SV, VRV, false, true><<<tiles, 256, lds, st>>>(a);          \
        else dwl_bwd_kernel<SV, VRV, false, false><<<tiles, 256, lds, st>>>(a);                 \
    } while (0)
    if (S == 1) DWL_LAUNCH(1, kVR1);
    else DWL_LAUNCH(2, kVR2);
#undef DWL_LAUNCH
    KERNEL_CHECK();
    dwl_finalize_kernel<<<s.C, 256, 0, st>>>(a.partials, a.splits, dw, dbias);
    KERNEL_CHECK();
    return true;
}

}  // namespace bcnn_hip
