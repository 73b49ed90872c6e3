// depthwise_march.hip -- 3x3 depthwise convolution (pad 1, stride 1 or 2), rows of up to 64 column groups: every lane owns
// a group of V = 4 / 2 / 1 adjacent columns of a plane (16 / 8 / 4 bytes, the widest the row length allows) and MARCHES
// down a band of rows with a three-row register window.
//
// Reference semantics: src/layers/bcnn_depthwise_conv_layer.c:165-293 (forward), :295-547 (backward); the stand-alone
// batch-norm that follows a depthwise layer in MobileNet: src/layers/bcnn_batchnorm_layer.c:196-242, :292-296.
//
// Why a third set of kernels (round 4): rocprofv3 SQ counters (profiles/r04_sq_pmc_depthwise.txt) showed the LDS-staged
// kernels of depthwise_lds.hip to be vector-ALU bound, not memory bound: 1453 vector instructions per wave on the
// 112 x 112 backward (118 per element and lane, 82 % of the SIMD issue cycles), most of them staging -- image scatter with
// magic divisions, zero fill, collect-and-copy-out loops, per-item index arithmetic -- around ~45 instructions of real work.
// Here nothing is staged:
//   * a lane's window is a float4 per row plus the two neighbour values it takes from the adjacent lanes with
//     ds_bpermute (no LDS storage, no vector-ALU cost); image borders are a select per row, rows outside the image are
//     zeros that never left a register;
//   * all global traffic is one V-float access per lane (stride 2 outputs: V / 2) with the lanes of a row contiguous, loaded
//     one or two rows ahead of its use; every element is read once per band (plus one halo row per band end, an L2 hit);
//     addresses are a uniform base plus a 32-bit byte offset per lane, advanced by one add per row and stream;
//   * results go straight from registers to global memory;
//   * lanes of a wave that do not fit a row (64 mod W/V) idle; a wave holds 64 / (W/V) independent bands, which may lie in
//     different planes, so small planes (14 x 14: nine per wave, 7 x 7 likewise) fill waves as well as large ones.
// Tap order and the separate multiply / add roundings are the reference's (forward and data gradient: bit-exact); the
// reductions (weight / bias gradient, batch-norm sums) are two-level in a fixed order: one partial per band, summed
// across the lanes of a band through a per-wave LDS slab in lane order, then over bands in double by the finalize kernels.
#include "bn_math.h"
#include "depthwise.h"

#include <initializer_list>

namespace bcnn_hip {

void dwl_finalize_launch(const float* partials, int splits, int C, float* dw, float* dbias, hipStream_t st);  // depthwise_lds.hip
float* reduce_scratch(size_t floats);                                                                          // blas1.hip

namespace {

#ifndef DWM_BWD_WAVES
#define DWM_BWD_WAVES 3  // waves per SIMD the backward kernels are compiled for (register budget 512 / that; at 4 the
                         // 112 x 112 instance spills inside its loop and takes twice the time)
#endif
#ifndef DWM_ROWS
#define DWM_ROWS 14  // rows a band marches (target; the plan evens bands out)
#endif

struct DwmGeom {
    int V;    // columns a lane owns: 4 (rows of whole 16-byte groups), 2 (even widths) or 1
    int L;    // lanes per row (W / V)
    int G;    // bands per wave
    int len;  // rows per band: stride 1 rows of x == rows of y; stride 2 rows of y (two rows of x each)
    int BPP;  // bands per plane
    long long bands;
};

inline int dwm_width(const DwShape& s) {
    if (s.stride == 2) return (s.W & 3) == 0 ? 4 : ((s.W & 1) == 0 ? 2 : 0);
    return (s.W & 3) == 0 ? 4 : ((s.W & 1) == 0 ? 2 : 1);
}

inline bool dwm_shape_ok(const DwShape& s) {
    if (s.ksz != 3 || s.pad != 1 || (s.stride != 1 && s.stride != 2)) return false;
    if (s.N < 1 || s.C < 1 || s.H < 1 || s.W < 1) return false;
    const int V = dwm_width(s);
    if (V == 0 || s.W / V > 64) return false;
    if ((long long)s.N * s.C * s.H * s.W >= 0x3fffffffLL) return false;  // byte offsets are 32-bit
    return true;
}

inline DwmGeom dwm_plan(const DwShape& s) {
    DwmGeom g;
    g.V = dwm_width(s);
    g.L = s.W / g.V;
    g.G = 64 / g.L;
    const int R = s.OH;
    int rows = DWM_ROWS;
#ifdef BCNN_HIP_EXPERIMENT
    if (const char* e = getenv("BCNN_HIP_DWM_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
#endif
    g.BPP = ceil_div(R, rows);
    g.len = ceil_div(R, g.BPP);
    g.BPP = ceil_div(R, g.len);
    g.bands = (long long)s.N * s.C * g.BPP;
    return g;
}

struct DwmLane {
    bool on;           // this lane works on a band
    bool first, last;  // first / last column group of the row
    int cg, grp;
    unsigned band;     // global band index
    int p, c, bi;      // plane, channel, band within the plane
    int addr_l, addr_r;  // ds_bpermute byte addresses of the lanes to the left / right
};

__device__ __forceinline__ DwmLane dwm_lane(int L, int G, int BPP, int C, unsigned bands) {
    DwmLane m;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    m.grp = lane / L;
    m.cg = lane - m.grp * L;
    m.band = wave * (unsigned)G + (unsigned)m.grp;
    m.on = m.grp < G && m.band < bands;
    const unsigned b = m.on ? m.band : 0u;
    m.p = (int)(b / (unsigned)BPP);
    m.bi = (int)(b - (unsigned)m.p * (unsigned)BPP);
    m.c = m.p % C;
    m.first = m.cg == 0;
    m.last = m.cg == L - 1;
    m.addr_l = ((lane + 63) & 63) << 2;
    m.addr_r = ((lane + 1) & 63) << 2;
    return m;
}

__device__ __forceinline__ float dwm_from(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// V floats of a row. Addresses are a uniform base plus a 32-bit byte offset per lane (one add per row and stream).
template <int V>
struct Vals {
    float v[V];
};
template <int V>
__device__ __forceinline__ Vals<V> dwm_ld(const float* base, unsigned byte_off, bool ok) {
    Vals<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = 0.f;
    if (ok) {
        const char* p = reinterpret_cast<const char*>(base) + byte_off;
        if constexpr (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
        } else if constexpr (V == 2) {
            const float2 t = *reinterpret_cast<const float2*>(p);
            o.v[0] = t.x; o.v[1] = t.y;
        } else {
            o.v[0] = *reinterpret_cast<const float*>(p);
        }
    }
    return o;
}
template <int V>
__device__ __forceinline__ void dwm_st(float* base, unsigned byte_off, const Vals<V>& o) {
    char* p = reinterpret_cast<char*>(base) + byte_off;
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
    else if constexpr (V == 2) *reinterpret_cast<float2*>(p) = make_float2(o.v[0], o.v[1]);
    else *reinterpret_cast<float*>(p) = o.v[0];
}

// a row of the window: the lane's V values between the neighbours' (index 0 = left neighbour ... V + 1 = right neighbour)
template <int V>
struct Row {
    float v[V + 2];
};
template <int V, bool LEFT = true, bool RIGHT = true>
__device__ __forceinline__ Row<V> dwm_row(const Vals<V>& x, const DwmLane& m) {
    Row<V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[1 + i] = x.v[i];
    r.v[0] = r.v[V + 1] = 0.f;
    if (LEFT) {
        const float l = dwm_from(m.addr_l, x.v[V - 1]);
        r.v[0] = m.first ? 0.f : l;
    }
    if (RIGHT) {
        const float rr = dwm_from(m.addr_r, x.v[0]);
        r.v[V + 1] = m.last ? 0.f : rr;
    }
    return r;
}

// batch-norm + activation of the producing convolution node, applied to what was loaded: the arithmetic of bn_one
// (bn_math.h) with the per-channel special cases folded into the constants -- multiplying by a scale of 1 is exact, "no add
// for a bias of 0 or 1" is an add of -0; a scale of exactly 0 (memset in the reference) keeps its select
struct DwmBnInC {
    float mean, sc, b;
    BnDiv rs;
    bool sc0, any_sc0;
};
__device__ __forceinline__ DwmBnInC dwm_bnin_consts(const DwBnIn& in, int c) {
    DwmBnInC k;
    k.mean = in.mean[c];
    k.sc = in.scale[c];
    k.b = in.bias[c];
    if (k.b == 0.0f || k.b == 1.0f) k.b = -0.0f;
    k.sc0 = k.sc == 0.0f;
    k.any_sc0 = __builtin_amdgcn_ballot_w64(k.sc0) != 0;
    k.rs.d = sqrtf(in.var[c] + 0.000001f);
    k.rs.r = __fdiv_rn(1.0f, k.rs.d);
    return k;
}
__device__ __forceinline__ float dwm_bnin(float x, const DwmBnInC& k, int act) {
    float v = bn_div(__fsub_rn(x, k.mean), k.rs);
    v = __fmul_rn(v, k.sc);
    if (k.any_sc0 && k.sc0) v = 0.f;  // wave-uniform first: no channel of a trained net has a scale of exactly 0
    v = __fadd_rn(v, k.b);
    return act_fwd_cheap(v, act, 0.f);
}
template <int V>
__device__ __forceinline__ Vals<V> dwm_bnin_v(const Vals<V>& x, const DwmBnInC& k, int act) {
    Vals<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = dwm_bnin(x.v[i], k, act);
    return o;
}

// sum of NV per-lane values over the lanes of each band of the wave, in lane order; band q's totals are handed to
// put(q, value index, total) by lanes 0 .. G * NV - 1 (loop when that exceeds 64). `slab` = the wave's [NV][kSlab] floats.
constexpr int kSlab = 65;  // odd pitch: the NV sums of a band read different banks
template <int NV, class Put>
__device__ __forceinline__ void dwm_band_sums(const float (&v)[NV], float* slab, int L, int G, Put put) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) slab[i * kSlab + lane] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int idx = lane; idx < G * NV; idx += 64) {
        const int q = idx / NV, i = idx - q * NV;
        const float* src = slab + i * kSlab + q * L;
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += src[l];
        put(q, i, t);
    }
}

// ================================================================================================
// forward
// ================================================================================================
struct DwmFwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    float* stats;  // NULL: none; [C][splits][2], splits = N * BPP
    DwBnIn in;
    int C, H, W, OH, OW, act;
    DwmGeom g;
};

// RELU: this layer's activation (and with BNIN the producer's) is ReLU, compiled in: the activation switch on a kernel
// argument costs a scalar jump table per element otherwise. Other cheap activations take the generic instance.
template <int S, int V, bool BNIN, int PF, bool RELU>
__global__ __launch_bounds__(256) void dwm_fwd_kernel(const DwmFwdArgs a) {
    __shared__ float red[4][2 * kSlab];
    const int act = RELU ? BCNN_HIP_ACT_RELU : a.act, in_act = RELU ? BCNN_HIP_ACT_RELU : a.in.act;
    constexpr int OV = S == 1 ? V : V / 2;  // outputs per lane and row
    const DwmLane m = dwm_lane(a.g.L, a.g.G, a.g.BPP, a.C, (unsigned)a.g.bands);
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = a.w[m.c * 9 + i];
    float b = a.bias[m.c];
    if (b == 0.0f || b == 1.0f) b = -0.0f;  // bcnn_add_bias quirk: no add for 0 and 1 (v + -0 == v for every v)
    DwmBnInC kin;
    if (BNIN) kin = dwm_bnin_consts(a.in, m.c);
    const unsigned xrow = (unsigned)a.W * 4u, yrow = (unsigned)a.OW * 4u;  // bytes per row
    const unsigned xbase = ((unsigned)m.p * a.H * a.W + m.cg * V) * 4u;
    const int r0 = m.bi * a.g.len, r1 = m.on ? min(r0 + a.g.len, a.OH) : r0;  // output rows [r0, r1)
    unsigned yo = (((unsigned)m.p * a.OH + r0) * a.OW + m.cg * OV) * 4u;
    float s1 = 0.f, s2 = 0.f;
    auto in_row = [&](int r) -> bool { return m.on && r >= 0 && r < a.H; };
    auto fetch = [&](int r) -> Vals<V> { return dwm_ld<V>(a.x, xbase + (unsigned)r * xrow, in_row(r)); };
    auto prep = [&](Vals<V> v, int r) -> Row<V> {
        if (BNIN && in_row(r)) v = dwm_bnin_v<V>(v, kin, in_act);
        return dwm_row<V, true, S == 1>(v, m);
    };
    auto finish = [&](float v) -> float { return act_fwd_cheap(__fadd_rn(v, b), act, 0.f); };
    auto emit = [&](const Row<V>& A, const Row<V>& B, const Row<V>& Cr, bool valid) {
        Vals<OV> o;
#pragma unroll
        for (int c = 0; c < OV; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[kw], A.v[S * c + kw]));
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[3 + kw], B.v[S * c + kw]));
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[6 + kw], Cr.v[S * c + kw]));
            o.v[c] = finish(acc);
        }
        if (valid) {
            dwm_st<OV>(a.y, yo, o);
#pragma unroll
            for (int c = 0; c < OV; ++c) {
                s1 += o.v[c];
                s2 = __fmaf_rn(o.v[c], o.v[c], s2);
            }
        }
        yo += yrow;
    };
    if constexpr (S == 1) {
        Row<V> A = prep(fetch(r0 - 1), r0 - 1), B = prep(fetch(r0), r0);
        Vals<V> ring[PF];  // rows r + 1 .. r + PF, loaded ahead of their use; nothing behind the band's halo row r1
#pragma unroll
        for (int u = 0; u < PF; ++u) ring[u] = fetch(r0 + 1 + u <= r1 ? r0 + 1 + u : -1);
        for (int i0 = 0; i0 < a.g.len; i0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int r = r0 + i0 + u;
                const Vals<V> cur = ring[u];
                ring[u] = fetch(r + 1 + PF <= r1 ? r + 1 + PF : -1);
                const Row<V> Cr = prep(cur, r + 1 <= r1 ? r + 1 : -1);
                emit(A, B, Cr, r < r1);
                A = B;
                B = Cr;
            }
        }
    } else {
        // output row r reads input rows 2r - 1, 2r, 2r + 1; the lane's outputs are columns OV cg ..
        Row<V> A = prep(fetch(2 * r0 - 1), 2 * r0 - 1);
        Vals<V> ring0[PF], ring1[PF];  // input rows 2 r, 2 r + 1 of output rows r .. r + PF - 1
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            ring0[u] = fetch(r0 + u < r1 ? 2 * (r0 + u) : -1);
            ring1[u] = fetch(r0 + u < r1 ? 2 * (r0 + u) + 1 : -1);
        }
        for (int i0 = 0; i0 < a.g.len; i0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int r = r0 + i0 + u;
                const bool valid = r < r1;
                const Vals<V> c0 = ring0[u], c1 = ring1[u];
                const bool more = r + PF < r1;
                ring0[u] = fetch(more ? 2 * (r + PF) : -1);
                ring1[u] = fetch(more ? 2 * (r + PF) + 1 : -1);
                const Row<V> B = prep(c0, valid ? 2 * r : -1), Cr = prep(c1, valid ? 2 * r + 1 : -1);
                emit(A, B, Cr, valid);
                A = Cr;
            }
        }
    }
    if (!a.stats) return;
    const float sv[2] = {s1, s2};
    const int splits = (int)(a.g.bands / a.C);  // N * BPP
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    dwm_band_sums<2>(sv, red[threadIdx.x >> 6], a.g.L, a.g.G, [&](int q, int i, float t) {
        const unsigned band = wave * (unsigned)a.g.G + (unsigned)q;
        if (band >= (unsigned)a.g.bands) return;
        const int p = (int)(band / (unsigned)a.g.BPP), bi = (int)(band - (unsigned)p * a.g.BPP);
        const int n = p / a.C, c = p - n * a.C;
        a.stats[((size_t)c * splits + (size_t)n * a.g.BPP + bi) * 2 + i] = t;
    });
}

// ================================================================================================
// backward
// ================================================================================================
constexpr int kDwmPart = 12;  // partial layout of depthwise_lds.hip's finalize: nine taps, bias sum, two unused

struct DwmBwdArgs {
    const float* x;
    const float* w;
    const float* y;
    float* dy;        // read (no batch-norm), written back when write_back
    float* dx;
    float* partials;  // [C][splits][12]
    float* in_sums;   // optional: [C][splits][2] backward sums of the producer's batch-norm
    DwBnBwd bn;
    DwBnIn in;
    float fM, rfM;
    int C, H, W, OH, OW, act, overwrite, write_back;
    DwmGeom g;
};

struct DwmBnC {
    float mean, sc, dm_m, dv2;
    BnDiv rs;
    bool sc0, any_sc0;
};

template <int S, int V, bool BN, bool BNIN, bool RELU>
__global__ __launch_bounds__(256, DWM_BWD_WAVES) void dwm_bwd_kernel(const DwmBwdArgs a) {
    __shared__ float red[4][kDwmPart * kSlab];
    const int act = RELU ? BCNN_HIP_ACT_RELU : a.act, in_act = RELU ? BCNN_HIP_ACT_RELU : a.in.act;
    constexpr int GV = S == 1 ? V : V / 2;  // gradient values per lane and row
    const DwmLane m = dwm_lane(a.g.L, a.g.G, a.g.BPP, a.C, (unsigned)a.g.bands);
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = a.w[m.c * 9 + i];
    DwmBnC kb;
    if (BN) {
        kb.mean = a.bn.mean[m.c];
        kb.rs.d = sqrtf(a.bn.var[m.c] + 0.00001f);
        kb.rs.r = __fdiv_rn(1.0f, kb.rs.d);
        kb.sc = a.bn.scale[m.c];
        kb.sc0 = kb.sc == 0.0f;
        kb.any_sc0 = __builtin_amdgcn_ballot_w64(kb.sc0) != 0;
        kb.dm_m = __fdiv_rn(a.bn.dmean[m.c], a.fM);
        kb.dv2 = __fmul_rn(a.bn.dvar[m.c], 2.0f);
    }
    DwmBnInC kin;
    if (BNIN) kin = dwm_bnin_consts(a.in, m.c);
    const BnDiv fM{a.fM, a.rfM};
    const bool sums = BNIN && a.in_sums != nullptr;
    const unsigned xrow = (unsigned)a.W * 4u, grow = (unsigned)a.OW * 4u;  // bytes per row
    const unsigned xbase = ((unsigned)m.p * a.H * a.W + m.cg * V) * 4u, gbase = ((unsigned)m.p * a.OH * a.OW + m.cg * GV) * 4u;
    const float* gsrc = BN ? a.bn.dz : a.dy;
    const bool need_y = BN || act != BCNN_HIP_ACT_NONE;
    const bool wb = !BN && a.write_back && act != BCNN_HIP_ACT_NONE;
    // gradient rows [r0, r1) are the band's own; stride 1: the same rows of x / dx, stride 2: x / dx rows [2 r0, min(2 r1, H))
    const int r0 = m.bi * a.g.len, r1 = m.on ? min(r0 + a.g.len, a.OH) : r0;
    float acc[kDwmPart];
#pragma unroll
    for (int i = 0; i < kDwmPart; ++i) acc[i] = 0.f;

    auto g_row_ok = [&](int r) -> bool { return m.on && r >= 0 && r < a.OH; };
    // bn_bwd_one of bn_math.h (bcnn_batchnorm_layer.c:292-296) with the per-channel constants folded, then act'(y)
    auto gval = [&](float gin, float yv) -> float {
        float g = gin;
        if (BN) {
            g = __fmul_rn(g, kb.sc);
            if (kb.any_sc0 && kb.sc0) g = 0.f;
            const float t1 = bn_div(g, kb.rs);
            const float t2 = bn_div(__fmul_rn(kb.dv2, __fsub_rn(yv, kb.mean)), fM);
            g = __fadd_rn(__fadd_rn(t1, t2), kb.dm_m);
        }
        if (act != BCNN_HIP_ACT_NONE) g *= act_bwd_cheap(yv, act, 0.f);
        return g;
    };
    // the producer's activation passes this element (its derivative is 0 or 1: none / ReLU)
    auto passes = [&](float y_in) -> bool { return act_bwd_cheap(y_in, in_act, 0.f) != 0.f; };
    struct Raw {
        Vals<GV> g, y;
    };
    auto fetch_g = [&](int r) -> Raw {
        Raw q;
        const bool ok = g_row_ok(r);
        const unsigned off = gbase + (unsigned)r * grow;
        q.g = dwm_ld<GV>(gsrc, off, ok);
        q.y = dwm_ld<GV>(a.y, off, ok && need_y);
        return q;
    };
    auto make_g = [&](const Raw& q, int r) -> Row<GV> {
        Vals<GV> g;
#pragma unroll
        for (int i = 0; i < GV; ++i) g.v[i] = 0.f;
        if (g_row_ok(r)) {
#pragma unroll
            for (int i = 0; i < GV; ++i) g.v[i] = gval(q.g.v[i], q.y.v[i]);
            if (wb && r >= r0 && r < r1) dwm_st<GV>(a.dy, gbase + (unsigned)r * grow, g);
        }
        return dwm_row<GV, S == 1, true>(g, m);
    };
    // the sums of the producer's batch-norm backward over what this lane just stored
    auto in_sums = [&](const Vals<V>& d, const Vals<V>& y_in, const Vals<V>& raw) {
#pragma unroll
        for (int c = 0; c < V; ++c) {
            const float gi = passes(y_in.v[c]) ? d.v[c] : 0.f;
            acc[10] += gi;
            acc[11] = __fmaf_rn(gi, raw.v[c] - kin.mean, acc[11]);
        }
    };

    if constexpr (S == 1) {
        Row<V> A = make_g(fetch_g(r0 - 1), r0 - 1), B = make_g(fetch_g(r0), r0);
        Raw gn = fetch_g(r0 + 1);
        unsigned xo = xbase + (unsigned)r0 * xrow;
        Vals<V> xn = dwm_ld<V>(a.x, xo, m.on && r0 < r1);
        for (int i = 0; i < a.g.len; ++i) {
            const int r = r0 + i;
            const bool valid = r < r1;
            const Raw gc = gn;
            const Vals<V> xraw = xn;
            gn = fetch_g(r + 2 <= r1 ? r + 2 : -1);
            xn = dwm_ld<V>(a.x, xo + xrow, m.on && r + 1 < r1);
            const Row<V> Cr = make_g(gc, r + 1 <= r1 ? r + 1 : -1);
            if (valid) {
                Vals<V> xv = xraw;
                if (BNIN) xv = dwm_bnin_v<V>(xraw, kin, in_act);
                // weight gradient from the rows of x this band owns: x[r][j] meets g[r - kh + 1][j - kw + 1]
#pragma unroll
                for (int c = 0; c < V; ++c) {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        acc[0 + kw] = __fmaf_rn(xv.v[c], Cr.v[c + 2 - kw], acc[0 + kw]);
                        acc[3 + kw] = __fmaf_rn(xv.v[c], B.v[c + 2 - kw], acc[3 + kw]);
                        acc[6 + kw] = __fmaf_rn(xv.v[c], A.v[c + 2 - kw], acc[6 + kw]);
                    }
                    acc[9] += B.v[c + 1];
                }
                // data gradient, taps in the reference's scatter order: descending kh, descending kw
                Vals<V> d = dwm_ld<V>(a.dx, xo, !a.overwrite);
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    float v = d.v[c];
#pragma unroll
                    for (int kw = 2; kw >= 0; --kw) v = __fadd_rn(v, __fmul_rn(w[6 + kw], A.v[c + 2 - kw]));
#pragma unroll
                    for (int kw = 2; kw >= 0; --kw) v = __fadd_rn(v, __fmul_rn(w[3 + kw], B.v[c + 2 - kw]));
#pragma unroll
                    for (int kw = 2; kw >= 0; --kw) v = __fadd_rn(v, __fmul_rn(w[0 + kw], Cr.v[c + 2 - kw]));
                    d.v[c] = v;
                }
                dwm_st<V>(a.dx, xo, d);
                if (sums) in_sums(d, xv, xraw);
            }
            xo += xrow;
            A = B;
            B = Cr;
        }
    } else {
        // a gradient row: the lane's GV values and (index GV + 1 of the row) the right neighbour's first
        Row<GV> B = make_g(fetch_g(r0), r0);
        Raw gn = fetch_g(r0 + 1);
        unsigned xo = xbase + (unsigned)(2 * r0) * xrow;
        Vals<V> xn0 = dwm_ld<V>(a.x, xo, m.on && r0 < r1), xn1 = dwm_ld<V>(a.x, xo + xrow, m.on && r0 < r1 && 2 * r0 + 1 < a.H);
        for (int i = 0; i < a.g.len; ++i) {
            const int r = r0 + i;
            const bool valid = r < r1, odd_ok = valid && 2 * r + 1 < a.H;
            const Raw gc = gn;
            const Vals<V> xr0 = xn0, xr1 = xn1;
            gn = fetch_g(r + 2 <= r1 ? r + 2 : -1);
            const bool more = m.on && r + 1 < r1;
            xn0 = dwm_ld<V>(a.x, xo + 2 * xrow, more);
            xn1 = dwm_ld<V>(a.x, xo + 3 * xrow, more && 2 * r + 3 < a.H);
            const Row<GV> Cg = make_g(gc, r + 1 <= r1 ? r + 1 : -1);
            if (valid) {
                Vals<V> x0 = xr0, x1 = xr1;
                if (BNIN) x0 = dwm_bnin_v<V>(xr0, kin, in_act);
                if (BNIN && odd_ok) x1 = dwm_bnin_v<V>(xr1, kin, in_act);
                Vals<V> d0 = dwm_ld<V>(a.dx, xo, !a.overwrite), d1 = dwm_ld<V>(a.dx, xo + xrow, !a.overwrite && odd_ok);
#pragma unroll
                for (int j = 0; j < GV; ++j) {
                    // the 2 x 2 input block under gradient column j: g00 = g[r][j], g01 = g[r][j + 1], g10 / g11 one row down
                    const float g00 = B.v[1 + j], g01 = B.v[2 + j], g10 = Cg.v[1 + j], g11 = Cg.v[2 + j];
                    const float xe0 = x0.v[2 * j], xe1 = x0.v[2 * j + 1], xo0 = x1.v[2 * j], xo1 = x1.v[2 * j + 1];
                    // weight gradient from the owned x: even row meets kh = 1 of g[r]; odd row kh = 2 of g[r] and kh = 0 of
                    // g[r + 1]; even columns meet kw = 1, odd columns kw = 0 (gradient column to the right) and kw = 2
                    acc[4] = __fmaf_rn(xe0, g00, acc[4]);
                    acc[3] = __fmaf_rn(xe1, g01, acc[3]);
                    acc[5] = __fmaf_rn(xe1, g00, acc[5]);
                    acc[7] = __fmaf_rn(xo0, g00, acc[7]);
                    acc[6] = __fmaf_rn(xo1, g01, acc[6]);
                    acc[8] = __fmaf_rn(xo1, g00, acc[8]);
                    acc[1] = __fmaf_rn(xo0, g10, acc[1]);
                    acc[0] = __fmaf_rn(xo1, g11, acc[0]);
                    acc[2] = __fmaf_rn(xo1, g10, acc[2]);
                    acc[9] += g00;
                    // data gradient: the four parity classes meet 1, 2, 2 and 4 taps, in the reference's scatter order
                    d0.v[2 * j] = __fadd_rn(d0.v[2 * j], __fmul_rn(w[4], g00));
                    d0.v[2 * j + 1] = __fadd_rn(__fadd_rn(d0.v[2 * j + 1], __fmul_rn(w[5], g00)), __fmul_rn(w[3], g01));
                    d1.v[2 * j] = __fadd_rn(__fadd_rn(d1.v[2 * j], __fmul_rn(w[7], g00)), __fmul_rn(w[1], g10));
                    d1.v[2 * j + 1] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(d1.v[2 * j + 1], __fmul_rn(w[8], g00)),
                                                                    __fmul_rn(w[6], g01)), __fmul_rn(w[2], g10)), __fmul_rn(w[0], g11));
                }
                dwm_st<V>(a.dx, xo, d0);
                if (odd_ok) dwm_st<V>(a.dx, xo + xrow, d1);
                if (sums) {
                    in_sums(d0, x0, xr0);
                    if (odd_ok) in_sums(d1, x1, xr1);
                }
            }
            xo += 2 * xrow;
            B = Cg;
        }
    }
    const int splits = (int)(a.g.bands / a.C);  // N * BPP
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    dwm_band_sums<kDwmPart>(acc, red[threadIdx.x >> 6], a.g.L, a.g.G, [&](int q, int i, float t) {
        const unsigned band = wave * (unsigned)a.g.G + (unsigned)q;
        if (band >= (unsigned)a.g.bands) return;
        const int p = (int)(band / (unsigned)a.g.BPP), bi = (int)(band - (unsigned)p * a.g.BPP);
        const int n = p / a.C, c = p - n * a.C;
        const size_t slot = (size_t)c * splits + (size_t)n * a.g.BPP + bi;
        if (i < 10) a.partials[slot * kDwmPart + i] = t;
        else if (sums) a.in_sums[slot * 2 + (i - 10)] = t;
    });
}

// every pointer a kernel touches with V-float accesses must be V * 4 byte aligned (planes and rows then are)
inline bool dwm_aligned(int V, std::initializer_list<const void*> ptrs) {
    for (const void* p : ptrs)
        if (p && (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(V * 4 - 1))) return false;
    return true;
}

}  // namespace

bool depthwise_march_ok(const DwShape& s) {
    static const int on = BCNN_EXP_ENV("BCNN_HIP_NO_DW_MARCH") ? 0 : 1;  // A/B switch (experiment build only)
    if (!on || !dwm_shape_ok(s)) return false;
    const DwmGeom g = dwm_plan(s);
    return g.bands < 0x7fffffffLL;
}

// slots per channel of the statistics / sums / weight-gradient partials (N * bands per plane)
size_t depthwise_march_splits(const DwShape& s) {
    if (!depthwise_march_ok(s)) return 0;
    const DwmGeom g = dwm_plan(s);
    return (size_t)s.N * g.BPP;
}

bool depthwise_forward_march(const float* x, const float* w, const float* bias, float* y, const DwShape& s, int act,
                             ConvStats* stats, const DwBnIn* in) {
    if (!depthwise_march_ok(s) || !act_is_cheap(act) || act == BCNN_HIP_ACT_PRELU) return false;
    if (in && (!in->mean || !act_is_cheap(in->act) || in->act == BCNN_HIP_ACT_PRELU)) return false;
    DwmFwdArgs a;
    a.g = dwm_plan(s);
    if (!dwm_aligned(a.g.V, {x}) || !dwm_aligned(s.stride == 1 ? a.g.V : a.g.V / 2, {y})) return false;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = nullptr;
    a.C = s.C; a.H = s.H; a.W = s.W; a.OH = s.OH; a.OW = s.OW; a.act = act;
    const int splits = s.N * a.g.BPP;
    if (stats) {
        stats->splits = 0;
        if (stats->partials && stats->capacity >= (size_t)s.C * splits * 2) {
            a.stats = stats->partials;
            stats->splits = splits;
        }
    }
    a.in = in ? *in : DwBnIn{nullptr, nullptr, nullptr, nullptr, 0};
    const unsigned waves = (unsigned)ceil_div(a.g.bands, a.g.G), blocks = (waves + 3) / 4;
    hipStream_t st = current_stream();
    int pf = 2;
#ifdef BCNN_HIP_EXPERIMENT
    if (const char* e = getenv("BCNN_HIP_DWM_PF")) pf = atoi(e);
#endif
    bool relu = act == BCNN_HIP_ACT_RELU && (!in || in->act == BCNN_HIP_ACT_RELU);
    if (BCNN_EXP_ENV("BCNN_HIP_DWM_NORELU")) relu = false;  // A/B switch (experiment build only)
#ifdef BCNN_HIP_EXPERIMENT
#define DWM_FWD_PF(SV, VV, BV)                                                                      \
    do {                                                                                            \
        if (!relu) dwm_fwd_kernel<SV, VV, BV, 2, false><<<blocks, 256, 0, st>>>(a);                 \
        else if (pf <= 1) dwm_fwd_kernel<SV, VV, BV, 1, true><<<blocks, 256, 0, st>>>(a);           \
        else if (pf == 2) dwm_fwd_kernel<SV, VV, BV, 2, true><<<blocks, 256, 0, st>>>(a);           \
        else dwm_fwd_kernel<SV, VV, BV, 4, true><<<blocks, 256, 0, st>>>(a);                        \
    } while (0)
#else
#define DWM_FWD_PF(SV, VV, BV)                                                                      \
    do {                                                                                            \
        if (!relu) dwm_fwd_kernel<SV, VV, BV, 2, false><<<blocks, 256, 0, st>>>(a);                 \
        else dwm_fwd_kernel<SV, VV, BV, 2, true><<<blocks, 256, 0, st>>>(a);                        \
    } while (0)
    (void)pf;
#endif
#define DWM_FWD(SV, VV)                      \
    do {                                     \
        if (in) DWM_FWD_PF(SV, VV, true);    \
        else DWM_FWD_PF(SV, VV, false);      \
    } while (0)
    if (s.stride == 1) {
        if (a.g.V == 4) DWM_FWD(1, 4);
        else if (a.g.V == 2) DWM_FWD(1, 2);
        else DWM_FWD(1, 1);
    } else {
        if (a.g.V == 4) DWM_FWD(2, 4);
        else DWM_FWD(2, 2);
    }
#undef DWM_FWD
#undef DWM_FWD_PF
    KERNEL_CHECK();
    return true;
}

bool depthwise_backward_march(const float* x, const float* w, const float* y, float* dy, float* dx, float* dw, float* dbias,
                              const DwShape& s, int act, int overwrite, int write_back, const DwBnBwd* bn, const DwBnIn* in,
                              ConvStats* in_sums) {
    if (in_sums) in_sums->splits = 0;
    if (!depthwise_march_ok(s) || !act_bwd_is_cheap(act) || act == BCNN_HIP_ACT_PRELU || !dx) return false;
    if (in && (!in->mean || !act_is_cheap(in->act) || in->act == BCNN_HIP_ACT_PRELU)) return false;
    DwmBwdArgs a;
    a.g = dwm_plan(s);
    if (!dwm_aligned(a.g.V, {x, dx}) || !dwm_aligned(s.stride == 1 ? a.g.V : a.g.V / 2, {y, dy, bn ? bn->dz : nullptr})) return false;
    a.x = x; a.w = w; a.y = y; a.dy = dy; a.dx = dx;
    a.C = s.C; a.H = s.H; a.W = s.W; a.OH = s.OH; a.OW = s.OW; a.act = act;
    a.overwrite = overwrite; a.write_back = write_back;
    a.fM = (float)((long long)s.N * s.OH * s.OW);
    a.rfM = 1.0f / a.fM;  // host division: IEEE, round to nearest
    a.bn = bn ? *bn : DwBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    a.in = in ? *in : DwBnIn{nullptr, nullptr, nullptr, nullptr, 0};
    const int splits = s.N * a.g.BPP;
    a.partials = reduce_scratch((size_t)s.C * splits * kDwmPart);
    a.in_sums = nullptr;
    // the sums of the producer's batch-norm backward are of the COMPLETE gradient: only when this kernel is its sole writer,
    // and for producer activations whose derivative is 0 or 1
    if (in && in_sums && in_sums->partials && overwrite && (in->act == BCNN_HIP_ACT_NONE || in->act == BCNN_HIP_ACT_RELU) &&
        in_sums->capacity >= (size_t)s.C * splits * 2) {
        a.in_sums = in_sums->partials;
        in_sums->splits = splits;
    }
    const unsigned waves = (unsigned)ceil_div(a.g.bands, a.g.G), blocks = (waves + 3) / 4;
    hipStream_t st = current_stream();
    bool relu = act == BCNN_HIP_ACT_RELU && (!in || in->act == BCNN_HIP_ACT_RELU);
    if (BCNN_EXP_ENV("BCNN_HIP_DWM_NORELU")) relu = false;  // A/B switch (experiment build only)
#define DWM_LAUNCH_R(SV, VV, RV)                                                                 \
    do {                                                                                         \
        if (bn && in) dwm_bwd_kernel<SV, VV, true, true, RV><<<blocks, 256, 0, st>>>(a);         \
        else if (bn) dwm_bwd_kernel<SV, VV, true, false, RV><<<blocks, 256, 0, st>>>(a);         \
        else if (in) dwm_bwd_kernel<SV, VV, false, true, RV><<<blocks, 256, 0, st>>>(a);         \
        else dwm_bwd_kernel<SV, VV, false, false, RV><<<blocks, 256, 0, st>>>(a);                \
    } while (0)
#define DWM_LAUNCH(SV, VV)                     \
    do {                                       \
        if (relu) DWM_LAUNCH_R(SV, VV, true);  \
        else DWM_LAUNCH_R(SV, VV, false);      \
    } while (0)
    if (s.stride == 1) {
        if (a.g.V == 4) DWM_LAUNCH(1, 4);
        else if (a.g.V == 2) DWM_LAUNCH(1, 2);
        else DWM_LAUNCH(1, 1);
    } else {
        if (a.g.V == 4) DWM_LAUNCH(2, 4);
        else DWM_LAUNCH(2, 2);
    }
#undef DWM_LAUNCH
#undef DWM_LAUNCH_R
    KERNEL_CHECK();
    dwl_finalize_launch(a.partials, splits, s.C, dw, dbias, st);
    return true;
}

}  // namespace bcnn_hip
