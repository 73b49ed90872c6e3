// depthwise_march.hip -- 3x3 depthwise convolution (pad 1, stride 1 or 2) for rows that are a multiple of 16 bytes:
// every lane owns a 16-byte column group of a plane and MARCHES down a band of rows with a three-row register window.
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
//   * all global traffic is 16-byte (stride 2 outputs: 8-byte) per lane with the lanes of a row contiguous, loaded one row
//     ahead of its use; every element is read once per band (plus one halo row per band end, an L2 hit);
//   * results go straight from registers to global memory;
//   * lanes of a wave that do not fit a row (64 mod W/4) idle; a wave holds 64 / (W/4) independent bands, which may lie in
//     different planes, so small planes fill waves as well as large ones.
// Tap order and the separate multiply / add roundings are the reference's (forward and data gradient: bit-exact); the
// reductions (weight / bias gradient, batch-norm sums) are two-level in a fixed order: one partial per band, summed
// across the lanes of a band through a per-wave LDS slab in lane order, then over bands in double by the finalize kernels.
#include "bn_math.h"
#include "depthwise.h"

namespace bcnn_hip {

void dwl_finalize_launch(const float* partials, int splits, int C, float* dw, float* dbias, hipStream_t st);  // depthwise_lds.hip
float* reduce_scratch(size_t floats);                                                                          // blas1.hip

namespace {

#ifndef DWM_ROWS
#define DWM_ROWS 14  // rows a band marches (target; the plan evens bands out)
#endif

struct DwmGeom {
    int L;    // lanes per row (W / 4)
    int G;    // bands per wave
    int len;  // rows per band: stride 1 rows of x == rows of y; stride 2 rows of y (two rows of x each)
    int BPP;  // bands per plane
    long long bands;
};

inline bool dwm_shape_ok(const DwShape& s) {
    if (s.ksz != 3 || s.pad != 1 || (s.stride != 1 && s.stride != 2)) return false;
    if (s.N < 1 || s.C < 1 || s.H < 1 || s.W < 4 || (s.W & 3) || s.W > 256) return false;
    if ((long long)s.N * s.C * s.H * s.W >= 0x7fffffffLL) return false;
    return true;
}

inline DwmGeom dwm_plan(const DwShape& s) {
    DwmGeom g;
    g.L = s.W / 4;
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

__device__ __forceinline__ float4 dwm_ld4(const float* p, bool ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float4*>(p);
    return v;
}
__device__ __forceinline__ float2 dwm_ld2(const float* p, bool ok) {
    float2 v = make_float2(0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float2*>(p);
    return v;
}

// a row of the window: four own values between the neighbours' (index 0 = left neighbour ... 5 = right neighbour)
struct Row6 {
    float v[6];
};
__device__ __forceinline__ Row6 dwm_row6(const float4& x, const DwmLane& m) {
    Row6 r;
    r.v[1] = x.x; r.v[2] = x.y; r.v[3] = x.z; r.v[4] = x.w;
    const float l = dwm_from(m.addr_l, x.w), rr = dwm_from(m.addr_r, x.x);
    r.v[0] = m.first ? 0.f : l;
    r.v[5] = m.last ? 0.f : rr;
    return r;
}

// batch-norm + activation of the producing convolution node, applied to what was loaded (bn_one of bn_math.h)
struct DwmBnInC {
    float mean, sc, b;
    BnDiv rs;
};
__device__ __forceinline__ DwmBnInC dwm_bnin_consts(const DwBnIn& in, int c) {
    DwmBnInC k;
    k.mean = in.mean[c];
    k.sc = in.scale[c];
    k.b = in.bias[c];
    k.rs.d = sqrtf(in.var[c] + 0.000001f);
    k.rs.r = __fdiv_rn(1.0f, k.rs.d);
    return k;
}
__device__ __forceinline__ float dwm_bnin(float x, const DwmBnInC& k, int act) {
    float dummy;
    return bn_one(x, k.mean, k.rs, k.sc, k.b, 0, act, &dummy);
}
__device__ __forceinline__ float4 dwm_bnin4(const float4& x, const DwmBnInC& k, int act) {
    return make_float4(dwm_bnin(x.x, k, act), dwm_bnin(x.y, k, act), dwm_bnin(x.z, k, act), dwm_bnin(x.w, k, act));
}

// sum of NV per-lane values over the lanes of each band of the wave, in lane order; band q's totals are handed to
// put(q, value index, total) by lanes 0 .. G * NV - 1 (loop when that exceeds 64). `slab` = the wave's [NV][64] floats.
template <int NV, class Put>
__device__ __forceinline__ void dwm_band_sums(const float (&v)[NV], float* slab, int L, int G, Put put) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) slab[i * 64 + lane] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int idx = lane; idx < G * NV; idx += 64) {
        const int q = idx / NV, i = idx - q * NV;
        const float* src = slab + i * 64 + q * L;
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

template <int S, bool BNIN, int PF>
__global__ __launch_bounds__(256) void dwm_fwd_kernel(const DwmFwdArgs a) {
    __shared__ float red[4][2 * 64];
    const DwmLane m = dwm_lane(a.g.L, a.g.G, a.g.BPP, a.C, (unsigned)a.g.bands);
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = a.w[m.c * 9 + i];
    float b = a.bias[m.c];
    if (b == 0.0f || b == 1.0f) b = -0.0f;  // bcnn_add_bias quirk: no add for 0 and 1 (v + -0 == v for every v)
    DwmBnInC kin;
    if (BNIN) kin = dwm_bnin_consts(a.in, m.c);
    const float* xp = a.x + (size_t)m.p * a.H * a.W + m.cg * 4;
    float* yp = a.y + (size_t)m.p * a.OH * a.OW + m.cg * (S == 1 ? 4 : 2);
    const int r0 = m.bi * a.g.len, r1 = m.on ? min(r0 + a.g.len, a.OH) : r0;  // output rows [r0, r1)
    float s1 = 0.f, s2 = 0.f;
    auto in_row = [&](int r) -> bool { return m.on && r >= 0 && r < a.H; };
    auto fetch = [&](int r) -> float4 { return dwm_ld4(xp + (long long)r * a.W, in_row(r)); };
    auto prep = [&](float4 v, int r) -> Row6 {
        if (BNIN && in_row(r)) v = dwm_bnin4(v, kin, a.in.act);
        return dwm_row6(v, m);
    };
    auto finish = [&](float v) -> float {
        v = __fadd_rn(v, b);
        v = act_fwd_cheap(v, a.act, 0.f);
        return v;
    };
    if (S == 1) {
        Row6 A = prep(fetch(r0 - 1), r0 - 1), B = prep(fetch(r0), r0);
        float4 ring[PF];  // rows r + 1 .. r + PF, loaded ahead of their use; nothing behind the band's halo row r1
#pragma unroll
        for (int u = 0; u < PF; ++u) ring[u] = fetch(r0 + 1 + u <= r1 ? r0 + 1 + u : -1);
        for (int i0 = 0; i0 < a.g.len; i0 += PF) {
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const int r = r0 + i0 + u;
            const bool valid = r < r1;
            const float4 cur = ring[u];
            ring[u] = fetch(r + 1 + PF <= r1 ? r + 1 + PF : -1);
            const Row6 Cr = prep(cur, r + 1 <= r1 ? r + 1 : -1);
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[kw], A.v[c + kw]));
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[3 + kw], B.v[c + kw]));
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[6 + kw], Cr.v[c + kw]));
                o[c] = finish(acc);
            }
            if (valid) {
                *reinterpret_cast<float4*>(yp + (size_t)r * a.OW) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s1 += o[c];
                    s2 = __fmaf_rn(o[c], o[c], s2);
                }
            }
            A = B;
            B = Cr;
          }
        }
    } else {
        // output row r reads input rows 2r - 1, 2r, 2r + 1; the lane's outputs are columns 2 cg, 2 cg + 1
        Row6 A = prep(fetch(2 * r0 - 1), 2 * r0 - 1);
        float4 ring0[PF], ring1[PF];  // input rows 2 r, 2 r + 1 of output rows r .. r + PF - 1
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
            const float4 c0 = ring0[u], c1 = ring1[u];
            const bool more = r + PF < r1;
            ring0[u] = fetch(more ? 2 * (r + PF) : -1);
            ring1[u] = fetch(more ? 2 * (r + PF) + 1 : -1);
            const Row6 B = prep(c0, valid ? 2 * r : -1), Cr = prep(c1, valid ? 2 * r + 1 : -1);
            float o[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[kw], A.v[2 * c + kw]));
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[3 + kw], B.v[2 * c + kw]));
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __fadd_rn(acc, __fmul_rn(w[6 + kw], Cr.v[2 * c + kw]));
                o[c] = finish(acc);
            }
            if (valid) {
                *reinterpret_cast<float2*>(yp + (size_t)r * a.OW) = make_float2(o[0], o[1]);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    s1 += o[c];
                    s2 = __fmaf_rn(o[c], o[c], s2);
                }
            }
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
    float mean, sc, dm_m, dv;
    BnDiv rs;
};

template <int S, bool BN, bool BNIN>
__global__ __launch_bounds__(256) void dwm_bwd_kernel(const DwmBwdArgs a) {
    __shared__ float red[4][kDwmPart * 64];
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
        kb.dm_m = __fdiv_rn(a.bn.dmean[m.c], a.fM);
        kb.dv = a.bn.dvar[m.c];
    }
    DwmBnInC kin;
    if (BNIN) kin = dwm_bnin_consts(a.in, m.c);
    const BnDiv fM{a.fM, a.rfM};
    const bool sums = BNIN && a.in_sums != nullptr;
    constexpr int GV = S == 1 ? 4 : 2;  // gradient values per lane and row
    const size_t xoff = (size_t)m.p * a.H * a.W + m.cg * 4, goff = (size_t)m.p * a.OH * a.OW + m.cg * GV;
    const float* xp = a.x + xoff;
    float* dxp = a.dx + xoff;
    const float* gp = (BN ? a.bn.dz : a.dy) + goff;
    const float* yp = a.y + goff;
    float* gwb = a.dy + goff;
    const bool need_y = BN || a.act != BCNN_HIP_ACT_NONE;
    const bool wb = !BN && a.write_back && a.act != BCNN_HIP_ACT_NONE;
    // gradient rows [r0, r1) are the band's own; stride 1: the same rows of x / dx, stride 2: x / dx rows [2 r0, min(2 r1, H))
    const int r0 = m.bi * a.g.len, r1 = m.on ? min(r0 + a.g.len, a.OH) : r0;
    float acc[kDwmPart];
#pragma unroll
    for (int i = 0; i < kDwmPart; ++i) acc[i] = 0.f;
    float s1 = 0.f, s2 = 0.f;

    auto g_row_ok = [&](int r) -> bool { return m.on && r >= 0 && r < a.OH; };
    auto x_row_ok = [&](int r) -> bool { return m.on && r >= 0 && r < a.H; };
    auto gval = [&](float gin, float yv) -> float {
        float g = gin;
        if (BN) g = bn_bwd_one(gin, 0.f, yv, kb.mean, kb.rs, kb.sc, kb.dm_m, kb.dv, fM, BCNN_HIP_ACT_NONE);
        if (a.act != BCNN_HIP_ACT_NONE) g *= act_bwd_cheap(yv, a.act, 0.f);
        return g;
    };
    // the producer's activation passes this element (its derivative is 0 or 1: none / ReLU)
    auto passes = [&](float y_in) -> bool { return act_bwd_cheap(y_in, a.in.act, 0.f) != 0.f; };

    if (S == 1) {
        struct Raw { float4 g, y; };
        auto fetch_g = [&](int r) -> Raw {
            Raw q;
            const bool ok = g_row_ok(r);
            q.g = dwm_ld4(gp + (long long)r * a.OW, ok);
            q.y = need_y ? dwm_ld4(yp + (long long)r * a.OW, ok) : make_float4(0.f, 0.f, 0.f, 0.f);
            return q;
        };
        auto make_g = [&](const Raw& q, int r) -> Row6 {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_row_ok(r)) {
                g = make_float4(gval(q.g.x, q.y.x), gval(q.g.y, q.y.y), gval(q.g.z, q.y.z), gval(q.g.w, q.y.w));
                if (wb && r >= r0 && r < r1) *reinterpret_cast<float4*>(gwb + (size_t)r * a.OW) = g;
            }
            return dwm_row6(g, m);
        };
        Row6 A = make_g(fetch_g(r0 - 1), r0 - 1), B = make_g(fetch_g(r0), r0);
        Raw gn = fetch_g(r0 + 1);
        float4 xn = dwm_ld4(xp + (long long)r0 * a.W, x_row_ok(r0) && r0 < r1);
        for (int i = 0; i < a.g.len; ++i) {
            const int r = r0 + i;
            const bool valid = r < r1;
            const Raw gc = gn;
            const float4 xraw = xn;
            gn = fetch_g(r + 2 <= r1 ? r + 2 : -1);
            xn = dwm_ld4(xp + (long long)(r + 1) * a.W, m.on && r + 1 < r1);
            const Row6 Cr = make_g(gc, r + 1);
            float4 xv = xraw;
            if (BNIN && valid) xv = dwm_bnin4(xraw, kin, a.in.act);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            if (valid) {
                // weight gradient from the rows of x this band owns: x[r][j] meets g[r - kh + 1][j - kw + 1]
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        acc[0 + kw] = __fmaf_rn(xs[c], Cr.v[c + 2 - kw], acc[0 + kw]);
                        acc[3 + kw] = __fmaf_rn(xs[c], B.v[c + 2 - kw], acc[3 + kw]);
                        acc[6 + kw] = __fmaf_rn(xs[c], A.v[c + 2 - kw], acc[6 + kw]);
                    }
                    acc[9] += B.v[c + 1];
                }
                // data gradient, taps in the reference's scatter order: descending kh, descending kw
                float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!a.overwrite) old = *reinterpret_cast<const float4*>(dxp + (size_t)r * a.W);
                float d[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = d[c];
#pragma unroll
                    for (int kw = 2; kw >= 0; --kw) v = __fadd_rn(v, __fmul_rn(w[6 + kw], A.v[c + 2 - kw]));
#pragma unroll
                    for (int kw = 2; kw >= 0; --kw) v = __fadd_rn(v, __fmul_rn(w[3 + kw], B.v[c + 2 - kw]));
#pragma unroll
                    for (int kw = 2; kw >= 0; --kw) v = __fadd_rn(v, __fmul_rn(w[0 + kw], Cr.v[c + 2 - kw]));
                    d[c] = v;
                }
                *reinterpret_cast<float4*>(dxp + (size_t)r * a.W) = make_float4(d[0], d[1], d[2], d[3]);
                if (sums) {
                    const float raw[4] = {xraw.x, xraw.y, xraw.z, xraw.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float gi = passes(xs[c]) ? d[c] : 0.f;
                        s1 += gi;
                        s2 = __fmaf_rn(gi, raw[c] - kin.mean, s2);
                    }
                }
            }
            A = B;
            B = Cr;
        }
    } else {
        // a row of the gradient: the lane's two values and the right neighbour's first
        struct G3 { float g0, g1, gr; };
        struct Raw { float2 g, y; };
        auto fetch_g = [&](int r) -> Raw {
            Raw q;
            const bool ok = g_row_ok(r);
            q.g = dwm_ld2(gp + (long long)r * a.OW, ok);
            q.y = need_y ? dwm_ld2(yp + (long long)r * a.OW, ok) : make_float2(0.f, 0.f);
            return q;
        };
        auto make_g = [&](const Raw& q, int r) -> G3 {
            float2 g = make_float2(0.f, 0.f);
            if (g_row_ok(r)) {
                g = make_float2(gval(q.g.x, q.y.x), gval(q.g.y, q.y.y));
                if (wb && r >= r0 && r < r1) *reinterpret_cast<float2*>(gwb + (size_t)r * a.OW) = g;
            }
            G3 o;
            o.g0 = g.x; o.g1 = g.y;
            const float rr = dwm_from(m.addr_r, g.x);
            o.gr = m.last ? 0.f : rr;
            return o;
        };
        G3 B = make_g(fetch_g(r0), r0);
        Raw gn = fetch_g(r0 + 1);
        float4 xn0 = dwm_ld4(xp + (long long)(2 * r0) * a.W, x_row_ok(2 * r0) && r0 < r1);
        float4 xn1 = dwm_ld4(xp + (long long)(2 * r0 + 1) * a.W, x_row_ok(2 * r0 + 1) && r0 < r1);
        for (int i = 0; i < a.g.len; ++i) {
            const int r = r0 + i;
            const bool valid = r < r1, odd_ok = valid && 2 * r + 1 < a.H;
            const Raw gc = gn;
            const float4 xr0 = xn0, xr1 = xn1;
            gn = fetch_g(r + 2 <= r1 ? r + 2 : -1);
            const bool more = m.on && r + 1 < r1;
            xn0 = dwm_ld4(xp + (long long)(2 * r + 2) * a.W, more);
            xn1 = dwm_ld4(xp + (long long)(2 * r + 3) * a.W, more && 2 * r + 3 < a.H);
            const G3 Cg = make_g(gc, r + 1);
            float4 x0 = xr0, x1 = xr1;
            if (BNIN && valid) x0 = dwm_bnin4(xr0, kin, a.in.act);
            if (BNIN && odd_ok) x1 = dwm_bnin4(xr1, kin, a.in.act);
            if (valid) {
                // weight gradient from the owned rows of x: even row 2r meets kh = 1 of g[r]; odd row 2r + 1 meets kh = 2 of
                // g[r] and kh = 0 of g[r + 1]; even columns meet kw = 1, odd columns kw = 0 (to the right) and kw = 2
                acc[4] = __fmaf_rn(x0.x, B.g0, acc[4]); acc[4] = __fmaf_rn(x0.z, B.g1, acc[4]);
                acc[3] = __fmaf_rn(x0.y, B.g1, acc[3]); acc[3] = __fmaf_rn(x0.w, B.gr, acc[3]);
                acc[5] = __fmaf_rn(x0.y, B.g0, acc[5]); acc[5] = __fmaf_rn(x0.w, B.g1, acc[5]);
                acc[7] = __fmaf_rn(x1.x, B.g0, acc[7]); acc[7] = __fmaf_rn(x1.z, B.g1, acc[7]);
                acc[6] = __fmaf_rn(x1.y, B.g1, acc[6]); acc[6] = __fmaf_rn(x1.w, B.gr, acc[6]);
                acc[8] = __fmaf_rn(x1.y, B.g0, acc[8]); acc[8] = __fmaf_rn(x1.w, B.g1, acc[8]);
                acc[1] = __fmaf_rn(x1.x, Cg.g0, acc[1]); acc[1] = __fmaf_rn(x1.z, Cg.g1, acc[1]);
                acc[0] = __fmaf_rn(x1.y, Cg.g1, acc[0]); acc[0] = __fmaf_rn(x1.w, Cg.gr, acc[0]);
                acc[2] = __fmaf_rn(x1.y, Cg.g0, acc[2]); acc[2] = __fmaf_rn(x1.w, Cg.g1, acc[2]);
                acc[9] += B.g0;
                acc[9] += B.g1;
                // data gradient of input rows 2r, 2r + 1 (the four parity classes meet 1, 2, 2 and 4 taps)
                float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
                if (!a.overwrite) {
                    o0 = *reinterpret_cast<const float4*>(dxp + (size_t)(2 * r) * a.W);
                    if (odd_ok) o1 = *reinterpret_cast<const float4*>(dxp + (size_t)(2 * r + 1) * a.W);
                }
                float4 d0, d1;
                d0.x = __fadd_rn(o0.x, __fmul_rn(w[4], B.g0));
                d0.y = __fadd_rn(__fadd_rn(o0.y, __fmul_rn(w[5], B.g0)), __fmul_rn(w[3], B.g1));
                d0.z = __fadd_rn(o0.z, __fmul_rn(w[4], B.g1));
                d0.w = __fadd_rn(__fadd_rn(o0.w, __fmul_rn(w[5], B.g1)), __fmul_rn(w[3], B.gr));
                d1.x = __fadd_rn(__fadd_rn(o1.x, __fmul_rn(w[7], B.g0)), __fmul_rn(w[1], Cg.g0));
                d1.y = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(o1.y, __fmul_rn(w[8], B.g0)), __fmul_rn(w[6], B.g1)),
                                           __fmul_rn(w[2], Cg.g0)), __fmul_rn(w[0], Cg.g1));
                d1.z = __fadd_rn(__fadd_rn(o1.z, __fmul_rn(w[7], B.g1)), __fmul_rn(w[1], Cg.g1));
                d1.w = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(o1.w, __fmul_rn(w[8], B.g1)), __fmul_rn(w[6], B.gr)),
                                           __fmul_rn(w[2], Cg.g1)), __fmul_rn(w[0], Cg.gr));
                *reinterpret_cast<float4*>(dxp + (size_t)(2 * r) * a.W) = d0;
                if (odd_ok) *reinterpret_cast<float4*>(dxp + (size_t)(2 * r + 1) * a.W) = d1;
                if (sums) {
                    const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                    const float yy[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    const float raw[8] = {xr0.x, xr0.y, xr0.z, xr0.w, xr1.x, xr1.y, xr1.z, xr1.w};
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if (c >= 4 && !odd_ok) break;
                        const float gi = passes(yy[c]) ? dd[c] : 0.f;
                        s1 += gi;
                        s2 = __fmaf_rn(gi, raw[c] - kin.mean, s2);
                    }
                }
            }
            B = Cg;
        }
    }
    acc[10] = s1;
    acc[11] = s2;
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
#define DWM_FWD(SV, BV)                                                              \
    do {                                                                             \
        if (pf <= 1) dwm_fwd_kernel<SV, BV, 1><<<blocks, 256, 0, st>>>(a);           \
        else if (pf == 2) dwm_fwd_kernel<SV, BV, 2><<<blocks, 256, 0, st>>>(a);      \
        else dwm_fwd_kernel<SV, BV, 4><<<blocks, 256, 0, st>>>(a);                   \
    } while (0)
    if (in) {
        if (s.stride == 1) DWM_FWD(1, true);
        else DWM_FWD(2, true);
    } else {
        if (s.stride == 1) DWM_FWD(1, false);
        else DWM_FWD(2, false);
    }
#undef DWM_FWD
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
#define DWM_LAUNCH(SV)                                                                   \
    do {                                                                                 \
        if (bn && in) dwm_bwd_kernel<SV, true, true><<<blocks, 256, 0, st>>>(a);         \
        else if (bn) dwm_bwd_kernel<SV, true, false><<<blocks, 256, 0, st>>>(a);         \
        else if (in) dwm_bwd_kernel<SV, false, true><<<blocks, 256, 0, st>>>(a);         \
        else dwm_bwd_kernel<SV, false, false><<<blocks, 256, 0, st>>>(a);                \
    } while (0)
    if (s.stride == 1) DWM_LAUNCH(1);
    else DWM_LAUNCH(2);
#undef DWM_LAUNCH
    KERNEL_CHECK();
    dwl_finalize_launch(a.partials, splits, s.C, dw, dbias, st);
    return true;
}

}  // namespace bcnn_hip
