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
//   * a lane's window is V floats per row plus the two neighbour values it takes from the adjacent lanes with
//     ds_bpermute (no LDS storage, no vector-ALU cost); image borders are a select per row, rows outside the image are
//     zeros that never left a register;
//   * all global traffic is one V-float access per lane (stride 2 outputs: V / 2) with the lanes of a row contiguous, loaded
//     one to four rows ahead of its use; every element is read once per band, plus one halo row per band end -- an L2 hit,
//     because the two bands either side of a boundary march towards each other (dwm_map);
//     addresses are a uniform base plus a 32-bit byte offset per lane, advanced by one add per row and stream;
//   * results go straight from registers to global memory;
//   * lanes of a wave that do not fit a row (64 mod W/V) idle; a wave holds 64 / (W/V) independent bands, which may lie in
//     different planes, so small planes (14 x 14: nine per wave, 7 x 7 likewise) fill waves as well as large ones.
// Round 5: every memory instruction is straight-line code on buffer descriptors (dwm_ld); a wave that kept its nine channels
// and marched through several images of them as one band (constants fetched and sums flushed once per run instead of once per
// 14 rows) was built and measured at 2 ... 16 images per wave, before and after that change: never faster (14 x 14 backward
// 128 against 120 us), and removed again.
// Tap order and the separate multiply / add roundings are the reference's (forward and data gradient: bit-exact); the
// reductions (weight / bias gradient, batch-norm sums) are two-level in a fixed order: one partial per band, summed
// across the lanes of a band through a per-wave LDS slab in lane order, then over bands in double by the finalize kernels.
#include "bn_math.h"
#include "depthwise.h"
#include "lds_dma.h"

#include <initializer_list>

namespace bcnn_hip {

void dwl_finalize_launch(const float* partials, int splits, int C, float* dw, float* dbias, hipStream_t st);  // depthwise_lds.hip
float* reduce_scratch(size_t floats);                                                                          // blas1.hip

namespace {

#ifndef DWM_BWD_WAVES
#define DWM_BWD_WAVES 3  // waves per SIMD the backward kernels are compiled for (register budget 512 / that; at 4 the
                         // 112 x 112 instance spills inside its loop and takes twice the time)
#endif
#ifndef DWM_FWD_WAVES
#define DWM_FWD_WAVES 4  // waves per SIMD the forward kernels are compiled for
#endif
#ifndef DWM_BWD_WAVES_SMALL
#define DWM_BWD_WAVES_SMALL 3  // the same for lanes of 2 / 1 columns
#endif
#ifndef DWM_BWD_PF_SMALL
#define DWM_BWD_PF_SMALL 4  // backward, lanes of 2 / 1 columns: rows requested ahead
#endif
#ifndef DWM_BWD_PF_WIDE
#define DWM_BWD_PF_WIDE 1  // backward, lanes of 4 columns: rows requested ahead
#endif
#ifndef DWM_ROWS_SMALL
#define DWM_ROWS_SMALL 7  // the same for planes below 56 rows and for stride 2 (see dwm_plan)
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
    if ((long long)s.N * s.C * s.H * s.W >= 0x1fffffffLL) return false;  // byte offsets below kOOB (2 GiB), lds_dma.h
    return true;
}

inline DwmGeom dwm_plan(const DwShape& s) {
    DwmGeom g;
    g.V = dwm_width(s);
    g.L = s.W / g.V;
    g.G = 64 / g.L;
    const int R = s.OH;
    // band length: 14 rows for stride-1 planes of 56 rows and more, 7 for everything else -- measured per MobileNet layer inside
    // the step (tools/exp/dwm_libs.sh with BCNN_HIP_DWM_ROWS = 5 / 7 / 10 / 14 / 28 / 56): more, shorter waves win on the small
    // planes and at stride 2 (28 x 28 backward 204 -> 195 us, 14 x 14 backward 116 -> 108, 112 -> 56 backward 382 -> 361), the
    // 112 x 112 / 56 x 56 stride-1 layers lose 5-20 % below 14 (the halo rows' share), and everything loses above 14.
    int rows = (s.stride == 1 && s.OH >= 56) ? DWM_ROWS : DWM_ROWS_SMALL;
#ifdef BCNN_HIP_EXPERIMENT
    if (const char* e = getenv("BCNN_HIP_DWM_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
#endif
    g.BPP = ceil_div(R, rows);
    g.len = ceil_div(R, g.BPP);
    g.BPP = ceil_div(R, g.len);
    g.bands = (long long)s.N * s.C * g.BPP;
    return g;
}

// slots per channel of the per-band partial sums
inline int dwm_splits(const DwShape& s, const DwmGeom& g) { return s.N * g.BPP; }

// Which band a lane group works on, and in which direction it marches. Planes cut into several bands (BPP > 1): the waves
// come in pairs -- the even wave takes even bands and marches DOWN, the odd wave takes the odd bands of the same planes and
// marches UP -- so the two bands either side of a boundary reach it at the same time (both at their last step, or both at
// their first) and the halo row one of them reads is the row the other has just read: an L2 hit instead of a second trip to
// HBM (rocprofv3 FETCH_SIZE with every band marching down: 1.20x the algorithmic bytes on the backward kernels, the halo
// rows' 2 / 14 almost in full). The direction is wave-uniform: a scalar branch picks the window's orientation.
struct DwmLane {
    bool on;           // this lane works on a band
    bool up;           // the band is marched from its last row to its first (wave-uniform)
    bool first, last;  // first / last column group of the row
    int cg, grp;
    int p, c, bi;      // plane, channel, band within the plane
    int addr_l, addr_r;  // ds_bpermute byte addresses of the lanes to the left / right
};

// lane group `grp` of wave `wave`: plane and band (false: none). planes = N * C.
__device__ __forceinline__ bool dwm_map(unsigned wave, int grp, int G, int BPP, unsigned planes, int& p, int& bi, bool& up) {
    up = false; p = 0; bi = 0;
    if (grp >= G) return false;
    if (BPP == 1) {
        const unsigned k = wave * (unsigned)G + (unsigned)grp;
        if (k >= planes) return false;
        p = (int)k;
        return true;
    }
    up = (wave & 1u) != 0u;
    const unsigned per = up ? (unsigned)BPP / 2u : ((unsigned)BPP + 1u) / 2u;  // bands of this parity per plane
    const unsigned k = (wave >> 1) * (unsigned)G + (unsigned)grp;
    if (k >= planes * per) return false;
    p = (int)(k / per);
    bi = 2 * (int)(k - (unsigned)p * per) + (up ? 1 : 0);
    return true;
}

__device__ __forceinline__ DwmLane dwm_lane(int L, int G, int BPP, int C, unsigned planes) {
    DwmLane m;
    const int lane = threadIdx.x & 63;
    // (readfirstlane: the compiler cannot see that threadIdx.x >> 6 is wave-uniform, and would run both orientations of the
    // window under exec masks, each with its own copy of the step's stores)
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    m.grp = lane / L;
    m.cg = lane - m.grp * L;
    m.on = dwm_map(wave, m.grp, G, BPP, planes, m.p, m.bi, m.up);
    m.up = BPP > 1 && (wave & 1u) != 0u;  // also for lanes without a band: the branch on it has to be wave-uniform
    m.c = m.p % C;
    m.first = m.cg == 0;
    m.last = m.cg == L - 1;
    m.addr_l = ((lane + 63) & 63) << 2;
    m.addr_r = ((lane + 1) & 63) << 2;
    return m;
}

// waves a launch needs
inline unsigned dwm_waves(const DwmGeom& g, long long planes) {
    if (g.BPP == 1) return (unsigned)ceil_div(planes, g.G);
    return 2u * (unsigned)ceil_div(planes * ((g.BPP + 1) / 2), g.G);
}

__device__ __forceinline__ float dwm_from(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// V floats of a row, through raw buffer descriptors (lds_dma.h): a lane that must not read / write hands in an out-of-range
// offset instead of sitting out a branch -- it gets 0.0, its store is dropped. That is not about the branch's cost: every
// memory instruction under divergent control flow is given a skip branch by the compiler, and with a skipped request on some
// path its waitcnt pass can no longer count how many requests are younger than the one it needs; it then waits for ALL of
// them (`s_waitcnt vmcnt(0)`), which is what these kernels did in every step until round 5: rows "requested PF steps ahead"
// were drained at the next use of any loaded value. Straight-line requests are counted (vmcnt(N)), the lead is real.
template <int V>
struct Vals {
    float v[V];
};
// 16-byte store: inline assembly carrying its own wait states (lds_dma.h explains why there is no intrinsic wrapper)
__device__ __forceinline__ void buffer_store_f32x4(const buf_f32x4 v, rsrc_i4 rs, unsigned voff) {
#ifdef NT_STORES
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rs) : "memory");
#else
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rs) : "memory");
#endif
}
template <int V>
__device__ __forceinline__ Vals<V> dwm_ld(rsrc_i4 rs, unsigned byte_off, bool ok) {
    Vals<V> o;
    const int vo = (int)(ok ? byte_off : kOOB);
    if constexpr (V == 4) {
        const buf_f32x4 t = buffer_load_f32x4(rs, vo, 0, 0);
        o.v[0] = t[0]; o.v[1] = t[1]; o.v[2] = t[2]; o.v[3] = t[3];
    } else if constexpr (V == 2) {
        const buf_f32x2 t = buffer_load_f32x2(rs, vo, 0, 0);
        o.v[0] = t[0]; o.v[1] = t[1];
    } else {
        o.v[0] = buffer_load_f32(rs, vo, 0, 0);
    }
    return o;
}
// V == 0 (compile-time "never"): nothing requested
template <int V>
__device__ __forceinline__ Vals<V> dwm_zero() {
    Vals<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = 0.f;
    return o;
}
#ifdef NT_STORES
#define DWM_STORE_AUX 2
#else
#define DWM_STORE_AUX 0
#endif
template <int V>
__device__ __forceinline__ void dwm_st(rsrc_i4 rs, unsigned byte_off, const Vals<V>& o, bool ok) {
    const unsigned vo = ok ? byte_off : kOOB;
    if constexpr (V == 4) buffer_store_f32x4(buf_f32x4{o.v[0], o.v[1], o.v[2], o.v[3]}, rs, vo);
    else if constexpr (V == 2) buffer_store_f32x2(buf_f32x2{o.v[0], o.v[1]}, rs, (int)vo, 0, DWM_STORE_AUX);
    else buffer_store_f32(o.v[0], rs, (int)vo, 0, DWM_STORE_AUX);
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
    int N, C, H, W, OH, OW, act;
    unsigned xbytes, ybytes;  // sizes of x and y (buffer descriptors)
    DwmGeom g;
};

// RELU: this layer's activation (and with BNIN the producer's) is ReLU, compiled in: the activation switch on a kernel
// argument costs a scalar jump table per element otherwise. Other cheap activations take the generic instance.
// (launch bound: four waves per SIMD, i.e. up to 128 registers. Compiled for the default of eight the allocator split the ring's
// live ranges and copied three of four slots at the loop's back edge -- copies of loaded values, i.e. a full drain per PF steps.)
template <int S, int V, bool BNIN, int PF, bool RELU>
__global__ __launch_bounds__(256, DWM_FWD_WAVES) void dwm_fwd_kernel(const DwmFwdArgs a) {
    __shared__ float red[4][2 * kSlab];
    const int act = RELU ? BCNN_HIP_ACT_RELU : a.act, in_act = RELU ? BCNN_HIP_ACT_RELU : a.in.act;
    constexpr int OV = S == 1 ? V : V / 2;  // outputs per lane and row
    const unsigned planes = (unsigned)(a.g.bands / a.g.BPP);
    const DwmLane m = dwm_lane(a.g.L, a.g.G, a.g.BPP, a.C, planes);
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
    // step k of the march is output row rf + dir * k, k < nrows
    const int nrows = r1 - r0, dir = m.up ? -1 : 1, rf = m.up ? r1 - 1 : r0;
    unsigned yo = (((unsigned)m.p * a.OH + rf) * a.OW + m.cg * OV) * 4u;
    const unsigned ystep = m.up ? 0u - yrow : yrow;
    float s1 = 0.f, s2 = 0.f;
    const rsrc_i4 rx = make_rsrc(a.x, a.xbytes), ry = make_rsrc(a.y, a.ybytes);
    auto in_row = [&](int r) -> bool { return m.on && r >= 0 && r < a.H; };
    auto fetch = [&](int r) -> Vals<V> { return dwm_ld<V>(rx, xbase + (unsigned)r * xrow, in_row(r)); };
    // (The window row is always made of NEW registers -- a select, or a move the compiler cannot fold. A row that merely renames
    // the loaded registers keeps them alive for the two steps it spends in the window, the slot's next request then lands
    // elsewhere, and the ring is rotated by copies at the loop's back edge: copies of loaded values, a drain per PF steps.)
    auto prep = [&](const Vals<V>& v, int r) -> Row<V> {
        Vals<V> o;
        if (BNIN) {
            const bool ok = in_row(r);
            const Vals<V> t = dwm_bnin_v<V>(v, kin, in_act);
#pragma unroll
            for (int i = 0; i < V; ++i) o.v[i] = ok ? t.v[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(o.v[i]) : "v"(v.v[i]));
        }
        return dwm_row<V, true, S == 1>(o, m);
    };
    auto finish = [&](float v) -> float { return act_fwd_cheap(__fadd_rn(v, b), act, 0.f); };
    // A / B / Cr: the rows above, at and below the output row (in image order, whichever way the band is marched)
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
        dwm_st<OV>(ry, yo, o, valid);
        if (valid) {
#pragma unroll
            for (int c = 0; c < OV; ++c) {
                s1 += o.v[c];
                s2 = __fmaf_rn(o.v[c], o.v[c], s2);
            }
        }
        yo += ystep;
    };
    if constexpr (S == 1) {
        // P: the row behind the march, Q: the output row's own input row, Nx: the row ahead (loaded PF steps before its use;
        // nothing is requested beyond the band's far halo row, step nrows)
        auto step_row = [&](int st) -> int { return st <= nrows ? rf + dir * st : -1; };
        Row<V> P = prep(fetch(rf - dir), rf - dir), Q = prep(fetch(step_row(0)), step_row(0));
        Vals<V> ring[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) ring[u] = fetch(step_row(1 + u));
        for (int i0 = 0; i0 < a.g.len; i0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int k = i0 + u;
                const Row<V> Nx = prep(ring[u], step_row(k + 1));
                ring[u] = fetch(step_row(k + 1 + PF));  // right after the slot's last use: into the same registers
                if (m.up) emit(Nx, Q, P, k < nrows);
                else emit(P, Q, Nx, k < nrows);
                P = Q;
                Q = Nx;
            }
        }
    } else {
        // output row r reads input rows 2r - 1, 2r, 2r + 1: per step the row 2r and the FAR row 2r + dir are new, the NEAR
        // row 2r - dir is the previous step's far row. The lane's outputs are columns OV cg ..
        Row<V> Nr = prep(fetch(m.on ? 2 * rf - dir : -1), m.on ? 2 * rf - dir : -1);
        Vals<V> ring0[PF], ring1[PF];  // rows 2r and 2r + dir of steps k .. k + PF - 1
        auto mid_row = [&](int st) -> int { return st < nrows ? 2 * (rf + dir * st) : -1; };
        auto far_row = [&](int st) -> int { return st < nrows ? 2 * (rf + dir * st) + dir : -1; };
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            ring0[u] = fetch(mid_row(u));
            ring1[u] = fetch(far_row(u));
        }
        for (int i0 = 0; i0 < a.g.len; i0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int k = i0 + u;
                const Row<V> Md = prep(ring0[u], mid_row(k)), Fr = prep(ring1[u], far_row(k));
                ring0[u] = fetch(mid_row(k + PF));  // right after the slots' last use
                ring1[u] = fetch(far_row(k + PF));
                if (m.up) emit(Fr, Md, Nr, k < nrows);
                else emit(Nr, Md, Fr, k < nrows);
                Nr = Fr;
            }
        }
    }
    if (!a.stats) return;
    const float sv[2] = {s1, s2};
    const int splits = (int)(a.g.bands / a.C);  // N * BPP
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    dwm_band_sums<2>(sv, red[threadIdx.x >> 6], a.g.L, a.g.G, [&](int q, int i, float t) {
        int p, bi;
        bool up;
        if (!dwm_map(wave, q, a.g.G, a.g.BPP, planes, p, bi, up)) return;
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
    int N, C, H, W, OH, OW, act, overwrite, write_back;
    unsigned xbytes, ybytes;  // sizes of x / dx and of y / dy / dz (buffer descriptors)
    DwmGeom g;
};

struct DwmBnC {
    float mean, sc, dm_m, dv2;
    BnDiv rs;
    bool sc0, any_sc0;
};

// PF: rows requested ahead of their use. 1 where a lane's rows are 16 bytes wide (the window already fills the register
// budget of three waves per SIMD); small planes -- whole 14 x 14 / 7 x 7 planes per band, 8- and 4-byte rows, a few hundred
// vector instructions per step -- were latency bound at 1 (waves 78 % waiting, 2.7 TB/s): they request PF = 4 rows ahead.
// OVW: dx is written, not added to (the executor's no-fill mode), compiled in: as a run-time flag the skipped read of dx still
// left its `s_waitcnt vmcnt(0)` in every step, which drained the rows requested ahead -- the kernels ran with no lead at all
// whatever PF said (round 5; found in the ISA after no ablation of arithmetic, exchanges or stores moved the time).
template <int S, int V, bool BN, bool BNIN, bool RELU, int PF, bool OVW>
__global__ __launch_bounds__(256, V == 4 ? DWM_BWD_WAVES : DWM_BWD_WAVES_SMALL) void dwm_bwd_kernel(const DwmBwdArgs a) {
    __shared__ float red[4][kDwmPart * kSlab];
    const int act = RELU ? BCNN_HIP_ACT_RELU : a.act, in_act = RELU ? BCNN_HIP_ACT_RELU : a.in.act;
    constexpr int GV = S == 1 ? V : V / 2;  // gradient values per lane and row
    const unsigned planes = (unsigned)(a.g.bands / a.g.BPP);
    const DwmLane m = dwm_lane(a.g.L, a.g.G, a.g.BPP, a.C, planes);
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = a.w[m.c * 9 + i];
#define DWM_CONST(expr, val) (expr)
    DwmBnC kb;
    if (BN) {
        kb.mean = DWM_CONST(a.bn.mean[m.c], 0.1f);
        kb.rs.d = DWM_CONST(sqrtf(a.bn.var[m.c] + 0.00001f), 1.25f);
        kb.rs.r = DWM_CONST(__fdiv_rn(1.0f, kb.rs.d), 0.8f);
        kb.sc = DWM_CONST(a.bn.scale[m.c], 1.5f);
        kb.sc0 = kb.sc == 0.0f;
        kb.any_sc0 = __builtin_amdgcn_ballot_w64(kb.sc0) != 0;
        kb.dm_m = DWM_CONST(__fdiv_rn(a.bn.dmean[m.c], a.fM), 0.001f);
        kb.dv2 = DWM_CONST(__fmul_rn(a.bn.dvar[m.c], 2.0f), 0.002f);
    }
    DwmBnInC kin;
    if (BNIN) kin = dwm_bnin_consts(a.in, m.c);
#undef DWM_CONST
    const BnDiv fM{a.fM, a.rfM};
    const bool sums = BNIN && a.in_sums != nullptr;
    const unsigned xrow = (unsigned)a.W * 4u, grow = (unsigned)a.OW * 4u;  // bytes per row
    const unsigned xbase = ((unsigned)m.p * a.H * a.W + m.cg * V) * 4u, gbase = ((unsigned)m.p * a.OH * a.OW + m.cg * GV) * 4u;
    const float* gsrc = BN ? a.bn.dz : a.dy;
    const rsrc_i4 rx = make_rsrc(a.x, a.xbytes), rdx = make_rsrc(a.dx, a.xbytes), rg = make_rsrc(gsrc, a.ybytes),
                  ryy = make_rsrc(a.y, a.ybytes), rdy = make_rsrc(a.dy, a.ybytes);
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
        q.g = dwm_ld<GV>(rg, off, ok);
        q.y = dwm_ld<GV>(ryy, off, ok && need_y);
        return q;
    };
    auto make_g = [&](const Raw& q, int r) -> Row<GV> {
        Vals<GV> g;
#pragma unroll
        for (int i = 0; i < GV; ++i) g.v[i] = 0.f;
        const bool ok = g_row_ok(r);
        if (ok) {
#pragma unroll
            for (int i = 0; i < GV; ++i) g.v[i] = gval(q.g.v[i], q.y.v[i]);
        }
        if (!BN) dwm_st<GV>(rdy, gbase + (unsigned)r * grow, g, ok && wb && r >= r0 && r < r1);
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

    // step k of the march is gradient row rf + dir * k, k < nrows
    const int nrows = r1 - r0, dir = m.up ? -1 : 1, rf = m.up ? r1 - 1 : r0;
    if constexpr (S == 1) {
        // the rows of x / dx the band owns, one per step; A / B / Cr of `body`: the gradient rows above, at and below it
        unsigned xo = xbase + (unsigned)rf * xrow;
        const unsigned xstep = m.up ? 0u - xrow : xrow;
        // `valid`: the lane has a row at this step (memory instructions stay outside divergent control flow, see dwm_ld)
        auto body = [&](const Row<V>& A, const Row<V>& B, const Row<V>& Cr, const Vals<V>& xraw, bool valid) {
            // data gradient, taps in the reference's scatter order: descending kh, descending kw
            Vals<V> d = OVW ? dwm_zero<V>() : dwm_ld<V>(rdx, xo, valid);
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
            dwm_st<V>(rdx, xo, d, valid);
            if (!valid) return;
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
            if (sums) in_sums(d, xv, xraw);
        };
        auto step_row = [&](int st) -> int { return st <= nrows ? rf + dir * st : -1; };
        Row<V> P = make_g(fetch_g(rf - dir), rf - dir), Q = make_g(fetch_g(step_row(0)), step_row(0));
        Raw gring[PF];      // gradient rows of steps k + 1 .. k + PF (the row ahead of the window), requested PF steps early
        Vals<V> xring[PF];  // the band's own x rows of steps k .. k + PF - 1
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            gring[u] = fetch_g(step_row(1 + u));
            xring[u] = dwm_ld<V>(rx, xo + (unsigned)u * xstep, m.on && u < nrows);
        }
        for (int k0 = 0; k0 < a.g.len; k0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int k = k0 + u;
                // a slot's next request goes out right AFTER its last use: the rows just used are dead, the new ones land in
                // the same registers and nothing has to be copied (and waited for) at the loop's back edge
                const Row<V> Nx = make_g(gring[u], step_row(k + 1));
                gring[u] = fetch_g(step_row(k + 1 + PF));
                if (m.up) body(Nx, Q, P, xring[u], k < nrows);
                else body(P, Q, Nx, xring[u], k < nrows);
                xring[u] = dwm_ld<V>(rx, xo + (unsigned)PF * xstep, m.on && k + PF < nrows);
                xo += xstep;
                P = Q;
                Q = Nx;
            }
        }
    } else {
        // a gradient row: the lane's GV values and (index GV + 1 of the row) the right neighbour's first. Step k works on
        // gradient row r and the input rows 2r, 2r + 1 with the gradient rows r (Bg) and r + 1 (Cg): marching down the new
        // row is r + 1 and r is carried, marching up r is new and r + 1 carried
        unsigned xo = xbase + (unsigned)(2 * rf) * xrow;
        const unsigned xstep = m.up ? 0u - 2u * xrow : 2u * xrow;
        // `valid` / `odd_ok`: the lane has rows 2r / 2r + 1 at this step (memory instructions stay outside divergent control flow)
        auto body = [&](const Row<GV>& Bg, const Row<GV>& Cg, const Vals<V>& xr0, const Vals<V>& xr1, bool valid, bool odd_ok) {
            Vals<V> d0 = OVW ? dwm_zero<V>() : dwm_ld<V>(rdx, xo, valid), d1 = OVW ? dwm_zero<V>() : dwm_ld<V>(rdx, xo + xrow, odd_ok);
#pragma unroll
            for (int j = 0; j < GV; ++j) {
                // the 2 x 2 input block under gradient column j: g00 = g[r][j], g01 = g[r][j + 1], g10 / g11 one row down
                const float g00 = Bg.v[1 + j], g01 = Bg.v[2 + j], g10 = Cg.v[1 + j], g11 = Cg.v[2 + j];
                // data gradient: the four parity classes meet 1, 2, 2 and 4 taps, in the reference's scatter order
                d0.v[2 * j] = __fadd_rn(d0.v[2 * j], __fmul_rn(w[4], g00));
                d0.v[2 * j + 1] = __fadd_rn(__fadd_rn(d0.v[2 * j + 1], __fmul_rn(w[5], g00)), __fmul_rn(w[3], g01));
                d1.v[2 * j] = __fadd_rn(__fadd_rn(d1.v[2 * j], __fmul_rn(w[7], g00)), __fmul_rn(w[1], g10));
                d1.v[2 * j + 1] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(d1.v[2 * j + 1], __fmul_rn(w[8], g00)),
                                                                __fmul_rn(w[6], g01)), __fmul_rn(w[2], g10)), __fmul_rn(w[0], g11));
            }
            dwm_st<V>(rdx, xo, d0, valid);
            dwm_st<V>(rdx, xo + xrow, d1, odd_ok);
            if (!valid) return;
            Vals<V> x0 = xr0, x1 = xr1;
            if (BNIN) x0 = dwm_bnin_v<V>(xr0, kin, in_act);
            if (BNIN && odd_ok) x1 = dwm_bnin_v<V>(xr1, kin, in_act);
#pragma unroll
            for (int j = 0; j < GV; ++j) {
                const float g00 = Bg.v[1 + j], g01 = Bg.v[2 + j], g10 = Cg.v[1 + j], g11 = Cg.v[2 + j];
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
            }
            if (sums) {
                in_sums(d0, x0, xr0);
                if (odd_ok) in_sums(d1, x1, xr1);
            }
        };
        const int down1 = m.up ? 0 : 1;
        auto new_row = [&](int st) -> int { return st < nrows ? rf + dir * st + down1 : -1; };  // the gradient row step st adds
        Row<GV> X = make_g(fetch_g(m.on ? rf + 1 - down1 : -1), m.on ? rf + 1 - down1 : -1);   // carried: r (down) / r + 1 (up)
        Raw gring[PF];                 // the gradient rows steps k .. k + PF - 1 add
        Vals<V> xring0[PF], xring1[PF];  // input rows 2r, 2r + 1 of those steps
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            gring[u] = fetch_g(new_row(u));
            const bool on = m.on && u < nrows;
            xring0[u] = dwm_ld<V>(rx, xo + (unsigned)u * xstep, on);
            xring1[u] = dwm_ld<V>(rx, xo + (unsigned)u * xstep + xrow, on && 2 * (rf + dir * u) + 1 < a.H);
        }
        for (int k0 = 0; k0 < a.g.len; k0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int k = k0 + u, r = rf + dir * k;
                const bool valid = k < nrows, odd_ok = valid && 2 * r + 1 < a.H;
                // a slot's next request goes out right after its last use (into the same registers)
                const Row<GV> Nw = make_g(gring[u], new_row(k));
                gring[u] = fetch_g(new_row(k + PF));
                if (m.up) body(Nw, X, xring0[u], xring1[u], valid, odd_ok);
                else body(X, Nw, xring0[u], xring1[u], valid, odd_ok);
                const bool more = m.on && k + PF < nrows;
                xring0[u] = dwm_ld<V>(rx, xo + (unsigned)PF * xstep, more);
                xring1[u] = dwm_ld<V>(rx, xo + (unsigned)PF * xstep + xrow, more && 2 * (r + dir * PF) + 1 < a.H);
                xo += xstep;
                X = Nw;
            }
        }
    }
    const int splits = (int)(a.g.bands / a.C);  // N * BPP
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    dwm_band_sums<kDwmPart>(acc, red[threadIdx.x >> 6], a.g.L, a.g.G, [&](int q, int i, float t) {
        int p, bi;
        bool up;
        if (!dwm_map(wave, q, a.g.G, a.g.BPP, planes, p, bi, up)) return;
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
    return (size_t)dwm_splits(s, g);
}

bool depthwise_forward_march(const float* x, const float* w, const float* bias, float* y, const DwShape& s, int act,
                             ConvStats* stats, const DwBnIn* in) {
    if (!depthwise_march_ok(s) || !act_is_cheap(act) || act == BCNN_HIP_ACT_PRELU) return false;
    if (in && (!in->mean || !act_is_cheap(in->act) || in->act == BCNN_HIP_ACT_PRELU)) return false;
    DwmFwdArgs a;
    a.g = dwm_plan(s);
    if (!dwm_aligned(a.g.V, {x}) || !dwm_aligned(s.stride == 1 ? a.g.V : a.g.V / 2, {y})) return false;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = nullptr;
    a.N = s.N; a.C = s.C; a.H = s.H; a.W = s.W; a.OH = s.OH; a.OW = s.OW; a.act = act;
    a.xbytes = (unsigned)((size_t)s.N * s.C * s.H * s.W * 4); a.ybytes = (unsigned)((size_t)s.N * s.C * s.OH * s.OW * 4);
    const int splits = dwm_splits(s, a.g);
    if (stats) {
        stats->splits = 0;
        if (stats->partials && stats->capacity >= (size_t)s.C * splits * 2) {
            a.stats = stats->partials;
            stats->splits = splits;
        }
    }
    a.in = in ? *in : DwBnIn{nullptr, nullptr, nullptr, nullptr, 0};
    const unsigned waves = dwm_waves(a.g, (long long)s.N * s.C), blocks = (waves + 3) / 4;
    hipStream_t st = current_stream();
    trace_kernel(in ? "dwm_fwd_kernel:bnin" : "dwm_fwd_kernel");
    // rows requested ahead: 2 for 16-byte lanes at stride 1 (with the producer's batch-norm applied on load the kernel is vector-ALU
    // bound and 4 costs it two waves per SIMD), 4 at stride 2 and for the 8- / 4-byte lanes of small planes (latency bound)
    int pf = (s.stride == 2 || a.g.V < 4) ? 4 : 2;
#ifdef BCNN_HIP_EXPERIMENT
    if (const char* e = getenv("BCNN_HIP_DWM_PF")) pf = atoi(e);
    if (s.stride == 2) { if (const char* e = getenv("BCNN_HIP_DWM_PF_S2")) pf = atoi(e); }
    if (a.g.V < 4) { if (const char* e = getenv("BCNN_HIP_DWM_PF_SMALL")) pf = atoi(e); }
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
        else if (pf == 2) dwm_fwd_kernel<SV, VV, BV, 2, true><<<blocks, 256, 0, st>>>(a);           \
        else dwm_fwd_kernel<SV, VV, BV, 4, true><<<blocks, 256, 0, st>>>(a);                        \
    } while (0)
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

// every condition under which depthwise_backward_march launches nothing (a caller that has to prepare dy asks first)
bool depthwise_backward_march_takes(const float* x, const float* y, const float* dy, const float* dx, const DwShape& s, int act,
                                    const DwBnBwd* bn, const DwBnIn* in) {
    if (!depthwise_march_ok(s) || !act_bwd_is_cheap(act) || act == BCNN_HIP_ACT_PRELU || !dx) return false;
    if (in && (!in->mean || !act_is_cheap(in->act) || in->act == BCNN_HIP_ACT_PRELU)) return false;
    const DwmGeom g = dwm_plan(s);
    return dwm_aligned(g.V, {x, dx}) && dwm_aligned(s.stride == 1 ? g.V : g.V / 2, {y, dy, bn ? bn->dz : nullptr});
}

bool depthwise_backward_march(const float* x, const float* w, const float* y, float* dy, float* dx, float* dw, float* dbias,
                              const DwShape& s, int act, int overwrite, int write_back, const DwBnBwd* bn, const DwBnIn* in,
                              ConvStats* in_sums) {
    if (in_sums) in_sums->splits = 0;
    if (!depthwise_backward_march_takes(x, y, dy, dx, s, act, bn, in)) return false;
    DwmBwdArgs a;
    a.g = dwm_plan(s);
    a.x = x; a.w = w; a.y = y; a.dy = dy; a.dx = dx;
    a.N = s.N; a.C = s.C; a.H = s.H; a.W = s.W; a.OH = s.OH; a.OW = s.OW; a.act = act;
    a.overwrite = overwrite; a.write_back = write_back;
    a.xbytes = (unsigned)((size_t)s.N * s.C * s.H * s.W * 4); a.ybytes = (unsigned)((size_t)s.N * s.C * s.OH * s.OW * 4);
    a.fM = (float)((long long)s.N * s.OH * s.OW);
    a.rfM = 1.0f / a.fM;  // host division: IEEE, round to nearest
    a.bn = bn ? *bn : DwBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    a.in = in ? *in : DwBnIn{nullptr, nullptr, nullptr, nullptr, 0};
    const int splits = dwm_splits(s, a.g);
    a.partials = reduce_scratch((size_t)s.C * splits * kDwmPart);
    a.in_sums = nullptr;
    // the sums of the producer's batch-norm backward are of the COMPLETE gradient: only when this kernel is its sole writer,
    // and for producer activations whose derivative is 0 or 1
    if (in && in_sums && in_sums->partials && overwrite && (in->act == BCNN_HIP_ACT_NONE || in->act == BCNN_HIP_ACT_RELU) &&
        in_sums->capacity >= (size_t)s.C * splits * 2) {
        a.in_sums = in_sums->partials;
        in_sums->splits = splits;
    }
    const unsigned waves = dwm_waves(a.g, (long long)s.N * s.C), blocks = (waves + 3) / 4;
    hipStream_t st = current_stream();
    trace_kernel(bn && in ? "dwm_bwd_kernel:bn+bnin" : bn ? "dwm_bwd_kernel:bn" : in ? "dwm_bwd_kernel:bnin" : "dwm_bwd_kernel");
    bool relu = act == BCNN_HIP_ACT_RELU && (!in || in->act == BCNN_HIP_ACT_RELU);
    if (BCNN_EXP_ENV("BCNN_HIP_DWM_NORELU")) relu = false;  // A/B switch (experiment build only)
#define DWM_LAUNCH_O(SV, VV, RV, OV)                                                                                \
    do {                                                                                                            \
        constexpr int PFV = (VV) == 4 ? DWM_BWD_PF_WIDE : DWM_BWD_PF_SMALL;                                         \
        if (bn && in) dwm_bwd_kernel<SV, VV, true, true, RV, PFV, OV><<<blocks, 256, 0, st>>>(a);                   \
        else if (bn) dwm_bwd_kernel<SV, VV, true, false, RV, PFV, OV><<<blocks, 256, 0, st>>>(a);                   \
        else if (in) dwm_bwd_kernel<SV, VV, false, true, RV, PFV, OV><<<blocks, 256, 0, st>>>(a);                   \
        else dwm_bwd_kernel<SV, VV, false, false, RV, PFV, OV><<<blocks, 256, 0, st>>>(a);                          \
    } while (0)
#define DWM_LAUNCH_R(SV, VV, RV)                       \
    do {                                               \
        if (overwrite) DWM_LAUNCH_O(SV, VV, RV, true); \
        else DWM_LAUNCH_O(SV, VV, RV, false);          \
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
#undef DWM_LAUNCH_O
    KERNEL_CHECK();
    dwl_finalize_launch(a.partials, splits, s.C, dw, dbias, st);
    return true;
}

}  // namespace bcnn_hip
