// conv_direct.hip -- LDS-free implicit-GEMM kernels for convolutions with a SMALL reduction length
// (K = C/g*k*k <= 32, e.g. the 3x3 stem of BASELINE configs[1]: 3 -> 64 channels at 224x224).
//
// For such layers the GEMM is HBM-bound (arithmetic intensity 12.9 FLOP/B at configs[1]), so the
// kernels are built to touch every activation byte exactly once and to spend as few issue slots as
// possible next to the MFMAs:
//   forward : the weight matrix lives in MFMA A-operand REGISTERS for the lifetime of a (persistent)
//             wave; each B operand (one im2col element per lane: lane&31 = output pixel, lane>>5 picks
//             the k of the pair) is loaded straight from global memory in fragment order -- a wave
//             load instruction reads two 128-byte runs of consecutive pixels; the input (77 MB) is
//             served by L1/L2/MALL. No LDS staging, no barriers, no materialised im2col.
//   dW      : the reduction runs over output pixels q. The MFMA reduction index is free to be
//             permuted, so step s pairs q0+s (lanes 0-31) with q0+8+s (lanes 32-63): every lane's
//             eight A values (dy, lane&31 = output channel) and eight B values (x taps, lane&31 = k)
//             are then 32 CONTIGUOUS bytes => two 16-byte loads per operand per 16 output pixels.
//             A 29th all-ones im2col column yields the bias gradient in the same pass.
// Reference semantics as in conv_fwd.hip / conv_bwd.hip (bcnn_conv_layer.c:367-587).
#include "conv_common.h"

// Cache policy of the result stores: 2 = nt (non-temporal). The output of a small-K layer is a stream far larger
// than L2 + Infinity Cache; written "nt" it does not displace the input taps there (configs[1]: forward -3 %, and the
// dW pass that follows no longer competes with the write-back of dirty output lines: 0.50 -> 0.46 ms).
#ifndef STORE_AUX
#define STORE_AUX 2
#endif

namespace bcnn_hip {

struct f4u { float x, y, z, w; } __attribute__((packed, aligned(4)));  // 16-byte load, 4-byte aligned

// ================================================================================================
// forward
// ================================================================================================
struct ConvDirectFwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* slopes;
    float* y;
    ConvShape s;
    int act, add_bias;
    int ntiles;        // ceil(total_q / 32)
    int tiles_per_block;
};

// TM: 32-row tiles of output channels; KS >= ceil(K/2): unrolled MFMA steps (the runtime count skips
// the unused tail); KSZ: kernel size (compile time). Loads and stores go through buffer descriptors:
// an out-of-range offset reads 0 / drops the store, which implements zero padding and ragged edges
// without exec-mask branches or post-load selects.
// ACTM: 0 = no activation, 1 = ReLU, 2 = any other cheap activation (runtime switch)
template <int TM, int KS, int KSZ, int ACTM>
__global__ __launch_bounds__(256, 3) void conv_fwd_direct_kernel(const ConvDirectFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float sbias[TM * 32];
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.y;
    const float* wg = a.w + (long long)g * s.Mg * s.K;
    constexpr int KK2 = KSZ * KSZ;
    constexpr unsigned OOB = 0x80000000u;  // >= num_records of both descriptors (tensors < 2 GiB)

    if (tid < TM * 32) {
        float b = 0.f;
        if (a.add_bias && tid < s.Mg) {
            b = a.bias[g * s.Mg + tid];
            if (b == 1.0f) b = 0.f;  // bcnn_add_scalar (AVX build) adds nothing for exactly 1.0f
        }
        sbias[tid] = b;
    }
    // A operand W[f = tm*32 + l31][k = 2*st + hi] resident in registers for the whole kernel; per step
    // the lane also keeps its tap's byte offset (OOB when k >= K) and, packed 4 per register, the tap's
    // bit index kr*KSZ + kc in the per-pixel validity mask (used on border tiles only).
    float areg[TM][KS];
    unsigned koffb[KS];
    unsigned tapbits[(KS + 3) / 4];
#pragma unroll
    for (int i = 0; i < (KS + 3) / 4; ++i) tapbits[i] = 0;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        const int k = 2 * st + hi;
        const bool kv = k < s.K;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int f = tm * 32 + l31;
            const bool ok = kv && (f < s.Mg);
            const float v = wg[ok ? (long long)f * s.K + k : 0];
            areg[tm][st] = ok ? v : 0.f;
        }
        const int c = k / KK2, r = k - c * KK2;
        const int kr = r / KSZ, kc = r - kr * KSZ;
        koffb[st] = kv ? (unsigned)(c * s.HW + kr * s.W + kc) * 4u : OOB;
        tapbits[st >> 2] |= (unsigned)(kv ? r : 31) << (8 * (st & 3));
    }
    __syncthreads();

    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((long long)s.N * s.C * s.HW * 4), 0x00020000);
    const auto ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((long long)s.N * s.F * s.OHOW * 4), 0x00020000);
    const unsigned xg_off = (unsigned)(g * s.Cg) * (unsigned)s.HW;
    const int t_begin = blockIdx.x * a.tiles_per_block;
    int t_end = t_begin + a.tiles_per_block;
    if (t_end > a.ntiles) t_end = a.ntiles;
    const bool full_m = (s.Mg == TM * 32);
    const unsigned fstride = (unsigned)s.OHOW * 4u;

    // per-tile lane state: pixel coordinates, byte offset of its top-left tap, output byte offset
    unsigned pn = 0, poh = 0, pow_ = 0;
    unsigned pixb = 0, ybyte = OOB;
    bool interior = false, qvalid = false;
    auto place = [&]() {  // derive offsets from (pn, poh, pow_)
        const int ih0 = (int)poh * s.stride - s.pad, iw0 = (int)pow_ * s.stride - s.pad;
        interior = qvalid && ih0 >= 0 && iw0 >= 0 && (ih0 + KSZ <= s.H) && (iw0 + KSZ <= s.W);
        pixb = ((pn * (unsigned)s.C) * (unsigned)s.HW + xg_off + (unsigned)(ih0 * s.W + iw0)) * 4u;
        ybyte = qvalid ? ((pn * (unsigned)s.F + (unsigned)(g * s.Mg)) * (unsigned)s.OHOW + poh * (unsigned)s.OW + pow_) * 4u +
                             4u * (unsigned)hi * fstride
                       : OOB;
    };
    auto locate = [&](int t) {  // full decode (two divisions)
        const unsigned q = (unsigned)t * 32u + (unsigned)l31;
        qvalid = q < (unsigned)s.total_q;
        const unsigned qq = qvalid ? q : 0u;
        pn = qq / (unsigned)s.OHOW;
        const unsigned pix = qq - pn * (unsigned)s.OHOW;
        poh = pix / (unsigned)s.OW;
        pow_ = pix - poh * (unsigned)s.OW;
        place();
    };
    const bool wide = s.OW >= 128;  // then +128 pixels wraps at most one row: no division per tile
    auto advance = [&](int t) {      // move the lane from tile t-4 to tile t
        if (wide) {
            const unsigned q = (unsigned)t * 32u + (unsigned)l31;
            qvalid = q < (unsigned)s.total_q;
            pow_ += 128u;
            if (pow_ >= (unsigned)s.OW) { pow_ -= (unsigned)s.OW; poh += 1u; }
            if (poh >= (unsigned)s.OH) { poh -= (unsigned)s.OH; pn += 1u; }
            place();
        } else {
            locate(t);
        }
    };
    auto tap_mask = [&]() -> unsigned {  // bit kr*KSZ + kc set <=> tap inside the image (border tiles)
        const int ih0 = (int)poh * s.stride - s.pad, iw0 = (int)pow_ * s.stride - s.pad;
        unsigned colm = 0, m = 0;
#pragma unroll
        for (int kc = 0; kc < KSZ; ++kc) colm |= ((unsigned)(iw0 + kc) < (unsigned)s.W ? 1u : 0u) << kc;
#pragma unroll
        for (int kr = 0; kr < KSZ; ++kr) m |= ((unsigned)(ih0 + kr) < (unsigned)s.H ? colm : 0u) << (kr * KSZ);
        return qvalid ? m : 0u;
    };
    auto ldx = [&](unsigned off) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    };

    // ---- tile pipeline ---------------------------------------------------------------------------
    // gfx9-family waves have ONE counter (vmcnt) for loads and stores, and loads may overtake stores: a
    // wait for operands issued BEFORE a tile's 32 stores has to wait for those stores too (the compiler
    // emits vmcnt(0)), which parks the wave for a full HBM write round trip per tile. The order below
    // makes every wait cover only work issued a whole tile earlier:
    //     wait(all) ; store(t-1) ; load(t+1) ; MFMA(t)
    // so the stores of tile t-1 and the operand loads of tile t+1 both fly under the MFMAs of tile t.
    // Two operand buffers alternate (loop unrolled by two to keep them in fixed registers).
    f32x16 acc[TM];
    auto load_tile = [&](float (&buf)[KS]) {  // operands of the tile the lane state points at
        if (__all(interior)) {  // every tap of every lane is inside the image: offset = pixel + tap, nothing else
                                // (a tap with k >= K carries OOB = 2^31: pixel + 2^31 stays out of range for tensors < 1 GiB)
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                buf[st] = ldx(pixb + koffb[st]);
            }
        } else {                // border tile (or tail): invalid taps get an out-of-range offset => 0
            const unsigned m = tap_mask();
            unsigned tb[(KS + 3) / 4];
#pragma unroll
            for (int i = 0; i < (KS + 3) / 4; ++i) {
                tb[i] = tapbits[i];
                asm volatile("" : "+v"(tb[i]));  // keep the 4-per-register packing (no hoisted unpacked copies)
            }
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                const unsigned tap = (tb[st >> 2] >> (8 * (st & 3))) & 0xffu;
                buf[st] = ldx(((m >> tap) & 1u) ? pixb + koffb[st] : OOB);
            }
        }
    };
    auto compute_tile = [&](const float (&buf)[KS]) {
        // the accumulators start from the bias (4 consecutive channels per 16-byte LDS read), so the
        // bias add costs no VALU: row(r) = (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 b4 = *reinterpret_cast<const float4*>(&sbias[tm * 32 + 8 * rq + 4 * hi]);
                acc[tm][rq * 4 + 0] = b4.x;
                acc[tm][rq * 4 + 1] = b4.y;
                acc[tm][rq * 4 + 2] = b4.z;
                acc[tm][rq * 4 + 3] = b4.w;
            }
#pragma unroll
        for (int st = 0; st < KS; ++st)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                acc[tm] = mfma32(areg[tm][st], buf[st], acc[tm]);
            }
    };
    // epilogue on the accumulators: activation, one store per element.
    // y[n][g*Mg + f][pix], f = tm*32 + (r&3) + 8*(r>>2) + 4*hi: registers r..r+3 are 4 consecutive channels
    auto store_tile = [&](unsigned ycur) {
        unsigned fs = fstride;
        asm volatile("" : "+s"(fs));  // recompute the 32 scalar channel offsets per tile instead of pinning 32 SGPRs
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[tm][r];
            if (ACTM == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] * (float)(v[r] > 0);
            } else if (ACTM == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = tm * 32 + mfma_row(r, lane);
                    const float sl = (a.act == BCNN_HIP_ACT_PRELU && f < s.Mg) ? a.slopes[g * s.Mg + f] : 0.f;
                    v[r] = act_fwd_cheap(v[r], a.act, sl);
                }
            }
            // one VGPR offset per tile (ycur); the channel stride rides in the scalar offset operand
            if (full_m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fr = tm * 32 + (r & 3) + 8 * (r >> 2);  // + 4*hi is folded into ycur
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ry, ycur,
                                                          (unsigned)fr * fs, STORE_AUX);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fr = tm * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned off = (fr + 4 * hi < s.Mg) ? ycur : OOB;  // rows beyond F/groups are dropped
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ry, off,
                                                          (unsigned)fr * fs, STORE_AUX);
                }
            }
        }
    };

    int t = t_begin + wid;
    if (t >= t_end) return;
    float bufA[KS], bufB[KS];
    locate(t);
    load_tile(bufA);
    unsigned yprev = OOB;
    bool have_prev = false;
    // ablation switches for tools/exp (normal builds define none of them)
#define DIRECT_WAIT() __builtin_amdgcn_s_waitcnt(0x0f70)
#define DIRECT_STORE(y) store_tile(y)
    // one pipeline stage: `cur` holds tile t's operands (issued one stage ago), `nxt` receives tile t+4's
#define DIRECT_STAGE(cur, nxt)                                                                          \
    {                                                                                                   \
        const unsigned ycur = ybyte;                                                                    \
        const bool more = (t + 4 < t_end); /* wave-uniform */                                           \
        DIRECT_WAIT();   /* vmcnt(0): operands of tile t, stores of tile t-8 */                          \
        if (have_prev) DIRECT_STORE(yprev);                                                             \
        if (more) {                                                                                     \
            advance(t + 4);                                                                             \
            load_tile(nxt);                                                                             \
        }                                                                                               \
        compute_tile(cur);                                                                              \
        yprev = ycur;                                                                                   \
        have_prev = true;                                                                               \
        if (!more) break;                                                                               \
        t += 4;                                                                                         \
    }
    for (;;) {
        DIRECT_STAGE(bufA, bufB)
        DIRECT_STAGE(bufB, bufA)
    }
#undef DIRECT_STAGE
    DIRECT_STORE(yprev);
#undef DIRECT_STORE
#undef DIRECT_WAIT
}

// Pulls a small tensor into the memory-side Infinity Cache (256 MB) ahead of a kernel that streams a much larger
// result. Measured on configs[1]: with the 77 MB input resident there and the 1.6 GB output written with
// non-temporal stores, HBM sees a pure write stream instead of writes finely interleaved with reads (bus
// turnarounds) -- forward 0.45 -> 0.39 ms including this pass (15 us).
__global__ __launch_bounds__(256) void cache_prefetch_kernel(const float4* __restrict__ p, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) *sink = acc;  // never true for finite data; keeps the loads alive
}

// input small enough for the Infinity Cache and dwarfed by the output: pull it in first, so that the kernel's HBM traffic
// is a pure write stream (shared with conv_window.hip)
void conv_prefetch_input(const float* x, const ConvShape& s, float* sink) {
    const size_t xb = (size_t)s.N * s.C * s.HW * 4, yb = (size_t)s.N * s.F * s.OHOW * 4;
    static const bool pf_on = [] { const char* e = BCNN_EXP_ENV("BCNN_HIP_NO_PREFETCH"); return !(e && e[0] == '1'); }();
    if (pf_on && xb <= (128u << 20) && yb >= 4 * xb && yb >= (256u << 20) && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        cache_prefetch_kernel<<<kCUs * 8, 256, 0, current_stream()>>>(reinterpret_cast<const float4*>(x), xb / 16, sink);
}

bool conv_forward_direct(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                         const ConvShape& s, int act, int raw) {
    if (s.pointwise || s.K > 32 || s.Mg > 64 || s.total_q == 0) return false;
    if ((long long)s.N * s.F * s.OHOW >= (1LL << 29) || (long long)s.N * s.C * s.HW >= (1LL << 28)) return false;
    if (!((s.ksz == 3 && s.Cg <= 3) || (s.ksz == 5 && s.Cg == 1))) return false;
    ConvDirectFwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.slopes = slopes; a.y = y; a.s = s;
    a.act = raw ? BCNN_HIP_ACT_NONE : act;
    a.add_bias = raw ? 0 : 1;
    a.ntiles = ceil_div(s.total_q, 32);
    // persistent-ish: ~8 workgroups per CU, each walking a contiguous run of tiles (its 4 waves share rows)
    int blocks = kCUs * 8;
    a.tiles_per_block = ceil_div(a.ntiles, blocks);
    if (a.tiles_per_block < 4) a.tiles_per_block = 4;
    blocks = ceil_div(a.ntiles, a.tiles_per_block);
    dim3 grid((unsigned)blocks, (unsigned)s.groups);
    const int ks = (s.K + 1) / 2;
    const int tm = (s.Mg <= 32) ? 1 : 2;
    KTimer kt(K_CONV_FWD, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    conv_prefetch_input(x, s, y);
    const int actm = (a.act == BCNN_HIP_ACT_NONE) ? 0 : (a.act == BCNN_HIP_ACT_RELU ? 1 : 2);
#define LAUNCH(TMv, KSv, KZ)                                                                            \
    do {                                                                                                \
        if (actm == 0) conv_fwd_direct_kernel<TMv, KSv, KZ, 0><<<grid, 256, 0, current_stream()>>>(a);      \
        else if (actm == 1) conv_fwd_direct_kernel<TMv, KSv, KZ, 1><<<grid, 256, 0, current_stream()>>>(a); \
        else conv_fwd_direct_kernel<TMv, KSv, KZ, 2><<<grid, 256, 0, current_stream()>>>(a);                \
    } while (0)
    if (s.ksz == 3 && ks == 14) { if (tm == 1) LAUNCH(1, 14, 3); else LAUNCH(2, 14, 3); }
    else if (s.ksz == 3 && ks == 9) { if (tm == 1) LAUNCH(1, 9, 3); else LAUNCH(2, 9, 3); }
    else if (s.ksz == 3 && ks == 5) { if (tm == 1) LAUNCH(1, 5, 3); else LAUNCH(2, 5, 3); }
    else if (s.ksz == 5 && ks == 13) { if (tm == 1) LAUNCH(1, 13, 5); else LAUNCH(2, 13, 5); }
    else return false;
#undef LAUNCH
    KERNEL_CHECK();
    return true;
}

// ================================================================================================
// dW (+ bias gradient)
// ================================================================================================
struct ConvDirectDwArgs {
    const float* x;
    const float* dy;
    float* partials;  // [nblocks][groups][TM*32][32]
    ConvShape s;
    int nwin;         // N*OH*(OW/16) windows of 16 consecutive output pixels
    int win_per_block;
    int bias_col;
};

template <int TM>
struct DwFrag {
    float4 a[TM][2];  // dy[f][q0 + 8*hi .. +7]
    float4 b[2];      // im2col column k at the same 8 output pixels
};

// Loads one 16-pixel window's MFMA fragments. Every load is unconditional from an always-legal
// address; padding / out-of-range lanes are zeroed with selects (no exec-mask branches, no scratch).
template <int TM>
__device__ __forceinline__ DwFrag<TM> dw_load_window(const ConvDirectDwArgs& a, int wdx, int wpr, int g, int l31,
                                                     int hi, int koff, int kr, int kc, bool ones, bool kvalid) {
    const ConvShape& s = a.s;
    DwFrag<TM> fr;
    const unsigned row = (unsigned)wdx / (unsigned)wpr;  // n*OH + oh
    const int ow_base = (int)((unsigned)wdx - row * (unsigned)wpr) * 16;
    const int ow0 = ow_base + 8 * hi;
    const unsigned n = row / (unsigned)s.OH, oh = row - n * (unsigned)s.OH;
    const float* gp = a.dy + ((long long)(n * (unsigned)s.F + (unsigned)(g * s.Mg)) * s.OHOW + (long long)(oh * s.OW + ow0));
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int f = tm * 32 + l31;
        const bool fv = f < s.Mg;
        const float4* p = reinterpret_cast<const float4*>(gp + (long long)(fv ? f : 0) * s.OHOW);
        float4 v0 = p[0], v1 = p[1];
        if (!fv) { v0 = make_float4(0.f, 0.f, 0.f, 0.f); v1 = v0; }
        fr.a[tm][0] = v0;
        fr.a[tm][1] = v1;
    }
    const int ih = (int)oh - s.pad + kr;
    const bool rowok = kvalid && (unsigned)ih < (unsigned)s.H;
    const int iw = ow0 - s.pad + kc;
    const float* xrow = a.x + ((long long)(n * (unsigned)s.C + (unsigned)(g * s.Cg)) * s.HW +
                               (long long)(koff + ((int)oh - s.pad) * s.W + (ow0 - s.pad)));
    const bool edge = (ow_base == 0) || (ow_base + 16 >= s.OW);  // wave-uniform
    float t0, t1, t2, t3, t4, t5, t6, t7;
    if (!edge) {
        const float* p = rowok ? xrow : a.x;
        const f4u v0 = *reinterpret_cast<const f4u*>(p);
        const f4u v1 = *reinterpret_cast<const f4u*>(p + 4);
        t0 = v0.x; t1 = v0.y; t2 = v0.z; t3 = v0.w; t4 = v1.x; t5 = v1.y; t6 = v1.z; t7 = v1.w;
        if (!rowok) { t0 = t1 = t2 = t3 = t4 = t5 = t6 = t7 = 0.f; }
    } else {
#define DW_EDGE_LOAD(E, T)                                                     \
        {                                                                      \
            const bool ok = rowok && (unsigned)(iw + E) < (unsigned)s.W;       \
            const float v = *(ok ? xrow + E : a.x);                            \
            T = ok ? v : 0.f;                                                  \
        }
        DW_EDGE_LOAD(0, t0) DW_EDGE_LOAD(1, t1) DW_EDGE_LOAD(2, t2) DW_EDGE_LOAD(3, t3)
        DW_EDGE_LOAD(4, t4) DW_EDGE_LOAD(5, t5) DW_EDGE_LOAD(6, t6) DW_EDGE_LOAD(7, t7)
#undef DW_EDGE_LOAD
    }
    if (ones) { t0 = t1 = t2 = t3 = t4 = t5 = t6 = t7 = 1.0f; }
    fr.b[0] = make_float4(t0, t1, t2, t3);
    fr.b[1] = make_float4(t4, t5, t6, t7);
    return fr;
}

template <int TM>
__global__ __launch_bounds__(256) void conv_dw_direct_kernel(const ConvDirectDwArgs a) {
    __shared__ float red[3][TM * 32][33];
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.y;
    const int wpr = s.OW >> 4;  // windows per output row

    // this lane's im2col column k = l31
    int koff = 0, kr = 0, kc = 0;
    bool ones = false, kvalid = false;
    if (l31 < s.K) {
        const int kk2 = s.ksz * s.ksz;
        const int c = l31 / kk2, r = l31 - c * kk2;
        kr = r / s.ksz; kc = r - kr * s.ksz;
        koff = c * s.HW + kr * s.W + kc;
        kvalid = true;
    } else if (l31 == s.K && a.bias_col) {
        ones = true;
    }

    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

    const int w_begin = blockIdx.x * a.win_per_block;
    int w_end = w_begin + a.win_per_block;
    if (w_end > a.nwin) w_end = a.nwin;

    int wdx = w_begin + wid;
    DwFrag<TM> nxt;
    if (wdx < w_end) nxt = dw_load_window<TM>(a, wdx, wpr, g, l31, hi, koff, kr, kc, ones, kvalid);
    for (; wdx < w_end; wdx += 4) {
        const DwFrag<TM> cur = nxt;
        if (wdx + 4 < w_end)  // next window's loads fly under the 8*TM MFMAs below
            nxt = dw_load_window<TM>(a, wdx + 4, wpr, g, l31, hi, koff, kr, kc, ones, kvalid);
#define DW_STEP(AV, BV)                                                        \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) acc[tm] = mfma32(cur.a[tm]AV, cur.b BV, acc[tm]);
        DW_STEP([0].x, [0].x) DW_STEP([0].y, [0].y) DW_STEP([0].z, [0].z) DW_STEP([0].w, [0].w)
        DW_STEP([1].x, [1].x) DW_STEP([1].y, [1].y) DW_STEP([1].z, [1].z) DW_STEP([1].w, [1].w)
#undef DW_STEP
    }

    // cross-wave reduction (waves 1..3 -> LDS -> wave 0), then one partial tile per workgroup
    if (wid > 0) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wid - 1][tm * 32 + mfma_row(r, lane)][l31] = acc[tm][r];
    }
    __syncthreads();
    if (wid == 0) {
        float* out = a.partials + ((size_t)blockIdx.x * s.groups + g) * (TM * 32) * 32;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = tm * 32 + mfma_row(r, lane);
                out[f * 32 + l31] = ((acc[tm][r] + red[0][f][l31]) + red[1][f][l31]) + red[2][f][l31];
            }
    }
}

// dW[g][f][k] += sum_p partials[p][g][f][k] (k < K), dbias[g*Mg + f] += column K. One workgroup per 16
// output elements, 16 partial-lanes each, fixed summation order.
__global__ __launch_bounds__(256) void conv_dw_direct_finalize_kernel(const float* __restrict__ partials,
                                                                      int nparts, int groups, int Mg, int K,
                                                                      int MP, int bias_col,
                                                                      float* __restrict__ dw,
                                                                      float* __restrict__ dbias) {
    __shared__ float red[16][17];
    const int kcols = K + (bias_col ? 1 : 0);
    const int total = groups * Mg * kcols;
    const int e = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
    float sum = 0.f;
    int g = 0, f = 0, k = 0;
    if (e < total) {
        k = e % kcols;
        const int t = e / kcols;
        f = t % Mg; g = t / Mg;
        for (int p = pl; p < nparts; p += 16) sum += partials[(((size_t)p * groups + g) * MP + f) * 32 + k];
    }
    red[pl][threadIdx.x & 15] = sum;
    __syncthreads();
    if (pl == 0 && e < total) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i][threadIdx.x & 15];
        if (k < K) dw[((size_t)g * Mg + f) * K + k] += tot;
        else dbias[g * Mg + f] += tot;
    }
}

// shared with conv_window.hip (same partial layout)
void conv_dw_direct_finalize(const float* partials, int nparts, int groups, int Mg, int K, int MP, int bias_col, float* dw,
                             float* dbias) {
    const int total = groups * Mg * (K + bias_col);
    conv_dw_direct_finalize_kernel<<<ceil_div(total, 16), 256, 0, current_stream()>>>(partials, nparts, groups, Mg, K, MP,
                                                                                       bias_col, dw, dbias);
    KERNEL_CHECK();
}

static bool dw_direct_ok(const ConvShape& s) {
    return !s.pointwise && s.K < 32 && s.Mg <= 64 && s.stride == 1 && (s.OW % 16) == 0 && s.total_q > 0 &&
           (long long)s.N * s.F * s.OHOW < (1LL << 32) && (long long)s.N * s.C * s.HW < (1LL << 32) &&
           s.H < 0x4000;
}

static void dw_direct_plan(const ConvShape& s, int* nwin, int* wpb, int* blocks) {
    *nwin = s.N * s.OH * (s.OW / 16);
    int b = kCUs * 4;
    int per = ceil_div(*nwin, b);
    if (per < 16) per = 16;
    *wpb = per;
    *blocks = ceil_div(*nwin, per);
}

size_t conv_dw_direct_workspace_floats(const ConvShape& s) {
    if (!dw_direct_ok(s)) return 0;
    int nwin, wpb, blocks;
    dw_direct_plan(s, &nwin, &wpb, &blocks);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    return (size_t)blocks * s.groups * tm * 32 * 32;
}

// returns false when the shape is not covered (caller falls back to the LDS-tiled kernel);
// when it returns true the bias gradient has been accumulated too (if dbias != NULL).
bool conv_backward_weights_direct(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s,
                                  float* workspace, size_t workspace_floats) {
    if (!dw_direct_ok(s)) return false;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) != 0) return false;
    int nwin, wpb, blocks;
    dw_direct_plan(s, &nwin, &wpb, &blocks);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    const size_t need = (size_t)blocks * s.groups * tm * 32 * 32;
    if (workspace == nullptr || workspace_floats < need) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n",
                workspace_floats, need);
        exit(1);
    }
    KTimer kt(K_CONV_DW, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    ConvDirectDwArgs a;
    a.x = x; a.dy = dy; a.partials = workspace; a.s = s; a.nwin = nwin; a.win_per_block = wpb;
    a.bias_col = dbias ? 1 : 0;
    dim3 grid((unsigned)blocks, (unsigned)s.groups);
    if (tm == 1) conv_dw_direct_kernel<1><<<grid, 256, 0, current_stream()>>>(a);
    else conv_dw_direct_kernel<2><<<grid, 256, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    conv_dw_direct_finalize(workspace, blocks, s.groups, s.Mg, s.K, tm * 32, a.bias_col, dw, dbias);
    return true;
}

}  // namespace bcnn_hip
