// conv_window.hip -- window-in-LDS kernels for 3x3 / stride-1 convolutions with a SMALL reduction length
// (K = C/g*9 <= 27: BASELINE configs[1], 3 -> 64 channels at 224x224), forward and weight gradient.
//
// Such a layer is HBM-bound (12.9 FLOP/B at configs[1]: 1.64 GB of result / gradient against 77 MB of input), so the
// activation stream should cost the multiplying waves nothing else. The LDS-free kernels of conv_direct.hip gather
// every im2col element per lane from global memory: 14 two-segment gathers per 32 output pixels share the waves' one
// memory counter with the 32 result stores, carry per-lane padding / border selects, and measured 0.40 / 0.47 ms
// (forward / dW) against 0.32 for "MFMAs + stores alone". Here a workgroup owns a strip of R output rows of one image:
//   1. the (R + 2) x (W + 2 pad) x C/g input rows it needs are copied ONCE into LDS with coalesced 16-byte buffer loads;
//      rows and columns outside the image arrive as zeros (out-of-range buffer offset), i.e. the zero padding is
//      materialised in the window and no tap needs a validity test afterwards;
//   2. forward: a wave multiplies 32-pixel tiles; the B operand of MFMA step st is ONE ds_read_b32 at
//      (pixel + hi * delta) + immediate -- the reduction index is ordered so that the two taps of a step lie a constant
//      distance apart (1 element, 1 row or 1 channel plane), hence four address registers per tile serve all 14 steps
//      and the only global-memory instructions left in the loop are the result stores;
//      dW: the im2col operand of a 16-pixel window is eight ds_read_b32 off one address register; dy is loaded in
//      fragment order with 16-byte buffer loads as before.
// The weights live in MFMA A-operand registers (forward) for the life of the workgroup. Reference semantics:
// bcnn_forward_conv_layer_cpu / bcnn_backward_conv_layer_cpu, bcnn_conv_layer.c:367-587 (im2col + gemm, add_bias quirk).
#include "conv_common.h"
#include "lds_dma.h"

#ifndef STORE_AUX
#define STORE_AUX 2  // nt: the result stream must not displace the input in L2 / Infinity Cache (conv_direct.hip)
#endif

namespace bcnn_hip {

constexpr int kWinOrg = 4;  // LDS column of image column 0 (16-byte aligned rows; columns < 4 hold the left padding)

// ---- reduction order of the forward kernel ---------------------------------------------------------------------
// step -> (tap of the lower half-wave, tap of the upper half-wave, kind of distance between them)
enum { WD_ELEM = 0, WD_ROW = 1, WD_PLANE = 2, WD_NONE = 3 };
struct WinStep { int c, kr, kc, kind, second; };  // second: the upper half-wave's tap exists
template <int CG>
struct WinSteps {
    static constexpr int N = (CG * 9 + 1) / 2;
    WinStep st[N];
    constexpr WinSteps() : st{} {
        int n = 0;
        for (int c = 0; c < CG; ++c)
            for (int kr = 0; kr < 3; ++kr) st[n++] = WinStep{c, kr, 0, WD_ELEM, 1};     // (kc 0 | kc 1)
        for (int c = 0; c < CG; ++c) st[n++] = WinStep{c, 0, 2, WD_ROW, 1};             // (kr 0 | kr 1) of kc 2
        for (int c = 0; c + 1 < CG; c += 2) st[n++] = WinStep{c, 2, 2, WD_PLANE, 1};    // (c | c + 1) of (kr 2, kc 2)
        if (CG & 1) st[n++] = WinStep{CG - 1, 2, 2, WD_NONE, 0};                        // the odd one out
    }
};

struct ConvWindowFwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* slopes;
    float* y;
    ConvShape s;
    int act, add_bias;
    int strips;  // ceil(OH / R)
};

// The window: [CG][R + 2][PITCH] floats, image row oh0 - pad + rr, image column L - kWinOrg.
template <int CG, int R, int PITCH>
__device__ __forceinline__ void window_fill(float* win, rsrc_i4 rx, const ConvShape& s, unsigned img_chan0, int oh0, int tid) {
    constexpr int ROWS = R + 2, P4 = PITCH / 4;
    for (int i = tid; i < CG * ROWS * P4; i += 256) {
        const int c = i / (ROWS * P4), rem = i - c * (ROWS * P4);
        const int rr = rem / P4, j = rem - rr * P4;
        const int ih = oh0 - s.pad + rr, iw0 = 4 * j - kWinOrg;
        const bool ok = (unsigned)ih < (unsigned)s.H && (unsigned)iw0 < (unsigned)s.W;  // W % 4 == 0: all four or none
        const unsigned off = ok ? ((img_chan0 + (unsigned)c) * (unsigned)s.HW + (unsigned)(ih * s.W + iw0)) * 4u : kOOB;
        const buf_f32x4 v = buffer_load_f32x4(rx, (int)off, 0, 0);
        *reinterpret_cast<buf_f32x4*>(win + (c * ROWS + rr) * PITCH + 4 * j) = v;
    }
}

// ================================================================================================
// forward
// ================================================================================================
template <int CG, int R, int PITCH, int TM, int ACTM>
__global__ __launch_bounds__(256, 4) void conv_fwd_window_kernel(const ConvWindowFwdArgs a) {
    constexpr int ROWS = R + 2, PLANE = ROWS * PITCH;
    constexpr WinSteps<CG> steps{};
    constexpr int NT = WinSteps<CG>::N;        // steps that read taps
    constexpr bool SPARE = (CG & 1) != 0;      // odd K: the last tap step's upper half-wave is free
    constexpr int KS = SPARE ? NT : NT + 1;    // + the bias: one more reduction row with B = 1 (like add_bias AFTER the gemm)
    __shared__ __attribute__((aligned(16))) float win[CG * PLANE];
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.y;
    const int n = (int)blockIdx.x / a.strips, strip = (int)blockIdx.x - n * a.strips;
    const int oh0 = strip * R;
    const int rows_here = (s.OH - oh0 < R) ? s.OH - oh0 : R;

    const rsrc_i4 rx = make_rsrc(a.x, (unsigned)((long long)s.N * s.C * s.HW * 4));
    const rsrc_i4 ry = make_rsrc(a.y, (unsigned)((long long)s.N * s.F * s.OHOW * 4));
    window_fill<CG, R, PITCH>(win, rx, s, (unsigned)(n * s.C + g * s.Cg), oh0, tid);
    // A operand: W[f = tm*32 + l31][tap of (step, half-wave)], zero where the tap or the filter does not exist; the bias
    // sits in the one reduction slot no tap uses
    const float* wg = a.w + (long long)g * s.Mg * s.K;
    float areg[TM][KS];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int f = tm * 32 + l31;
        float bv = 0.f;
        if (a.add_bias && f < s.Mg) {
            bv = a.bias[g * s.Mg + f];
            if (bv == 1.0f) bv = 0.f;  // bcnn_add_scalar (AVX build) adds nothing for exactly 1.0f (bcnn_mat.c:381-383)
        }
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            const WinStep p = steps.st[st];
            int c = p.c, kr = p.kr, kc = p.kc;
            if (hi) {
                if (p.kind == WD_ELEM) kc += 1;
                else if (p.kind == WD_ROW) kr += 1;
                else if (p.kind == WD_PLANE) c += 1;
            }
            const int k = c * 9 + kr * 3 + kc;
            const bool ok = (!hi || p.second) && f < s.Mg;
            const float v = wg[ok ? (long long)f * s.K + k : 0];
            areg[tm][st] = ok ? v : ((hi && !p.second) ? bv : 0.f);
        }
        if (!SPARE) areg[tm][KS - 1] = hi ? 0.f : bv;
    }
    __syncthreads();

    // lane constants: LDS byte offset of the lane's pixel in window row 0 for each distance kind, output byte offset
    const int base_col = kWinOrg - s.pad;
    const char* winb = reinterpret_cast<const char*>(win);
    int lds_lane[4];
    lds_lane[WD_ELEM] = 4 * (l31 + base_col + hi);
    lds_lane[WD_ROW] = 4 * (l31 + base_col + hi * PITCH);
    lds_lane[WD_PLANE] = 4 * (l31 + base_col + hi * PLANE);
    lds_lane[WD_NONE] = 4 * (l31 + base_col);
    const unsigned fstride = (unsigned)s.OHOW * 4u;
    const unsigned y_lane = (unsigned)l31 * 4u + 4u * (unsigned)hi * fstride;
    const unsigned y_img = ((unsigned)(n * s.F + g * s.Mg) * (unsigned)s.OHOW + (unsigned)(oh0 * s.OW)) * 4u;
    const int tpr = (s.OW + 31) >> 5;  // 32-pixel tiles per output row
    const int ntile = rows_here * tpr;
    const bool full_m = (s.Mg == TM * 32);
    const bool ragged_w = (s.OW & 31) != 0;

    int row = 0, ct = wid;
    while (ct >= tpr) { ct -= tpr; ++row; }
    // operands of the tile at (row, ct): NT ds_read_b32 off four address registers
    auto read_tile = [&](float (&b)[KS]) {
        const int soff = 4 * (row * PITCH + ct * 32);  // wave-uniform
        int vb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vb[i] = lds_lane[i] + soff;
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            const WinStep p = steps.st[st];
            b[st] = *reinterpret_cast<const float*>(winb + vb[p.kind] + 4 * (p.c * PLANE + p.kr * PITCH + p.kc));
        }
        if (SPARE) b[NT - 1] = hi ? 1.0f : b[NT - 1];
        else b[KS - 1] = 1.0f;
    };
    auto tile_out = [&]() -> unsigned {  // byte offset of the lane's pixel in channel 4*hi; beyond OW: dropped by address
        const unsigned o = y_img + (unsigned)(row * s.OW + ct * 32) * 4u + y_lane;
        return (ragged_w && ct * 32 + l31 >= s.OW) ? kOOB : o;
    };
    auto step_tile = [&]() {
        ct += 4;
        while (ct >= tpr) { ct -= tpr; ++row; }
    };
    auto compute_store = [&](const float (&b)[KS], unsigned ycur) {
        f32x16 acc[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
#pragma unroll
        for (int st = 0; st < KS; ++st)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[tm] = mfma32(areg[tm][st], b[st], acc[tm]);
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
            // one VGPR offset per tile; the channel stride rides in the scalar offset operand (+ 4*hi is in y_lane)
            if (full_m) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buffer_store_f32(v[r], ry, (int)ycur, (int)((unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * fs), STORE_AUX);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fr = tm * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned off = (fr + 4 * hi < s.Mg) ? ycur : kOOB;  // rows beyond F/groups are dropped
                    buffer_store_f32(v[r], ry, (int)off, (int)((unsigned)fr * fs), STORE_AUX);
                }
            }
        }
    };

    // two operand buffers: the next tile's LDS reads are issued before the current tile's MFMAs
    if (wid >= ntile) return;
    float bufA[KS], bufB[KS];
    read_tile(bufA);
    int t = wid;
#define WINDOW_STAGE(cur, nxt)                                       \
    {                                                                \
        const unsigned ycur = tile_out();                            \
        const bool more = (t + 4 < ntile); /* wave-uniform */        \
        if (more) {                                                  \
            step_tile();                                             \
            read_tile(nxt);                                          \
        }                                                            \
        compute_store(cur, ycur);                                    \
        if (!more) break;                                            \
        t += 4;                                                      \
    }
    for (;;) {
        WINDOW_STAGE(bufA, bufB)
        WINDOW_STAGE(bufB, bufA)
    }
#undef WINDOW_STAGE
}

static int window_pitch(const ConvShape& s) {
    const int need = ((s.OW + 31) & ~31) + 6;  // widest LDS column a tile / window lane touches, + 1
    if (need <= 232) return 232;
    if (need <= 264) return 264;
    return 0;
}

static bool window_ok(const ConvShape& s) {
    return !s.pointwise && s.ksz == 3 && s.stride == 1 && s.Cg >= 1 && s.Cg <= 3 && s.Mg <= 64 && s.pad <= 4 &&
           (s.W % 4) == 0 && s.total_q > 0 && window_pitch(s) != 0 && s.W + kWinOrg <= window_pitch(s) &&
           (long long)s.N * s.F * s.OHOW < (1LL << 29) && (long long)s.N * s.C * s.HW < (1LL << 29) && s.OH >= 1 && s.OW >= 1;
}

bool conv_forward_window(const float* x, const float* w, const float* bias, const float* slopes, float* y, const ConvShape& s,
                         int act, int raw) {
    if (!window_ok(s)) return false;
    constexpr int R = 8;
    ConvWindowFwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.slopes = slopes; a.y = y; a.s = s;
    a.act = raw ? BCNN_HIP_ACT_NONE : act;
    a.add_bias = raw ? 0 : 1;
    a.strips = ceil_div(s.OH, R);
    const dim3 grid((unsigned)(s.N * a.strips), (unsigned)s.groups);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    const int actm = (a.act == BCNN_HIP_ACT_NONE) ? 0 : (a.act == BCNN_HIP_ACT_RELU ? 1 : 2);
    const int pitch = window_pitch(s);
    KTimer kt(K_CONV_FWD, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
#define LAUNCH4(CGv, Pv, TMv, Av) conv_fwd_window_kernel<CGv, R, Pv, TMv, Av><<<grid, 256, 0, current_stream()>>>(a)
#define LAUNCH3(CGv, Pv, TMv) do { if (actm == 0) LAUNCH4(CGv, Pv, TMv, 0); else if (actm == 1) LAUNCH4(CGv, Pv, TMv, 1); \
                                   else LAUNCH4(CGv, Pv, TMv, 2); } while (0)
#define LAUNCH2(CGv, Pv) do { if (tm == 1) LAUNCH3(CGv, Pv, 1); else LAUNCH3(CGv, Pv, 2); } while (0)
#define LAUNCH1(CGv) do { if (pitch == 232) LAUNCH2(CGv, 232); else LAUNCH2(CGv, 264); } while (0)
    if (s.Cg == 1) LAUNCH1(1);
    else if (s.Cg == 2) LAUNCH1(2);
    else LAUNCH1(3);
#undef LAUNCH1
#undef LAUNCH2
#undef LAUNCH3
#undef LAUNCH4
    KERNEL_CHECK();
    return true;
}

// ================================================================================================
// dW (+ bias gradient)
// ================================================================================================
struct ConvWindowDwArgs {
    const float* x;
    const float* dy;
    float* partials;  // [nblocks][groups][TM*32][32]
    ConvShape s;
    int strips;             // per image
    int total_strips;       // N * strips
    int strips_per_block;
    int bias_col;
};

// Reduction over output pixels q; MFMA step e of a 16-pixel window pairs q0 + e (lower half-wave) with q0 + 8 + e
// (upper): a lane's eight dy values are 32 contiguous bytes (two 16-byte loads), its eight im2col values eight
// consecutive LDS floats. Column l31 = tap k (natural order), column K = all ones (bias gradient), beyond = zeros.
template <int CG, int R, int PITCH, int TM>
__global__ __launch_bounds__(256, 4) void conv_dw_window_kernel(const ConvWindowDwArgs a) {
    constexpr int ROWS = R + 2, PLANE = ROWS * PITCH;
    constexpr int RED = 3 * TM * 32 * 33;
    constexpr int WINF = CG * PLANE + 16;  // + eight ones, eight zeros
    __shared__ __attribute__((aligned(16))) float smem[WINF > RED ? WINF : RED];
    float* win = smem;
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.y;
    const rsrc_i4 rx = make_rsrc(a.x, (unsigned)((long long)s.N * s.C * s.HW * 4));
    const rsrc_i4 rdy = make_rsrc(a.dy, (unsigned)((long long)s.N * s.F * s.OHOW * 4));

    // this lane's im2col column
    const int K = CG * 9;
    const bool tap = l31 < K;
    int lds_lane;  // byte offset for window row 0, output column 0 (taps) or of the constant run (others)
    {
        const int c = l31 / 9, r9 = l31 - c * 9, kr = r9 / 3, kc = r9 - kr * 3;
        lds_lane = tap ? c * PLANE + kr * PITCH + kc + (kWinOrg - s.pad) + 8 * hi
                       : CG * PLANE + ((l31 == K && a.bias_col) ? 0 : 8);
        lds_lane *= 4;  // bytes
    }
    const char* winb = reinterpret_cast<const char*>(win);
    const unsigned moves = tap ? 1u : 0u;  // constant columns do not follow the window
    const unsigned fstride = (unsigned)s.OHOW * 4u;
    // dy: rows beyond F/groups are out of range for good (kOOB + anything below 2^31 stays out of range)
    unsigned dy_lane[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
        dy_lane[tm] = (tm * 32 + l31 < s.Mg) ? (unsigned)(tm * 32 + l31) * fstride + 32u * (unsigned)hi : kOOB;

    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

    const int wpr = (s.OW + 15) >> 4;  // 16-pixel windows per output row
    int sidx = blockIdx.x * a.strips_per_block;
    int s_end = sidx + a.strips_per_block;
    if (s_end > a.total_strips) s_end = a.total_strips;
    for (; sidx < s_end; ++sidx) {
        const int n = sidx / a.strips, strip = sidx - n * a.strips;
        const int oh0 = strip * R;
        const int rows_here = (s.OH - oh0 < R) ? s.OH - oh0 : R;
        __syncthreads();  // the previous strip's readers are done with the window
        window_fill<CG, R, PITCH>(win, rx, s, (unsigned)(n * s.C + g * s.Cg), oh0, tid);
        if (tid < 16) win[CG * PLANE + tid] = tid < 8 ? 1.0f : 0.0f;
        __syncthreads();
        const unsigned dy_img = ((unsigned)(n * s.F + g * s.Mg) * (unsigned)s.OHOW + (unsigned)(oh0 * s.OW)) * 4u;
        const int nwin = rows_here * wpr;
        int row = 0, cw = wid;
        while (cw >= wpr) { cw -= wpr; ++row; }
        for (int wdx = wid; wdx < nwin; wdx += 4) {
            const int ow0 = cw * 16;
            // a half window beyond OW (OW % 16 == 8) reads dy as zeros
            const unsigned dsc = dy_img + (unsigned)(row * s.OW + ow0) * 4u;
            const bool half_ok = ow0 + 8 * hi < s.OW;
            buf_f32x4 av[TM][2];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const unsigned off = half_ok ? dy_lane[tm] + dsc : kOOB;
                av[tm][0] = buffer_load_f32x4(rdy, (int)off, 0, 0);
                av[tm][1] = buffer_load_f32x4(rdy, (int)off + 16, 0, 0);
            }
            const int vb = lds_lane + (int)__umul24(moves, (unsigned)(4 * (row * PITCH + ow0)));
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = *reinterpret_cast<const float*>(winb + vb + 4 * e);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tm] = mfma32(av[tm][e >> 2][e & 3], bv[e], acc[tm]);
            cw += 4;
            while (cw >= wpr) { cw -= wpr; ++row; }
        }
    }

    // cross-wave reduction (waves 1..3 -> LDS -> wave 0), then one partial tile per workgroup
    __syncthreads();
    float(*red)[TM * 32][33] = reinterpret_cast<float(*)[TM * 32][33]>(smem);
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

// conv_direct.hip
void conv_dw_direct_finalize(const float* partials, int nparts, int groups, int Mg, int K, int MP, int bias_col, float* dw,
                             float* dbias);

static bool dw_window_ok(const ConvShape& s) { return window_ok(s) && (s.OW % 8) == 0; }

constexpr int kDwWinR = 7;

static void dw_window_plan(const ConvShape& s, int* strips, int* spb, int* blocks) {
    *strips = ceil_div(s.OH, kDwWinR);
    const int total = s.N * *strips;
    int b = kCUs * 4;
    if (b > total) b = total;
    *spb = ceil_div(total, b);
    *blocks = ceil_div(total, *spb);
}

size_t conv_dw_window_workspace_floats(const ConvShape& s) {
    if (!dw_window_ok(s)) return 0;
    int strips, spb, blocks;
    dw_window_plan(s, &strips, &spb, &blocks);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    return (size_t)blocks * s.groups * tm * 32 * 32;
}

// false: shape not covered. true: dW accumulated, and the bias gradient too when dbias != NULL.
bool conv_backward_weights_window(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s,
                                  float* workspace, size_t workspace_floats) {
    if (!dw_window_ok(s)) return false;
    int strips, spb, blocks;
    dw_window_plan(s, &strips, &spb, &blocks);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    const size_t need = (size_t)blocks * s.groups * tm * 32 * 32;
    if (workspace == nullptr || workspace_floats < need) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n", workspace_floats,
                need);
        exit(1);
    }
    KTimer kt(K_CONV_DW, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    ConvWindowDwArgs a;
    a.x = x; a.dy = dy; a.partials = workspace; a.s = s;
    a.strips = strips; a.total_strips = s.N * strips; a.strips_per_block = spb;
    a.bias_col = dbias ? 1 : 0;
    const dim3 grid((unsigned)blocks, (unsigned)s.groups);
    const int pitch = window_pitch(s);
#define LAUNCH3(CGv, Pv, TMv) conv_dw_window_kernel<CGv, kDwWinR, Pv, TMv><<<grid, 256, 0, current_stream()>>>(a)
#define LAUNCH2(CGv, Pv) do { if (tm == 1) LAUNCH3(CGv, Pv, 1); else LAUNCH3(CGv, Pv, 2); } while (0)
#define LAUNCH1(CGv) do { if (pitch == 232) LAUNCH2(CGv, 232); else LAUNCH2(CGv, 264); } while (0)
    if (s.Cg == 1) LAUNCH1(1);
    else if (s.Cg == 2) LAUNCH1(2);
    else LAUNCH1(3);
#undef LAUNCH1
#undef LAUNCH2
#undef LAUNCH3
    KERNEL_CHECK();
    conv_dw_direct_finalize(workspace, blocks, s.groups, s.Mg, s.K, tm * 32, a.bias_col, dw, dbias);
    return true;
}

}  // namespace bcnn_hip
