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
//      dW: see the second half of the file -- dy reaches LDS by LDS-DMA, one filter-plane row per instruction.
// The weights live in MFMA A-operand registers (forward) for the life of the workgroup. Reference semantics:
// bcnn_forward_conv_layer_cpu / bcnn_backward_conv_layer_cpu, bcnn_conv_layer.c:367-587 (im2col + gemm, add_bias quirk).
#include "conv_common.h"
#include "lds_dma.h"

#include <mutex>

// Tuning knobs (tools/exp/win_variants.sh builds the file with -D...): WABL_FWD_R (strip height), WABL_NW (multiplying waves),
// WABL_NBUF, ROWS_TM, ROWS_INTERLEAVE, DY_DMA_PLAIN. The timing-only switches of round 5 (WABL_NOSTORE / WABL_NOMFMA / WABL_PREFETCH,
// ROWS_ABL_NODMA / ROWS_ABL_NOMFMA: wrong results, right time) were taken out at the end of round 6; `git log -S ROWS_ABL_NODMA`.
#ifndef WABL_FWD_R
#define WABL_FWD_R 4  // output rows per strip (round 5: the loader wave hides the fill, so the halo rows of short strips cost nothing;
#endif                // forward-only loops 0.275 ms with 4 rows against 0.293 with 8, inside the fwd / dW alternation 0.305 / 0.31)
#ifndef WABL_NBUF
#define WABL_NBUF 2  // windows in LDS (loader form): the loader wave runs WABL_NBUF - 1 strips ahead
#endif
#ifndef WABL_NW
#define WABL_NW 8    // multiplying waves per workgroup of the forward kernel (4 / 7 / 12 / 15: +1 ... +4 %)
#endif
// Measured on configs[1] (tools/exp/win_variants.sh): one 8-wave workgroup per CU with both filter blocks 0.387 ms, two
// independent 4-wave workgroups (one filter block each) 0.345; requests issued in one burst at the top of a row beat
// requests spread behind the windows' MFMAs (0.345 / 0.415); non-temporal dy requests leave the dW time alone and keep
// the layer input in the Infinity Cache for the next forward pass (0.37 -> 0.34 ms).
// stem forward, measured (tools/exp/stem_variants.sh, N = 128): 4-row strips, 3 workgroups per CU 0.303 ms; 8-row strips, 2
// per CU 0.291; 2 rows 0.36; batches of 4 / 8 / 16 operand reads ahead of the MFMAs 0.292 / 0.291 / 0.296, the compiler's
// own read-wait-multiply schedule 0.30
#ifndef STEM_R
#define STEM_R 8
#endif
#ifndef STEM_WAVES
#define STEM_WAVES 2
#endif
#ifndef ROWS_TM
#define ROWS_TM 1
#endif
#ifndef ROWS_NS
#define ROWS_NS 2  // stages of the dW row ring (2: two workgroups per CU; more: one, rows requested ROWS_NS - 1 ahead)
#endif
#ifndef ROWS_INTERLEAVE
#define ROWS_BURST 1
#endif
#ifndef DY_DMA_PLAIN
#define DY_DMA_NT 1
#endif
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

void conv_prefetch_input(const float* x, const ConvShape& s, float* sink);  // conv_direct.hip

struct ConvWindowFwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* slopes;
    float* y;
    ConvShape s;
    int act, add_bias;
    int strips;  // ceil(OH / R)
    int* sched;  // {next unit beyond the grid's first round, workgroups done}: both 0 between launches (nullptr: static walk)
};

// The window: [CG][R + 2][PITCH] floats, image row oh0 - pad + rr, image column L - kWinOrg.
// ================================================================================================
// forward
// ================================================================================================
// window_fill by LDS-DMA, for the loader wave: piece i (16 bytes) of the window lies at LDS byte 16 * i, so instruction k of
// a wave deposits pieces 64 k ... 64 k + 63 in one 1 KB run; a lane's global offset is per piece (kOOB -> zeros: the padding)
template <int CG, int R, int PITCH>
__device__ __forceinline__ void window_fill_dma(unsigned lds_base, rsrc_i4 rx, const ConvShape& s, unsigned img_chan0, int oh0, int lane,
                                                bool valid) {
    constexpr int ROWS = R + 2, P4 = PITCH / 4, NP = CG * ROWS * P4;
#pragma unroll 4
    for (int k = 0; k < (NP + 63) / 64; ++k) {
        const int i = k * 64 + lane;
        const int c = i / (ROWS * P4), rem = i - c * (ROWS * P4);
        const int rr = rem / P4, j = rem - rr * P4;
        const int ih = oh0 - s.pad + rr, iw0 = 4 * j - kWinOrg;
        const bool ok = valid && i < NP && (unsigned)ih < (unsigned)s.H && (unsigned)iw0 < (unsigned)s.W;
        const unsigned off = ok ? ((img_chan0 + (unsigned)c) * (unsigned)s.HW + (unsigned)(ih * s.W + iw0)) * 4u : kOOB;
        dma_row_x4(rx, lds_base + (unsigned)k * 1024u, off, 0);
    }
}

// One persistent workgroup per CU: NW multiplying waves and one more wave that only moves input. While strip k is multiplied
// out of one window, the loader wave asks a device counter for the next strip, fills the other window with LDS-DMA and waits
// for it -- its vmcnt holds loads only, where a multiplying wave's counter is full of result stores that retire out of order
// with loads -- so a strip costs the multiplying waves ONE barrier and no fill, and short strips (4 output
// rows) cost nothing extra. 256 strips in flight instead of 1000+: fewer concurrent output streams is what the memory side
// wants here (round 5, warm clocks, same box: one 4-wave workgroup per 8-row strip 0.33 ms; 1280 persistent 4-wave
// workgroups 0.31-0.32; 256 x 8 waves + loader, 8-row strips 0.294; 4-row strips 0.275; the strips dealt out statically
// instead of by the counter +0.01-0.04).
template <int CG, int R, int PITCH, int TM, int ACTM, int NW>
__global__ __launch_bounds__(64 * (NW + 1), 2) void conv_fwd_window_kernel(const ConvWindowFwdArgs a) {
    constexpr int ROWS = R + 2, PLANE = ROWS * PITCH;
    constexpr int NBUF = WABL_NBUF;
    constexpr int WINSZ = (CG * PLANE + 255) & ~255;  // floats per window buffer, whole DMA instructions
    constexpr WinSteps<CG> steps{};
    constexpr int NT = WinSteps<CG>::N;        // steps that read taps
    constexpr bool SPARE = (CG & 1) != 0;      // odd K: the last tap step's upper half-wave is free
    constexpr int KS = SPARE ? NT : NT + 1;    // + the bias: one more reduction row with B = 1 (like add_bias AFTER the gemm)
    __shared__ __attribute__((aligned(1024))) float win_all[NBUF * WINSZ];
    __shared__ int next_unit[NBUF < 2 ? 2 : NBUF];
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.y;
    const rsrc_i4 rx = make_rsrc(a.x, (unsigned)((long long)s.N * s.C * s.HW * 4));
    const rsrc_i4 ry = make_rsrc(a.y, (unsigned)((long long)s.N * s.F * s.OHOW * 4));
    const int nunits = s.N * a.strips;
    if (wid == NW) {
        // ---- the loader wave: window k % NBUF and next_unit[k % NBUF] are ready before barrier k; the fill of strip
        // k + NBUF - 2 is requested right after barrier k - 1 (its buffer held strip k - 2), so a fill has NBUF - 1 strip
        // times to arrive. This wave's vmcnt counts nothing but its DMA instructions, which retire in order. ----
        constexpr int NI = (CG * (R + 2) * (PITCH / 4) + 63) / 64;  // DMA instructions per window
        constexpr int D = NBUF - 1;
        static_assert((D - 1) * NI < 64, "vmcnt is a 6-bit counter");
        int unit = (int)blockIdx.x;  // unit of the next strip to request
        int j = 0;                   // its index
        for (int k = 0;; ++k) {
            for (; j <= k + D - 1; ++j) {
                // beyond the last unit the same instructions run with every lane out of range (zeros into a window nobody
                // reads): the count the wait below relies on stays the same
                const int n = unit / a.strips, strip = unit - n * a.strips;
                window_fill_dma<CG, R, PITCH>(lds_offset(win_all + (j % NBUF) * WINSZ), rx, s, (unsigned)(n * s.C + g * s.Cg), strip * R, lane,
                                              unit < nunits);
                if (lane == 0) next_unit[j % NBUF] = unit;
                if (unit < nunits) {
                    if (NBUF == 2 && a.sched) unit = (int)gridDim.x + __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(a.sched + 2 * g, 1) : 0);
                    else unit += (int)gridDim.x;
                }
            }
            dma_wait_n<(D - 1) * NI>();
            lds_barrier();
            if (__builtin_amdgcn_readfirstlane(next_unit[k % NBUF]) >= nunits) break;
        }
        dma_wait();
        if (NBUF == 2 && a.sched && lane == 0) {  // the last workgroup of the launch to leave puts both counters back to zero
            if (atomicAdd(a.sched + 2 * g + 1, 1) == (int)gridDim.x - 1) {
                a.sched[2 * g] = 0;
                a.sched[2 * g + 1] = 0;
            }
        }
        return;
    }
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
    // lane constants: LDS byte offset of the lane's pixel in window row 0 for each distance kind, output byte offset
    const int base_col = kWinOrg - s.pad;
    int lds_lane[4];
    lds_lane[WD_ELEM] = 4 * (l31 + base_col + hi);
    lds_lane[WD_ROW] = 4 * (l31 + base_col + hi * PITCH);
    lds_lane[WD_PLANE] = 4 * (l31 + base_col + hi * PLANE);
    lds_lane[WD_NONE] = 4 * (l31 + base_col);
    const unsigned fstride = (unsigned)s.OHOW * 4u;
    const unsigned y_lane = (unsigned)l31 * 4u + 4u * (unsigned)hi * fstride;
    const int tpr = (s.OW + 31) >> 5;  // 32-pixel tiles per output row
    const bool full_m = (s.Mg == TM * 32);
    const bool ragged_w = (s.OW & 31) != 0;

    // ---- the multiplying waves: strip iter comes out of window iter % NBUF; which unit it is, the loader wave says ----
    int iter = 0;
    lds_barrier();  // barrier 0: the first window is in place
    for (int unit = (int)blockIdx.x; unit < nunits; ++iter) {
    const int n = unit / a.strips, strip = unit - n * a.strips;
    const int oh0 = strip * R;
    const int rows_here = (s.OH - oh0 < R) ? s.OH - oh0 : R;
    const char* winb = reinterpret_cast<const char*>(win_all + (iter % NBUF) * WINSZ);
    const unsigned y_img = ((unsigned)(n * s.F + g * s.Mg) * (unsigned)s.OHOW + (unsigned)(oh0 * s.OW)) * 4u;
    const int ntile = rows_here * tpr;

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
        ct += NW;
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
            for (int tm = 0; tm < TM; ++tm) {
                acc[tm] = mfma32(areg[tm][st], b[st], acc[tm]);
            }
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
    if (wid < ntile) {  // wave-uniform; the barriers are reached by every wave
    float bufA[KS], bufB[KS];
    read_tile(bufA);
    int t = wid;
#define WINDOW_STAGE(cur, nxt)                                       \
    {                                                                \
        const unsigned ycur = tile_out();                            \
        const bool more = (t + NW < ntile); /* wave-uniform */       \
        if (more) {                                                  \
            step_tile();                                             \
            read_tile(nxt);                                          \
        }                                                            \
        compute_store(cur, ycur);                                    \
        if (!more) break;                                            \
        t += NW;                                                     \
    }
    for (;;) {
        WINDOW_STAGE(bufA, bufB)
        WINDOW_STAGE(bufB, bufA)
    }
#undef WINDOW_STAGE
    }
    lds_barrier();  // barrier iter + 1: every wave is done with this window, the loader wave with the next one
    unit = next_unit[(iter + 1) % NBUF];
    }  // strips of this workgroup
}

// Counters of the dynamic strip walk: kSchedGroups pairs per (device, stream), zero when allocated and put back to zero by
// every launch that used them. One set per STREAM: launches of a stream run in order, so the launch that finds the pair has it
// to itself and finds it zero; launches on different streams never share one (a rotation over a global pool did, once enough
// launches lay between two of them). More than kSchedStreams streams per device: the later ones walk their strips statically
// (nullptr), which measured 0.01-0.04 ms slower on configs[1] and is always correct.
constexpr int kSchedStreams = 64, kSchedGroups = 8;
static int* window_sched_slot(int groups) {
    static int* base[64] = {};
    static hipStream_t owner[64][kSchedStreams] = {};
    static int owners[64] = {};
    static std::mutex mu;
    if (groups > kSchedGroups) return nullptr;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return nullptr;
    const hipStream_t st = current_stream();
    std::lock_guard<std::mutex> lock(mu);
    if (!base[dev]) {
        HIP_CHECK(hipMalloc((void**)&base[dev], sizeof(int) * 2 * kSchedStreams * kSchedGroups));
        HIP_CHECK(hipMemset(base[dev], 0, sizeof(int) * 2 * kSchedStreams * kSchedGroups));
    }
    int idx = -1;
    for (int i = 0; i < owners[dev]; ++i)
        if (owner[dev][i] == st) { idx = i; break; }
    if (idx < 0) {
        if (owners[dev] == kSchedStreams) return nullptr;
        idx = owners[dev]++;
        owner[dev][idx] = st;
    }
    return base[dev] + 2 * kSchedGroups * idx;
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
    constexpr int R = WABL_FWD_R;
    ConvWindowFwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.slopes = slopes; a.y = y; a.s = s;
    a.act = raw ? BCNN_HIP_ACT_NONE : act;
    a.add_bias = raw ? 0 : 1;
    a.strips = ceil_div(s.OH, R);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    const int actm = (a.act == BCNN_HIP_ACT_NONE) ? 0 : (a.act == BCNN_HIP_ACT_RELU ? 1 : 2);
    const int pitch = window_pitch(s);
    KTimer kt(K_CONV_FWD, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    // One workgroup per CU (fewer when there are fewer strips); strips beyond the first round are handed out by a device counter.
    const int nunits = s.N * a.strips;
    a.sched = window_sched_slot(s.groups);
    int res = kCUs / s.groups > 0 ? kCUs / s.groups : 1;  // blockIdx.y = group
#ifdef BCNN_HIP_EXPERIMENT
    if (const char* genv = getenv("BCNN_HIP_WINDOW_GRID")) { if (atoi(genv) > 0) res = atoi(genv); }   // read per call: A/B inside one
    if (const char* denv = getenv("BCNN_HIP_WINDOW_DYN")) { if (atoi(denv) == 0) a.sched = nullptr; }  // process (tools/exp/placement.py)
#endif
    const dim3 grid((unsigned)(nunits < res ? nunits : res), (unsigned)s.groups);
#define LAUNCH4(CGv, Pv, TMv, Av) conv_fwd_window_kernel<CGv, R, Pv, TMv, Av, WABL_NW><<<grid, 64 * (WABL_NW + 1), 0, current_stream()>>>(a)
#define LAUNCH3(CGv, Pv, TMv) do { if (actm == 0) LAUNCH4(CGv, Pv, TMv, 0); else if (actm == 1) LAUNCH4(CGv, Pv, TMv, 1); \
                                   else LAUNCH4(CGv, Pv, TMv, 2); } while (0)
#define LAUNCH2(CGv, Pv) do { if (tm == 1) LAUNCH3(CGv, Pv, 1); else LAUNCH3(CGv, Pv, 2); } while (0)
#define LAUNCH1(CGv) do { if (pitch == 232) LAUNCH2(CGv, 232); else LAUNCH2(CGv, 264); } while (0)
    trace_kernel("conv_fwd_window_kernel");
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
// forward, 7x7 / stride 2 (the RGB stem of ResNet-18: 3 -> 64 channels, 224^2 -> 112^2)
// ================================================================================================
// The same idea for the other few-channel layer of the benchmark. On the LDS-DMA GEMM (conv_igemm_dma.hip, "rowmode") the
// stem runs at 0.51 of the fp32-MFMA peak: a 64 x 64 tile re-stages its 160 x 64 weight rows by DMA for every 64 pixels
// and gathers 160 B rows. Here the weights live in registers (a wave owns 32 filters: 74 A operands) and a workgroup stages
// the 13 input rows of 4 output rows once (36 KB); the 147 taps + the bias fill 74 MFMA steps exactly, ordered as above
// -- (kc, kc+1) pairs, then the (kr, kr+1) pairs of column 6, then (c, c+1) of tap (6, 6) -- so that four address
// registers + immediates serve all of them. Tiles are 32 CONSECUTIVE output pixels of the strip (output rows are
// contiguous in a plane), so OW = 112 leaves no ragged tile; a lane's window origin follows from its (row, column).
// With a fused batch-norm behind it (raw output) the kernel also emits the per-channel sum / sum of squares of what it
// stores: per-lane partial sums over the wave's tiles, one half-wave DPP reduction per workgroup.
template <int CG>
struct StemSteps {  // taps of 7x7 filters; see WinSteps
    static constexpr int N = (CG * 49 + 1) / 2;
    WinStep st[N];
    constexpr StemSteps() : st{} {
        int n = 0;
        for (int c = 0; c < CG; ++c)
            for (int kr = 0; kr < 7; ++kr)
                for (int kc = 0; kc < 6; kc += 2) st[n++] = WinStep{c, kr, kc, WD_ELEM, 1};
        for (int c = 0; c < CG; ++c)
            for (int kr = 0; kr < 6; kr += 2) st[n++] = WinStep{c, kr, 6, WD_ROW, 1};
        for (int c = 0; c + 1 < CG; c += 2) st[n++] = WinStep{c, 6, 6, WD_PLANE, 1};
        if (CG & 1) st[n++] = WinStep{CG - 1, 6, 6, WD_NONE, 0};
    }
};

struct ConvStemFwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* slopes;
    float* y;
    float* stats;  // optional [F][splits][2]
    ConvShape s;
    int act, add_bias;
    int strips;    // ceil(OH / R)
    int splits;    // N * strips * 2
};

// half-wave sums on the DPP path: lanes 31 and 63 end up with the totals of lanes 0-31 / 32-63
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror: 16-lane sums
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1, 3
    return v;
}

template <int CG, int R, int PITCH, int ACTM, bool STATS>
__global__ __launch_bounds__(256, STEM_WAVES) void conv_fwd_stem_kernel(const ConvStemFwdArgs a) {
    constexpr int KSZ = 7, S = 2;
    constexpr int ROWS = S * (R - 1) + KSZ, PLANE = ROWS * PITCH;
    constexpr StemSteps<CG> steps{};
    constexpr int NT = StemSteps<CG>::N;
    constexpr bool SPARE = (CG & 1) != 0;
    constexpr int KS = SPARE ? NT : NT + 1;
    __shared__ __attribute__((aligned(16))) float win[CG * PLANE];
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int fh = wid & 1, tpar = wid >> 1;  // this wave's block of 32 filters, its parity of tiles
    const int g = blockIdx.y;
    const int n = (int)blockIdx.x / a.strips, strip = (int)blockIdx.x - n * a.strips;
    const int oh0 = strip * R;
    const int rows_here = (s.OH - oh0 < R) ? s.OH - oh0 : R;
    const rsrc_i4 rx = make_rsrc(a.x, (unsigned)((long long)s.N * s.C * s.HW * 4));
    const rsrc_i4 ry = make_rsrc(a.y, (unsigned)((long long)s.N * s.F * s.OHOW * 4));

    // window: input rows S*oh0 - pad .. + ROWS, columns -kWinOrg .. ; out-of-image floats arrive as zeros
    {
        constexpr int P4 = PITCH / 4;
        const unsigned chan0 = (unsigned)(n * s.C + g * s.Cg);
        for (int i = tid; i < CG * ROWS * P4; i += 256) {
            const int c = i / (ROWS * P4), rem = i - c * (ROWS * P4);
            const int rr = rem / P4, j = rem - rr * P4;
            const int ih = S * oh0 - s.pad + rr, iw0 = 4 * j - kWinOrg;
            const bool ok = (unsigned)ih < (unsigned)s.H && (unsigned)iw0 < (unsigned)s.W;
            const unsigned off = ok ? ((chan0 + (unsigned)c) * (unsigned)s.HW + (unsigned)(ih * s.W + iw0)) * 4u : kOOB;
            *reinterpret_cast<buf_f32x4*>(win + (c * ROWS + rr) * PITCH + 4 * j) = buffer_load_f32x4(rx, (int)off, 0, 0);
        }
    }
    // A operand of this wave's 32 filters
    const float* wg = a.w + (long long)g * s.Mg * s.K;
    float areg[KS];
    {
        const int f = fh * 32 + l31;
        float bv = 0.f;
        if (a.add_bias && f < s.Mg) {
            bv = a.bias[g * s.Mg + f];
            if (bv == 1.0f) bv = 0.f;  // quirk 2 (bcnn_mat.c:381-383)
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
            const int k = c * 49 + kr * 7 + kc;
            const bool ok = (!hi || p.second) && f < s.Mg;
            const float v = wg[ok ? (long long)f * s.K + k : 0];
            areg[st] = ok ? v : ((hi && !p.second) ? bv : 0.f);
        }
        if (!SPARE) areg[KS - 1] = hi ? 0.f : bv;
    }
    __syncthreads();

    const char* winb = reinterpret_cast<const char*>(win);
    const int base_col = kWinOrg - s.pad;
    const unsigned fstride = (unsigned)s.OHOW * 4u;
    const unsigned y_lane = 4u * (unsigned)hi * fstride + (unsigned)(fh * 32) * fstride;
    const unsigned y_img = ((unsigned)(n * s.F + g * s.Mg) * (unsigned)s.OHOW + (unsigned)(oh0 * s.OW)) * 4u;
    const int npix = rows_here * s.OW;           // output pixels of the strip, contiguous in every plane
    const int ntile = (npix + 31) >> 5;
    const bool full_m = (s.Mg - fh * 32 >= 32);
    const bool wave_on = fh * 32 < s.Mg;

    float ssum[16], ssq[16];
    if (STATS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ssum[r] = ssq[r] = 0.f;
    }
    if (wave_on) {
        for (int t = tpar; t < ntile; t += 2) {
            const int q = t * 32 + l31;
            const bool valid = q < npix;
            int row = 0, ow = valid ? q : 0;  // R is small: the row by comparison, no division
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const bool past = ow >= s.OW;
                row += past ? 1 : 0;
                ow -= past ? s.OW : 0;
            }
            const int origin = 4 * (S * row * PITCH + S * ow + base_col);
            int vb[4];
            vb[WD_ELEM] = origin + 4 * hi;
            vb[WD_ROW] = origin + 4 * hi * PITCH;
            vb[WD_PLANE] = origin + 4 * hi * PLANE;
            vb[WD_NONE] = origin;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // B operands in batches of SB steps, one batch ahead of the MFMAs that consume them (left to itself the
            // compiler reads two values, waits for them, multiplies twice: the LDS latency shows after every pair)
#ifndef STEM_SB
#define STEM_SB 8
#endif
            constexpr int SB = STEM_SB, NB = (NT + SB - 1) / SB;
            float bq[2][SB];
            auto read_batch = [&](int bi, float (&dst)[SB]) {
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int st = bi * SB + j;
                    if (st < NT) {
                        const WinStep p = steps.st[st];
                        dst[j] = *reinterpret_cast<const float*>(winb + vb[p.kind] + 4 * (p.c * PLANE + p.kr * PITCH + p.kc));
                    }
                }
            };
            read_batch(0, bq[0]);
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                if (bi + 1 < NB) read_batch(bi + 1, bq[(bi + 1) & 1]);
#ifndef STEM_FREE_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int st = bi * SB + j;
                    if (st < NT) {
                        float b = bq[bi & 1][j];
                        if (SPARE && st == NT - 1) b = hi ? 1.0f : b;
                        acc = mfma32(areg[st], b, acc);
                    }
                }
#ifndef STEM_FREE_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (!SPARE) acc = mfma32(areg[KS - 1], 1.0f, acc);

            const unsigned ycur = valid ? y_img + (unsigned)q * 4u + y_lane : kOOB;
            unsigned fs = fstride;
            asm volatile("" : "+s"(fs));
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r];
            if (ACTM == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] * (float)(v[r] > 0);
            } else if (ACTM == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = fh * 32 + mfma_row(r, lane);
                    const float sl = (a.act == BCNN_HIP_ACT_PRELU && f < s.Mg) ? a.slopes[g * s.Mg + f] : 0.f;
                    v[r] = act_fwd_cheap(v[r], a.act, sl);
                }
            }
            if (STATS) {
                const float m = valid ? 1.f : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t0 = v[r] * m;
                    ssum[r] += t0;
                    ssq[r] += t0 * t0;
                }
            }
            if (full_m) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buffer_store_f32(v[r], ry, (int)ycur, (int)((unsigned)((r & 3) + 8 * (r >> 2)) * fs), 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fr = (r & 3) + 8 * (r >> 2);
                    const unsigned off = (fh * 32 + fr + 4 * hi < s.Mg) ? ycur : kOOB;
                    buffer_store_f32(v[r], ry, (int)off, (int)((unsigned)fr * fs), 0);
                }
            }
        }
    }
    if (STATS && wave_on) {  // slot of this wave: (image, strip, tile parity)
        const int slot = ((int)blockIdx.x) * 2 + tpar;
        const rsrc_i4 rst = make_rsrc(a.stats, (unsigned)((size_t)s.F * a.splits * 2 * sizeof(float)));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = half_wave_sum_dpp(ssum[r]), sq = half_wave_sum_dpp(ssq[r]);
            const int f = fh * 32 + mfma_row(r, lane);  // lanes 31 / 63 hold the sums of their half's filter
            const unsigned off = (l31 == 31 && f < s.Mg) ? (unsigned)(((g * s.Mg + f) * a.splits + slot) * 8) : kOOB;
            buffer_store_f32x2(buf_f32x2{sv, sq}, rst, (int)off, 0, 0);
        }
    }
}

static bool stem_ok(const ConvShape& s) {
    return !s.pointwise && s.ksz == 7 && s.stride == 2 && s.pad <= 4 && s.Cg >= 1 && s.Cg <= 3 && s.Mg <= 64 &&
           (s.W % 4) == 0 && s.W + kWinOrg <= 232 && 2 * (s.OW - 1) + 7 + (kWinOrg - s.pad) <= 232 && s.total_q > 0 &&
           (long long)s.N * s.F * s.OHOW < (1LL << 29) && (long long)s.N * s.C * s.HW < (1LL << 29);
}

// stats (optional, raw mode only): room for F * splits * 2 floats is checked against stats->capacity
bool conv_forward_stem(const float* x, const float* w, const float* bias, const float* slopes, float* y, const ConvShape& s,
                       int act, int raw, ConvStats* stats) {
    if (stats) stats->splits = 0;
    if (!stem_ok(s)) return false;
    constexpr int R = STEM_R;
    ConvStemFwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.slopes = slopes; a.y = y; a.s = s;
    a.act = raw ? BCNN_HIP_ACT_NONE : act;
    a.add_bias = raw ? 0 : 1;
    a.strips = ceil_div(s.OH, R);
    a.splits = s.N * a.strips * 2;
    const bool want_stats = stats && raw && stats->partials && (size_t)s.F * a.splits * 2 <= stats->capacity;
    a.stats = want_stats ? stats->partials : nullptr;
    const dim3 grid((unsigned)(s.N * a.strips), (unsigned)s.groups);
    const int actm = (a.act == BCNN_HIP_ACT_NONE) ? 0 : (a.act == BCNN_HIP_ACT_RELU ? 1 : 2);
    KTimer kt(K_CONV_FWD, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
#define SLAUNCH3(CGv, Av, STv) conv_fwd_stem_kernel<CGv, R, 232, Av, STv><<<grid, 256, 0, current_stream()>>>(a)
#define SLAUNCH2(CGv) do { if (want_stats) SLAUNCH3(CGv, 0, true); else if (actm == 0) SLAUNCH3(CGv, 0, false); \
                           else if (actm == 1) SLAUNCH3(CGv, 1, false); else SLAUNCH3(CGv, 2, false); } while (0)
    trace_kernel("conv_fwd_stem_kernel");
    if (s.Cg == 1) SLAUNCH2(1);
    else if (s.Cg == 2) SLAUNCH2(2);
    else SLAUNCH2(3);
#undef SLAUNCH2
#undef SLAUNCH3
    KERNEL_CHECK();
    if (want_stats) stats->splits = a.splits;
    return true;
}

// ================================================================================================
// dW (+ bias gradient), row-streamed: dy through LDS by LDS-DMA
// ================================================================================================
// A first version kept the LDS-free kernel's dy path (conv_direct.hip: 16-byte loads in MFMA fragment order) next to the
// input window in LDS: one such load instruction touches 32 filter planes with 2 x 16 bytes each, HBM sees 64-byte pieces
// scattered over 64 pages, and the kernel took 0.41 ms where the same bytes read as contiguous 1 KB runs took 0.33 and
// cache hits 0.26 (timing-only variants). Here ONE instruction moves one output row of one filter plane -- up to 1 KB
// contiguous -- straight into LDS (`buffer_load_dwordx4 ... lds`: no registers, no vector ALU, any number in flight), and
// the fragments are read back with conflict-free ds_read_b128 (row pitch = 4 x odd floats). A persistent workgroup of four
// waves owns 32 filters and a share of the N*OH output rows, which it walks through a two-stage ring: {32 dy rows, the
// 3 x C/g input rows} of row r + 1 are requested -- two instructions behind every window's MFMAs -- while row r is
// multiplied; the 8-pixel windows of a row are dealt to the four waves. Two such workgroups fit a CU (75 KB of LDS each)
// and run out of phase: one's wait for its last rows to land is the other's multiply time.
struct ConvRowsDwArgs {
    const float* x;
    const float* dy;
    float* partials;  // [nblocks][groups][MP][32], MP = 32 * filter blocks
    ConvShape s;
    int total_rows;      // N * OH
    int rows_per_block;
    int bias_col;
    int mp;
};

template <int CG, int PD, int PX, int TM>
__global__ __launch_bounds__(256 * TM, 1) void conv_dw_rows_kernel(const ConvRowsDwArgs a) {
    constexpr int NF = 32 * TM, NW = 4 * TM;        // filters per workgroup, waves (4 per block of 32 filters)
    constexpr int DYS = NF * PD, XS = CG * 3 * PX;  // floats per stage
    constexpr int XREQ = (CG * 3 + NW - 1) / NW, NREQ = 8 + XREQ;  // LDS-DMA instructions per wave and row
    constexpr int RED = TM * 3 * 32 * 33, NS = (PD > 228 && ROWS_NS > 3) ? 3 : ROWS_NS;  // the wide pitch has room for three
    static_assert(2 * DYS >= RED, "the final reduction aliases the dy stages");
    __shared__ __attribute__((aligned(16))) float lds[NS * DYS + NS * XS + 16];
    float* xS = lds + NS * DYS;
    float* cst = xS + NS * XS;  // eight ones, eight zeros
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.y;
    const int fh = wid >> 2, slot = wid & 3;         // filter block inside the workgroup, window slot
    const int fb = blockIdx.z * TM + fh;             // block of 32 filters inside the group
    const rsrc_i4 rx = make_rsrc(a.x, (unsigned)((long long)s.N * s.C * s.HW * 4));
    const rsrc_i4 rdy = make_rsrc(a.dy, (unsigned)((long long)s.N * s.F * s.OHOW * 4));

    // the input stages' padding columns are never written again: zero them (and everything else) once
    for (int i = tid; i < NS * XS; i += 256 * TM) xS[i] = 0.f;
    if (tid < 16) cst[tid] = tid < 8 ? 1.0f : 0.0f;
    __syncthreads();

    // lane constants (byte offsets into lds[])
    const int K = CG * 9;
    const bool tap = l31 < K;
    const unsigned a_lane = (unsigned)(((fh * 32 + l31) * PD + 4 * hi) * 4);
    unsigned b_lane, b_step;  // taps follow the window, the constant columns do not
    {
        const int c = l31 / 9, r9 = l31 - c * 9, kr = r9 / 3, kc = r9 - kr * 3;
        b_lane = tap ? (unsigned)((NS * DYS + (c * 3 + kr) * PX + kc + (kWinOrg - s.pad) + 4 * hi) * 4)
                     : (unsigned)((NS * DYS + NS * XS + ((l31 == K && a.bias_col) ? 0 : 8)) * 4);
        b_step = tap ? 1u : 0u;
    }
    const char* ldsb = reinterpret_cast<const char*>(lds);
    const unsigned lds0 = lds_offset(lds);
    const unsigned dma_lane = (unsigned)lane * 16u;
    const bool dy_lane_on = lane * 4 < s.OW, x_lane_on = lane * 4 < s.W;

    // request number k (wave-uniform, 0 .. NREQ-1) of this wave for output row (n, oh) -> stage `buf`:
    // k < 8: filter row wid*8 + k of the workgroup's dy rows; k >= 8: input row (c, rr) number wid + NW*(k-8), clamped (a repeat writes the
    // same bytes to the same place, which keeps the instruction count per wave and row constant for the vmcnt waits)
    auto request = [&](int k, int n, int oh, int buf) {
        if (k < 8) {
            const int f = blockIdx.z * NF + wid * 8 + k;
            const bool fok = f < s.Mg;
            const unsigned soff = __builtin_amdgcn_readfirstlane(
                fok ? ((unsigned)(n * s.F + g * s.Mg + f) * (unsigned)s.OHOW + (unsigned)(oh * s.OW)) * 4u : 0u);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * DYS + (wid * 8 + k) * PD) * 4));
            const unsigned voff = fok ? dma_lane : kOOB;
            if (dy_lane_on) {
#ifdef DY_DMA_NT
                dma_row_x4_nt(rdy, dst, voff, soff);
#else
                dma_row_x4(rdy, dst, voff, soff);
#endif
            }
        } else {
            const int i0 = wid + NW * (k - 8), i = i0 < CG * 3 ? i0 : CG * 3 - 1;
            const int c = i / 3, rr = i - c * 3;
            const int ih = oh - s.pad + rr;
            const bool ok = (unsigned)ih < (unsigned)s.H;
            const unsigned soff = __builtin_amdgcn_readfirstlane(
                ((unsigned)(n * s.C + g * s.Cg + c) * (unsigned)s.HW + (unsigned)((ok ? ih : 0) * s.W)) * 4u);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((NS * DYS + buf * XS + i * PX + kWinOrg) * 4));
            const unsigned voff = ok ? dma_lane : kOOB;
            if (x_lane_on) dma_row_x4(rx, dst, voff, soff);
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nwin = s.OW >> 3;  // 8-pixel windows per row: step e pairs pixel w*8 + e with w*8 + 4 + e
    const int r0 = blockIdx.x * a.rows_per_block;
    int r1 = r0 + a.rows_per_block;
    if (r1 > a.total_rows) r1 = a.total_rows;

    for (int q = 0; q < NS - 1; ++q)  // the first NS - 1 rows into stages 0 .. NS - 2
        if (r0 + q < r1) {
            const int n = (r0 + q) / s.OH, oh = (r0 + q) - n * s.OH;
            for (int k = 0; k < NREQ; ++k) request(k, n, oh, q);
        }
    dma_wait();
    lds_barrier();
    int buf = 0;
    for (int r = r0; r < r1; ++r) {
        const int nbuf = buf == 0 ? NS - 1 : buf - 1;  // the stage read during row r - 1: free for row r + NS - 1
        const int nn = (r + NS - 1) / s.OH, noh = (r + NS - 1) - nn * s.OH;  // the row requested during this one
        const int kend = (r + NS - 1 < r1) ? NREQ : 0;
        int k = 0;
#ifdef ROWS_BURST
        while (k < kend) { request(k, nn, noh, nbuf); ++k; }
#endif
        unsigned va = a_lane + (unsigned)(buf * DYS * 4) + (unsigned)(slot * 32);
        unsigned vb = b_lane + b_step * (unsigned)(buf * XS * 4 + slot * 32);
        auto read_window = [&](buf_f32x4& av, float (&bv)[4]) {
            av = *reinterpret_cast<const buf_f32x4*>(ldsb + va);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = *reinterpret_cast<const float*>(ldsb + vb + 4 * e);
            va += 128u;
            vb += b_step * 128u;
        };
        if (slot < nwin) {  // two operand sets: the next window's LDS reads are issued before this window's MFMAs
            buf_f32x4 avA, avB;
            float bvA[4], bvB[4];
            read_window(avA, bvA);
            int w = slot;
#define ROWS_MFMA(ca, cb) _Pragma("unroll") for (int e = 0; e < 4; ++e) acc = mfma32(ca[e], cb[e], acc);
#define ROWS_STAGE(ca, cb, na, nb)                                                                 \
            {                                                                                      \
                const bool more = (w + 4 < nwin); /* uniform */                                    \
                if (more) read_window(na, nb);                                                     \
                ROWS_MFMA(ca, cb)                                                                  \
                /* two requests for the next row in the shadow of the last MFMA */                 \
                if (k < kend) { request(k, nn, noh, nbuf); ++k; }                               \
                if (k < kend) { request(k, nn, noh, nbuf); ++k; }                               \
                if (!more) break;                                                                  \
                w += 4;                                                                            \
            }
            for (;;) {
                ROWS_STAGE(avA, bvA, avB, bvB)
                ROWS_STAGE(avB, bvB, avA, bvA)
            }
#undef ROWS_STAGE
#undef ROWS_MFMA
        }
        while (k < kend) { request(k, nn, noh, nbuf); ++k; }  // narrow rows: fewer windows than requests
        // this wave's requests for row r + 1 have landed (the rows behind it may fly on: vmcnt retires in order, every row
        // is NREQ instructions per wave) ...
        if (NS > 2) {
            const int ahead = min(r + NS - 1, r1 - 1) - (r + 1);  // rows requested after row r + 1
            if (NS > 3 && ahead >= 2) dma_wait_n<2 * NREQ>();
            else if (ahead >= 1) dma_wait_n<NREQ>();
            else dma_wait();
        } else {
            dma_wait();
        }
        lds_barrier();  // ... and so have everybody's; everybody is done reading stage `buf`
        buf = buf + 1 == NS ? 0 : buf + 1;
    }

    // cross-wave reduction (slots 1..3 -> LDS -> slot 0), then this filter block's rows of the workgroup's partial tile
    float(*red)[3][32][33] = reinterpret_cast<float(*)[3][32][33]>(lds);
    if (slot > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[fh][slot - 1][mfma_row(r, lane)][l31] = acc[r];
    }
    __syncthreads();
    if (slot == 0) {
        float* out = a.partials + ((size_t)blockIdx.x * s.groups + g) * a.mp * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = mfma_row(r, lane);
            out[(fb * 32 + fr) * 32 + l31] = ((acc[r] + red[fh][0][fr][l31]) + red[fh][1][fr][l31]) + red[fh][2][fr][l31];
        }
    }
}

// ================================================================================================
// dW (+ bias gradient) of the 7x7 / stride-2 stem, row-streamed
// ================================================================================================
// dW[f][tap] = sum over output pixels of dy[f][q] * x[tap's input pixel of q]: 64 x 147 (+ 1 bias column) outputs, reduction
// over N*OH*OW = 1.6 M pixels. On the LDS-DMA weight-gradient GEMM ("rowmode") the layer runs at 0.40 of the fp32-MFMA
// peak. Here a persistent workgroup of 8 waves per CU walks its share of the N*OH output rows through the two-stage LDS
// ring of conv_dw_rows_kernel (one `buffer_load_dwordx4 ... lds` per dy row of a filter plane and per input row, all
// requested in one burst a row ahead) and multiplies with v_mfma_f32_16x16x4_f32: wave (fb, th) owns filters
// 16 fb .. 16 fb + 15 and taps 80 th .. 80 th + 79 -- five 16 x 16 accumulators, 20 registers -- so the 64 x 160 result
// needs no cross-wave reduction and every wave does the same work on every row (7 windows of 16 pixels x 20 MFMAs).
// The reduction index of a window is permuted so that a lane's four dy values are 16 contiguous bytes (ds_read_b128,
// conflict-free at a row pitch of 120 floats); its four im2col values per tap block are four ds_read_b32 at stride 2.
struct ConvStemDwArgs {
    const float* x;
    const float* dy;
    float* partials;  // [nblocks][groups][64][160]
    ConvShape s;
    int total_rows;      // N * OH
    int rows_per_block;
    int bias_col;
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kStemTapCols = 160;  // 10 blocks of 16: 147 taps, the ones column, zeros

template <int CG, int PD, int PX>
__global__ __launch_bounds__(512, 1) void conv_dw_stem_kernel(const ConvStemDwArgs a) {
    constexpr int XR = CG * 7;                          // input rows per stage
    constexpr int DYS = 64 * PD, XS = XR * PX;          // floats per stage
    constexpr int XREQ = (XR + 7) / 8, NREQ = 8 + XREQ; // LDS-DMA instructions per wave and row
    __shared__ __attribute__((aligned(16))) float lds[2 * DYS + 2 * XS + 16];
    float* cst = lds + 2 * DYS + 2 * XS;  // eight ones, eight zeros
    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int fb = wid & 3, th = wid >> 2;
    const int g = blockIdx.y;
    const rsrc_i4 rx = make_rsrc(a.x, (unsigned)((long long)s.N * s.C * s.HW * 4));
    const rsrc_i4 rdy = make_rsrc(a.dy, (unsigned)((long long)s.N * s.F * s.OHOW * 4));

    // padding columns of both kinds of stage are never written again (dy beyond OW, x outside the image): zero everything once
    for (int i = tid; i < 2 * DYS + 2 * XS; i += 512) lds[i] = 0.f;
    if (tid < 16) cst[tid] = tid < 8 ? 1.0f : 0.0f;
    __syncthreads();

    const int K = CG * 49;
    const unsigned a_lane = (unsigned)(((fb * 16 + l15) * PD + 4 * kq) * 4);
    unsigned b_lane[5], b_step[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const int tap = (th * 5 + b) * 16 + l15;
        const int c = tap / 49, r49 = tap - c * 49, kr = r49 / 7, kc = r49 - kr * 7;
        const bool is_tap = tap < K;
        // x[c][2 oh - pad + kr][2 px - pad + kc] with px = 16 w + 4 kq + e  ->  row (c, kr) of the stage, column 2 px + kc + org - pad
        b_lane[b] = is_tap ? (unsigned)((2 * DYS + (c * 7 + kr) * PX + kc + (kWinOrg - s.pad) + 8 * kq) * 4)
                           : (unsigned)((2 * DYS + 2 * XS + ((tap == K && a.bias_col) ? 0 : 8)) * 4);
        b_step[b] = is_tap ? 1u : 0u;
    }
    const char* ldsb = reinterpret_cast<const char*>(lds);
    const unsigned lds0 = lds_offset(lds);
    const unsigned dma_lane = (unsigned)lane * 16u;
    const bool dy_lane_on = lane * 4 < s.OW, x_lane_on = lane * 4 < s.W;

    auto request = [&](int k, int n, int oh, int buf) {
        if (k < 8) {
            const int f = wid * 8 + k;
            const bool fok = f < s.Mg;
            const unsigned soff = __builtin_amdgcn_readfirstlane(
                fok ? ((unsigned)(n * s.F + g * s.Mg + f) * (unsigned)s.OHOW + (unsigned)(oh * s.OW)) * 4u : 0u);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * DYS + f * PD) * 4));
            const unsigned voff = fok ? dma_lane : kOOB;
            if (dy_lane_on) dma_row_x4_nt(rdy, dst, voff, soff);
        } else {
            const int i0 = wid + 8 * (k - 8), i = i0 < XR ? i0 : XR - 1;
            const int c = i / 7, kr = i - c * 7;
            const int ih = 2 * oh - s.pad + kr;
            const bool ok = (unsigned)ih < (unsigned)s.H;
            const unsigned soff = __builtin_amdgcn_readfirstlane(
                ((unsigned)(n * s.C + g * s.Cg + c) * (unsigned)s.HW + (unsigned)((ok ? ih : 0) * s.W)) * 4u);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((2 * DYS + buf * XS + i * PX + kWinOrg) * 4));
            const unsigned voff = ok ? dma_lane : kOOB;
            if (x_lane_on) dma_row_x4(rx, dst, voff, soff);
        }
    };

    f32x4_t acc[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nwin = (s.OW + 15) >> 4;
    const int r0 = blockIdx.x * a.rows_per_block;
    int r1 = r0 + a.rows_per_block;
    if (r1 > a.total_rows) r1 = a.total_rows;

    if (r0 < r1) {
        const int n = r0 / s.OH, oh = r0 - n * s.OH;
        for (int k = 0; k < NREQ; ++k) request(k, n, oh, 0);
    }
    dma_wait();
    lds_barrier();
    for (int r = r0; r < r1; ++r) {
        const int buf = (r - r0) & 1;
        if (r + 1 < r1) {
            const int nn = (r + 1) / s.OH, noh = (r + 1) - nn * s.OH;
            for (int k = 0; k < NREQ; ++k) request(k, nn, noh, buf ^ 1);
        }
        unsigned va = a_lane + (unsigned)(buf * DYS * 4);
        unsigned vb[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) vb[b] = b_lane[b] + b_step[b] * (unsigned)(buf * XS * 4);
        // (a hand-made two-stage pipeline of these reads against the MFMAs measured slower: 0.324 against 0.307 ms)
        for (int w = 0; w < nwin; ++w) {
            const f32x4_t av = *reinterpret_cast<const f32x4_t*>(ldsb + va);
            float bv[5][4];
#pragma unroll
            for (int b = 0; b < 5; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[b][e] = *reinterpret_cast<const float*>(ldsb + vb[b] + 8 * e);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[b][e], acc[b], 0, 0, 0);
            va += 64u;
#pragma unroll
            for (int b = 0; b < 5; ++b) vb[b] += b_step[b] * 128u;
        }
        dma_wait();
        lds_barrier();
    }
    // D: column = lane & 15 (tap of the block), row = 4 * (lane >> 4) + register (filter of the block)
    float* out = a.partials + ((size_t)blockIdx.x * s.groups + g) * 64 * kStemTapCols;
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            out[(fb * 16 + 4 * kq + rr) * kStemTapCols + (th * 5 + b) * 16 + l15] = acc[b][rr];
}

// dW[g][f][k] += sum_p partials[p][g][f][k] (k < K), dbias[g*Mg + f] += column K; fixed summation order
__global__ __launch_bounds__(256) void conv_dw_stem_finalize_kernel(const float* __restrict__ partials, int nparts, int groups,
                                                                   int Mg, int K, int bias_col, float* __restrict__ dw,
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
        for (int p = pl; p < nparts; p += 16) sum += partials[(((size_t)p * groups + g) * 64 + f) * kStemTapCols + k];
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

static bool dw_stem_ok(const ConvShape& s) {
    return stem_ok(s) && (s.OW % 4) == 0 && ((s.OW + 15) & ~15) <= 120 && 2 * (((s.OW + 15) & ~15) - 1) + 7 + (kWinOrg - s.pad) <= 232;
}

static void dw_stem_plan(const ConvShape& s, int* rpb, int* blocks) {
    const int total = s.N * s.OH;
    int b = kCUs;  // one persistent workgroup of 8 waves per CU (98 KB of LDS)
    if (b > total) b = total;
    *rpb = ceil_div(total, b);
    *blocks = ceil_div(total, *rpb);
}

size_t conv_dw_stem_workspace_floats(const ConvShape& s) {
    if (!dw_stem_ok(s)) return 0;
    int rpb, blocks;
    dw_stem_plan(s, &rpb, &blocks);
    return (size_t)blocks * s.groups * 64 * kStemTapCols;
}

// false: shape not covered. true: dW accumulated, and the bias gradient too when dbias != NULL.
bool conv_backward_weights_stem(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s, float* workspace,
                                size_t workspace_floats) {
    if (!dw_stem_ok(s)) return false;
    ConvStemDwArgs a;
    int blocks;
    dw_stem_plan(s, &a.rows_per_block, &blocks);
    const size_t need = (size_t)blocks * s.groups * 64 * kStemTapCols;
    if (workspace == nullptr || workspace_floats < need) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n", workspace_floats,
                need);
        exit(1);
    }
    KTimer kt(K_CONV_DW, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    a.x = x; a.dy = dy; a.partials = workspace; a.s = s;
    a.total_rows = s.N * s.OH;
    a.bias_col = dbias ? 1 : 0;
    const dim3 grid((unsigned)blocks, (unsigned)s.groups);
    trace_kernel("conv_dw_stem_kernel");
    if (s.Cg == 1) conv_dw_stem_kernel<1, 120, 232><<<grid, 512, 0, current_stream()>>>(a);
    else if (s.Cg == 2) conv_dw_stem_kernel<2, 120, 232><<<grid, 512, 0, current_stream()>>>(a);
    else conv_dw_stem_kernel<3, 120, 232><<<grid, 512, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    const int total = s.groups * s.Mg * (s.K + a.bias_col);
    conv_dw_stem_finalize_kernel<<<ceil_div(total, 16), 256, 0, current_stream()>>>(workspace, blocks, s.groups, s.Mg, s.K,
                                                                                    a.bias_col, dw, dbias);
    KERNEL_CHECK();
    return true;
}

// conv_direct.hip
void conv_dw_direct_finalize(const float* partials, int nparts, int groups, int Mg, int K, int MP, int bias_col, float* dw,
                             float* dbias);

static bool dw_window_ok(const ConvShape& s) { return window_ok(s) && (s.OW % 8) == 0; }

static void dw_rows_plan(const ConvShape& s, int* rpb, int* blocks) {
    const int total = s.N * s.OH;
    const int fblocks = (s.Mg + 31) / 32;
#if ROWS_TM == 2
    int b = kCUs;  // one persistent workgroup of 8 waves per CU: both filter blocks, 133 KB of LDS
    (void)fblocks;
#else
    // persistent workgroups, one per block of 32 filters: two per CU at the narrow pitch (75 KB of LDS each), one at the wide
    int b = ((window_pitch(s) == 232 && ROWS_NS == 2 ? 2 : 1) * kCUs) / fblocks;
#endif
    if (b < 1) b = 1;
    if (b > total) b = total;
    *rpb = ceil_div(total, b);
    *blocks = ceil_div(total, *rpb);
}

size_t conv_dw_window_workspace_floats(const ConvShape& s) {
    if (!dw_window_ok(s)) return 0;
    int rpb, blocks;
    dw_rows_plan(s, &rpb, &blocks);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    return (size_t)blocks * s.groups * tm * 32 * 32;
}

// false: shape not covered. true: dW accumulated, and the bias gradient too when dbias != NULL.
bool conv_backward_weights_window(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s,
                                  float* workspace, size_t workspace_floats) {
    if (!dw_window_ok(s)) return false;
    ConvRowsDwArgs ra;
    int rblocks;
    dw_rows_plan(s, &ra.rows_per_block, &rblocks);
    const int tm = (s.Mg <= 32) ? 1 : 2;
    const size_t need = (size_t)rblocks * s.groups * tm * 32 * 32;
    if (workspace == nullptr || workspace_floats < need) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n", workspace_floats,
                need);
        exit(1);
    }
    KTimer kt(K_CONV_DW, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    ra.x = x; ra.dy = dy; ra.partials = workspace; ra.s = s;
    ra.total_rows = s.N * s.OH;
    ra.bias_col = dbias ? 1 : 0;
    ra.mp = tm * 32;
    const int pitch = window_pitch(s);  // dy pitch = 4 x odd floats >= OW
    constexpr int RTM = ROWS_TM;        // blocks of 32 filters per workgroup
    const int wtm = tm < RTM ? tm : RTM;
    const dim3 rgrid((unsigned)rblocks, (unsigned)s.groups, (unsigned)(tm / wtm));
#define RLAUNCH2(CGv, PDv, PXv) do { if (wtm == 1) conv_dw_rows_kernel<CGv, PDv, PXv, 1><<<rgrid, 256, 0, current_stream()>>>(ra); \
                                     else conv_dw_rows_kernel<CGv, PDv, PXv, RTM><<<rgrid, 256 * RTM, 0, current_stream()>>>(ra); } while (0)
#ifndef ROWS_PX
#define ROWS_PX 232
#endif
#define RLAUNCH1(CGv) do { if (pitch == 232) RLAUNCH2(CGv, 228, ROWS_PX); else RLAUNCH2(CGv, 260, 264); } while (0)
    trace_kernel("conv_dw_rows_kernel");
    if (s.Cg == 1) RLAUNCH1(1);
    else if (s.Cg == 2) RLAUNCH1(2);
    else RLAUNCH1(3);
#undef RLAUNCH1
#undef RLAUNCH2
    KERNEL_CHECK();
    conv_dw_direct_finalize(workspace, rblocks, s.groups, s.Mg, s.K, tm * 32, ra.bias_col, dw, dbias);
    return true;
}

}  // namespace bcnn_hip
