// conv_winograd43_dw.hip -- weight gradient of a 3x3 / s1 / p1 convolution on planes of whole 4 x 4 tiles with Winograd
// F(4x4, 3x3) in its transposed form (round 6):
//
//     dU_xi[f][c] = sum over tiles t of  dM_xi[f][t] . V_xi[c][t],   dM = A dy_t A^T (4 x 4 -> 6 x 6),  V = B^T d_t B (6 x 6),
//     dw[f][c] += G^T dU[f][c] G                                       (36 instead of 64 multiplies per 4 x 4 outputs, fp32)
//
// on the skeleton of conv_winograd43b.hip (wino43b_mma.h): v_mfma_f32_16x16x4_f32 with m = output channel, n = input channel,
// k = tile; a workgroup owns 64 output x 32 input channels x all 36 positions (144 accumulator registers per lane, 8 waves)
// and a range of tiles, walked in periods of 8 tiles = two sub-chunks of 4 (one MFMA step per position):
//   A stage [sub-chunk 2][xi/4 9][k 4][f 64][xi%4]   2 x 36,864 B   dM, written by the waves (no packed weights here)
//   B stage [period parity 2][sub-chunk 2][xi/4 9][k 4][c 32][xi%4]   2 x 36,864 B   V
// Both operands are transformed in the loop: per period 256 patches (8 tiles x 32 channels) and 512 dy tiles (8 x 64). The two
// groups of four waves alternate by period: T transforms the NEXT period's V in the first sub-chunk and the next period's first
// dy window in the second one, R this period's second dy window in the first sub-chunk. Every wave issues the same twelve
// 16-byte / 4-byte requests per period at the same places of the loop (one definition per register and iteration,
// DESIGN.md section 4.0); which tensor and tile they address depends on the role: a dy tile's rows ARE rows 1..4 of a patch
// of the other tensor, so the role only picks descriptor, channel count and tile. The requests sit BETWEEN the fragment groups
// of the two MFMA phases (wb_mma's `between`): a request instruction holds its wave while the texture path is busy -- a
// patch row touches 8-16 lines, an edge duty 48 --, and there the SIMD's other wave multiplies meanwhile; behind the
// transforms, in front of the barrier, the same stalls had everybody waiting (0.156 -> 0.139 ms at 56 x 56).
// A lane's transform item: patch = (tile l % 8 of the period, channel l / 8 of the wave's eight): the neighbour columns come
// from the neighbouring lanes (DPP wave shifts), the ends of an 8-lane group from edge duties of lanes 1..6 as in
// conv_winograd43b.hip; dy = (tile l % 4 of the window, output channel l / 4 of the wave's sixteen). The eight lanes a
// ds_write_b128 is serviced in are then eight tiles of ONE channel -- eight rows of the stage, the same banks: the rows are
// rotated by 2 (tile / 2) channels, which leaves the 16-byte writes 2-way conflicting (their issue time hides that) and the
// fragment reads conflict-free (a ds_read_b128 group holds rows k and k + 1 of a pair: they must share their rotation).
// The partial dU of a workgroup leaves as G^T dU G (9 instead of 36 values per channel pair); wino43_dw_finalize_kernel adds
// the splits in a fixed order onto dw (beta = 1: the momentum carry of the reference, bcnn_conv_layer.c:533-560).
#include "conv_common.h"
#include "lds_dma.h"
#include "wino43_math.h"
#include "wino43b_mma.h"

namespace bcnn_hip {

constexpr int WD4_BC = 32;  // input channels per workgroup (the MFMA's n); output channels: WB_BF = 64
constexpr int WD4_KT = 8;   // tiles per period

struct Wino43DwArgs {
    const float* x;    // [N][C][H][W]
    const float* dy;   // [N][F][H][W]
    float* partials;   // [split][block][f 64][c 32][9]
    int N, C, F, H, W, TH, TW;
    unsigned T;        // tiles
    int fblocks, cblocks, splits;
    unsigned tiles_per_split;  // multiple of 8
    unsigned x_bytes, dy_bytes;
    unsigned magic_img, magic_tw;  // ceil(2^32 / (TH TW)), ceil(2^32 / TW): magic_div
};

__global__ __launch_bounds__(64 * WB_NW, 2) void wino43_dw_kernel(const Wino43DwArgs a) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * WB_USTAGE + 2 * WB_VSTAGE];  // 147,456 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, w4 = wid & 3;     // role group; transform role: channels 8 w4 .. / 16 w4 ..
    const int cb = wid & 3, tbw = wid >> 2;     // MFMA role: output channels 16 cb .., input channels 16 tbw ..
    const int l15 = lane & 15, lq = lane >> 4;
    const int l7 = lane & 7, lj = lane >> 3;    // patch item
    const int k4 = lane & 3, lf = lane >> 2;    // dy item
    const int nob = a.fblocks * a.cblocks;
    const int sp = (int)blockIdx.x / nob, ob = (int)blockIdx.x - sp * nob;
    const int fb = ob % a.fblocks, cbk = ob / a.fblocks;
    const unsigned t_begin = (unsigned)sp * a.tiles_per_split;
    const unsigned t_left = a.T - t_begin;
    const int np = (int)((t_left < a.tiles_per_split ? t_left : a.tiles_per_split) + WD4_KT - 1) / WD4_KT;
    const int HW = a.H * a.W;
    const unsigned row_bytes = (unsigned)a.W * 4u;
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const rsrc_i4 rs_x = make_rsrc(a.x, a.x_bytes), rs_dy = make_rsrc(a.dy, a.dy_bytes);
    const int c_own = cbk * WD4_BC + 8 * w4 + lj, f_own = fb * WB_BF + 16 * w4 + lf;

    // (every factor below is < 2^24 -- the plan checks it --: v_mul_u32_u24 is a full-rate instruction, v_mul_lo_u32 is not)
    auto coords = [&](unsigned t, unsigned& n, int& th, int& tw) {
        n = __umulhi(t, a.magic_img);
        const unsigned rr = t - __umul24(n, per_img);
        th = (int)__umulhi(rr, a.magic_tw);
        tw = (int)(rr - __umul24((unsigned)th, (unsigned)a.TW));
    };

    // ---- the requests: twelve instructions, the same for both kinds of item ----
    // A[0..5]: rows -1 .. 4 of the tile's 4 x 4 block, own four columns (a dy item: rows 0..3 = A[1..4], the others blank);
    // e0 / e1: this lane's edge duties (patch items, lanes 1..6 of an 8-lane group: row l7 - 1 of the first tile's left /
    // the last tile's right column); pf: 1 left padding, 2 right padding (patch items). B[0..3]: a dy item.
    buf_f32x4 A[6], B[4];
    float e0, e1;
    unsigned pf;
    unsigned qa_mid, qa_top, qa_bot, qa_l, qa_r, qb_mid;   // the voffsets of the next requests (prep_a / prep_b)
    rsrc_i4 qa_rs;
    auto prep_a = [&](bool is_dy, unsigned t_period) {
        const unsigned t = t_period + (unsigned)(is_dy ? k4 : l7);
        const int chans = is_dy ? a.F : a.C, ch = is_dy ? f_own : c_own;
        qa_rs = is_dy ? rs_dy : rs_x;
        unsigned n; int th, tw;
        coords(t, n, th, tw);
        const bool ok = t < a.T && ch < chans;
        const unsigned chan_off = (unsigned)ch * (unsigned)HW * 4u;
        const unsigned img = (unsigned)chans * (unsigned)HW * 4u;
        qa_mid = ok ? __umul24(n, img) + chan_off + (__umul24((unsigned)th, row_bytes) + (unsigned)tw * 4u) * 4u : kOOB;
        qa_top = (ok && !is_dy && th > 0) ? qa_mid - row_bytes : kOOB;
        qa_bot = (!is_dy && 4 * th + 4 < a.H) ? qa_mid : kOOB;
        pf = (tw == 0 ? 1u : 0u) | (tw + 1 == a.TW ? 2u : 0u);
        // edge duties: the 8-lane group's first and last tile (patch items only: a uniform branch around arithmetic, no request in it)
        qa_l = kOOB; qa_r = kOOB;
        if (!is_dy) {
            const bool duty = l7 >= 1 && l7 <= 6 && ch < chans;
            const int row = l7 - 1;
            // (the group's tiles are consecutive and a tile row has at least seven: at most one row / image boundary either way)
            int twl = tw - l7, twr = tw + 7 - l7;
            const int wl = twl < 0 ? 1 : 0, wr = twr >= a.TW ? 1 : 0;
            twl += wl ? a.TW : 0; twr -= wr ? a.TW : 0;
            int thl = th - wl, thr = th + wr;
            const int wl2 = thl < 0 ? 1 : 0, wr2 = thr >= a.TH ? 1 : 0;
            thl += wl2 ? a.TH : 0; thr -= wr2 ? a.TH : 0;
            const unsigned nl = n - (unsigned)wl2, nr = n + (unsigned)wr2;
            const int ihl = 4 * thl - 1 + row, ihr = 4 * thr - 1 + row;
            const bool okl = duty && t - (unsigned)l7 < a.T && twl > 0 && ihl >= 0 && ihl < a.H;
            const bool okr = duty && t - (unsigned)l7 + 7u < a.T && twr + 1 < a.TW && ihr >= 0 && ihr < a.H;
            if (okl) qa_l = __umul24(nl, img) + chan_off + (__umul24((unsigned)ihl, (unsigned)a.W) + (unsigned)(4 * twl - 1)) * 4u;
            if (okr) qa_r = __umul24(nr, img) + chan_off + (__umul24((unsigned)ihr, (unsigned)a.W) + (unsigned)(4 * twr + 4)) * 4u;
        }
    };
    // part 1: rows -1, 0, 1; part 2: rows 2, 3, 4; part 3: the edge duties
    auto issue_a = [&](int part) {
        if (part == 1 || part == 2) {
#pragma unroll
            for (int i = 3 * (part - 1); i < 3 * part; ++i) {
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(i == 0 ? 0 : i - 1) * row_bytes));
                A[i] = buffer_load_f32x4(qa_rs, (int)(i == 0 ? qa_top : i == 5 ? qa_bot : qa_mid), (int)so, 0);
            }
        } else {
            e0 = buffer_load_f32(qa_rs, (int)qa_l, 0, 0);
            e1 = buffer_load_f32(qa_rs, (int)qa_r, 0, 0);
        }
    };
    auto request_a = [&](bool is_dy, unsigned t_period) {
        prep_a(is_dy, t_period);
        issue_a(1); issue_a(2); issue_a(3);
    };
    auto prep_b = [&](bool live, unsigned t_window) {
        qb_mid = kOOB;
        if (live) {  // uniform
            const unsigned t = t_window + (unsigned)k4;
            unsigned n; int th, tw;
            coords(t, n, th, tw);
            if (t < a.T && f_own < a.F)
                qb_mid = __umul24(n, (unsigned)a.F * (unsigned)HW * 4u) + (unsigned)f_own * (unsigned)HW * 4u
                         + (__umul24((unsigned)th, row_bytes) + (unsigned)tw * 4u) * 4u;
        }
    };
    auto issue_b = [&](int part) {   // part 1: rows 0, 1; part 2: rows 2, 3
        if (part > 2) return;
#pragma unroll
        for (int r = 2 * (part - 1); r < 2 * part; ++r) {
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)r * row_bytes));
            B[r] = buffer_load_f32x4(rs_dy, (int)qb_mid, (int)so, 0);
        }
    };
    // ---- B^T d B of the patch in A / e0 / e1 -> V stage vs (see conv_winograd43b.hip: write_v; here 8-lane groups) ----
    auto write_v = [&](int vs) {
        const bool pad_l = (pf & 1u) != 0, pad_r = (pf & 2u) != 0;
        const int gb = lane & ~7;
        float tt[6][6];
        {
            float cl[6], cr[6], out[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int evl = __builtin_amdgcn_ds_bpermute(4 * (gb + 1 + i), __float_as_int(e0));
                const int evr = __builtin_amdgcn_ds_bpermute(4 * (gb + 1 + i), __float_as_int(e1));
                const float own_first = A[i][0], own_last = A[i][3];  // (through scalars: hipcc 7.2)
                const int l = __builtin_amdgcn_update_dpp(evl, __float_as_int(own_last), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                cl[i] = pad_l ? 0.f : __builtin_bit_cast(float, l7 == 0 ? evl : l);
                const int r = __builtin_amdgcn_update_dpp(evr, __float_as_int(own_first), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                cr[i] = pad_r ? 0.f : __builtin_bit_cast(float, l7 == 7 ? evr : r);
            }
            w43_bt(cl, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][0] = out[i];
            w43_bt(cr, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][5] = out[i];
        }
#pragma unroll
        for (int j = 1; j <= 4; ++j) {
            const float col[6] = {A[0][j - 1], A[1][j - 1], A[2][j - 1], A[3][j - 1], A[4][j - 1], A[5][j - 1]};
            float out[6];
            w43_bt(col, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][j] = out[i];
        }
        float* v = lds + 2 * WB_USTAGE + vs * WB_VSTAGE + (l7 >> 2) * WB_VC + ((l7 & 3) * 32 + ((8 * w4 + lj + 2 * (l7 >> 1)) & 31)) * 4;
#pragma unroll
        for (int ip = 0; ip < 3; ++ip) {
            float o0[6], o1[6];
            w43_bt(tt[2 * ip], o0);
            w43_bt(tt[2 * ip + 1], o1);
            *reinterpret_cast<f32x4*>(v + (3 * ip) * 512) = f32x4{o0[0], o0[1], o0[2], o0[3]};
            *reinterpret_cast<f32x4*>(v + (3 * ip + 1) * 512) = f32x4{o0[4], o0[5], o1[0], o1[1]};
            *reinterpret_cast<f32x4*>(v + (3 * ip + 2) * 512) = f32x4{o1[2], o1[3], o1[4], o1[5]};
        }
    };
    // ---- A dy A^T of the 4 x 4 block in r0..r3 -> A stage `stage` ----
    auto write_dm = [&](int stage, const buf_f32x4& r0, const buf_f32x4& r1, const buf_f32x4& r2, const buf_f32x4& r3) {
        float tt[6][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float col[4] = {r0[j], r1[j], r2[j], r3[j]};
            float out[6];
            w43_a(col, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][j] = out[i];
        }
        float* u = lds + stage * WB_USTAGE + (k4 * 64 + ((16 * w4 + lf + 2 * (k4 >> 1)) & 63)) * 4;
#pragma unroll
        for (int ip = 0; ip < 3; ++ip) {
            float o0[6], o1[6];
            w43_a(tt[2 * ip], o0);
            w43_a(tt[2 * ip + 1], o1);
            *reinterpret_cast<f32x4*>(u + (3 * ip) * 1024) = f32x4{o0[0], o0[1], o0[2], o0[3]};
            *reinterpret_cast<f32x4*>(u + (3 * ip + 1) * 1024) = f32x4{o0[4], o0[5], o1[0], o1[1]};
            *reinterpret_cast<f32x4*>(u + (3 * ip + 2) * 1024) = f32x4{o1[2], o1[3], o1[4], o1[5]};
        }
    };

    f32x4 acc[36];
    const float* const ufrag = lds + (lq * 64 + ((16 * cb + l15 + 2 * (lq >> 1)) & 63)) * 4;  // + sub * WB_USTAGE + (xi / 4) * 1024
    const float* const vfrag0 = lds + 2 * WB_USTAGE + (lq * 32 + ((16 * tbw + l15 + 2 * (lq >> 1)) & 31)) * 4;
    const float* const vfrag1 = lds + 2 * WB_USTAGE + WB_VC + (lq * 32 + ((16 * tbw + l15 + 2 * ((4 + lq) >> 1)) & 31)) * 4;

    // ---- prologue: V(0) and dM(0, first window) in place; the loop's requests of "period -1" in flight ----
    const bool t0 = grp == 0;  // T in the even periods
    request_a(!t0, t_begin);   // T of period 0: the patches of period 0; R: the dy tiles of window (0, 0)
    if (t0) write_v(0);
    else write_dm(0, A[1], A[2], A[3], A[4]);
    request_a(!t0, t_begin + (t0 ? 8u : 4u));  // T: patches of period 1; R: dy of window (0, 1)

    for (int p = 0; p < np; ++p) {
        const bool t_role = (grp == (p & 1));
        const int vs = p & 1;
        const unsigned t_p = t_begin + 8u * (unsigned)p;
        // ---- first sub-chunk: A stage 0, V[vs][0] ----
        lds_barrier();
        // the requests ride inside the MFMA phases: a wave that the texture path holds up stalls while its SIMD partner
        // multiplies. B: T's dy of window (p + 1, 0), transformed behind the second sub-chunk
        prep_b(t_role, t_p + 8u);
        __builtin_amdgcn_sched_barrier(0);
        if (p == 0) wb_mma<true>(acc, ufrag, vfrag0 + vs * WB_VSTAGE, issue_b);
        else wb_mma<false>(acc, ufrag, vfrag0 + vs * WB_VSTAGE, issue_b);
        __builtin_amdgcn_sched_barrier(0);
        if (t_role) write_v(vs ^ 1);                          // V(p + 1)
        else write_dm(1, A[1], A[2], A[3], A[4]);             // dM(p, second window)
        // ---- second sub-chunk: A stage 1, V[vs][1] ----
        lds_barrier();
        // A: R's patches of period p + 2 (it transforms them as T of period p + 1); T's dy of window (p + 1, 1)
        prep_a(t_role, t_p + (t_role ? 12u : 16u));
        __builtin_amdgcn_sched_barrier(0);
        wb_mma<false>(acc, ufrag + WB_USTAGE, vfrag1 + vs * WB_VSTAGE, issue_a);
        __builtin_amdgcn_sched_barrier(0);
        if (t_role) write_dm(0, B[0], B[1], B[2], B[3]);      // dM(p + 1, first window)
    }

    // ---- G^T dU G of the wave's 16 x 16 channel pairs; lane: output channels 16 cb + 4 lq + i, input channel 16 tbw + l15 ----
    float* const out = a.partials + ((size_t)blockIdx.x * WB_BF + 16 * cb + 4 * lq) * (WD4_BC * 9) + (16 * tbw + l15) * 9;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float tt[3][6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float m[6] = {acc[q][i], acc[6 + q][i], acc[12 + q][i], acc[18 + q][i], acc[24 + q][i], acc[30 + q][i]};
            float y[3];
            w43_gt(m, y);
#pragma unroll
            for (int r = 0; r < 3; ++r) tt[r][q] = y[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float y[3];
            w43_gt(tt[r], y);
#pragma unroll
            for (int b = 0; b < 3; ++b) out[i * (WD4_BC * 9) + r * 3 + b] = y[b];
        }
    }
}

// dw[f][c][3][3] += sum over the splits (in order, four chains like wino_dw_fused_finalize_kernel) of partial[sp][ob][f][c][.]
__global__ __launch_bounds__(256) void wino43_dw_finalize_kernel(const float* __restrict__ partials, int splits, int fblocks,
                                                                 int cblocks, int F, int C, float* __restrict__ dw) {
    const int nob = fblocks * cblocks;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (ob, fl, cl, k): k fastest
    const int per_ob = WB_BF * WD4_BC * 9;
    if (idx >= nob * per_ob) return;
    const int ob = idx / per_ob, rem = idx - ob * per_ob;
    const int fl = rem / (WD4_BC * 9), r2 = rem - fl * (WD4_BC * 9);
    const int cl = r2 / 9, k = r2 - cl * 9;
    const int f = (ob % fblocks) * WB_BF + fl, c = (ob / fblocks) * WD4_BC + cl;
    if (f >= F || c >= C) return;
    const float* p = partials + (size_t)ob * per_ob + rem;
    const size_t stride = (size_t)nob * per_ob;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sp = 0;
    for (; sp + 3 < splits; sp += 4) {
        s0 += p[(size_t)sp * stride];
        s1 += p[(size_t)(sp + 1) * stride];
        s2 += p[(size_t)(sp + 2) * stride];
        s3 += p[(size_t)(sp + 3) * stride];
    }
    for (; sp < splits; ++sp) s0 += p[(size_t)sp * stride];
    dw[((size_t)f * C + c) * 9 + k] += (s0 + s1) + (s2 + s3);
}

struct Wino43DwPlan {
    bool ok;
    int fblocks, cblocks, splits;
    unsigned T, tiles_per_split;
    size_t partial_floats;
};

static int g_w43dw_force = -1;  // experiment build: BCNN_HIP_WINOGRAD43_DW=0/1 overrides the rule
static Wino43DwPlan wino43_dw_plan(const ConvShape& s, int cus = kCUs) {
    Wino43DwPlan p;
    p.ok = false; p.partial_floats = 0;
    if (s.ksz != 3 || s.stride != 1 || s.pad != 1 || s.groups != 1) return p;
    if ((s.H & 3) || (s.W & 3) || s.W < 28) return p;  // whole tiles; an 8-lane group spans at most two tile rows (TW >= 7)
    if ((size_t)s.N * s.C * s.HW * 4 >= 0x7ffffff0ull || (size_t)s.N * s.F * s.HW * 4 >= 0x7ffffff0ull) return p;
    if (g_w43dw_force < 0) {
        const char* e = BCNN_EXP_ENV("BCNN_HIP_WINOGRAD43_DW");
        g_w43dw_force = e ? (e[0] == '0' ? 0 : 1) : 2;
    }
    if (g_w43dw_force == 0) return p;
    if (g_w43dw_force == 2 && (s.C < 64 || s.F < 64)) return p;
    const int TH = s.H / 4, TW = s.W / 4;
    const unsigned long long T = (unsigned long long)s.N * TH * TW;
    if (T * (unsigned long long)(TH * TW) >= 0xffffffffull) return p;  // the multiply-high divisions are exact below that
    if (T + 64 >= (1ull << 24) || (size_t)(s.C > s.F ? s.C : s.F) * s.HW * 4 >= (1u << 24)) return p;  // 24-bit multiplies
    p.T = (unsigned)T;
    p.fblocks = (s.F + WB_BF - 1) / WB_BF; p.cblocks = (s.C + WD4_BC - 1) / WD4_BC;
    const int nob = p.fblocks * p.cblocks;
    int splits = cus / nob;
    if (splits < 1) splits = 1;
    unsigned per = (p.T + (unsigned)splits - 1) / (unsigned)splits;
    per = (per + WD4_KT - 1) / WD4_KT * WD4_KT;
    if (per < 4 * WD4_KT) per = 4 * WD4_KT;
    p.tiles_per_split = per;
    p.splits = (int)((p.T + per - 1) / per);
    if (g_w43dw_force == 2 && p.splits * nob < cus / 2) return p;  // too few tiles to fill the chip
    p.partial_floats = (size_t)p.splits * nob * (WB_BF * WD4_BC * 9);
    p.ok = true;
    return p;
}

size_t conv_dw_winograd43_workspace_floats(const ConvShape& s) { return wino43_dw_plan(s).partial_floats; }

static unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

bool conv_backward_weights_winograd43(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                      size_t workspace_floats) {
    const bool yield_cus = conv_side_stream_deferred() && wino43_dw_plan(s).ok;  // (the workspace was sized for the full plan)
    const Wino43DwPlan p = wino43_dw_plan(s, yield_cus ? kCUs * 3 / 4 : kCUs);
    if (!p.ok) return false;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) return false;  // 16-byte rows
    if (workspace == nullptr || workspace_floats < p.partial_floats) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n", workspace_floats,
                p.partial_floats);
        exit(1);
    }
    const double tiles = (double)s.N * (s.H / 4) * (s.W / 4);
    KTimer kt(K_CONV_DW_WINO43, 2.0 * 36.0 * tiles * s.C * s.F, 4.0 * ((double)s.N * s.HW * (s.C + s.F) + 9.0 * s.C * s.F));
    Wino43DwArgs a;
    a.x = x; a.dy = dy; a.partials = workspace;
    a.N = s.N; a.C = s.C; a.F = s.F; a.H = s.H; a.W = s.W; a.TH = s.H / 4; a.TW = s.W / 4;
    a.T = p.T; a.tiles_per_split = p.tiles_per_split;
    a.fblocks = p.fblocks; a.cblocks = p.cblocks; a.splits = p.splits;
    a.x_bytes = (unsigned)((size_t)s.N * s.C * s.HW * 4);
    a.dy_bytes = (unsigned)((size_t)s.N * s.F * s.HW * 4);
    a.magic_img = magic_of((unsigned)(a.TH * a.TW));
    a.magic_tw = magic_of((unsigned)a.TW);
    const int nob = p.fblocks * p.cblocks;
    trace_kernel("wino43_dw_kernel");
    wino43_dw_kernel<<<(unsigned)(p.splits * nob), 64 * WB_NW, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    const int total = nob * WB_BF * WD4_BC * 9;
    wino43_dw_finalize_kernel<<<(unsigned)((total + 255) / 256), 256, 0, current_stream()>>>(workspace, p.splits, p.fblocks,
                                                                                            p.cblocks, s.F, s.C, dw);
    KERNEL_CHECK();
    return true;
}

}  // namespace bcnn_hip
