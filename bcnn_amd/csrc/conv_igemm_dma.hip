// conv_igemm_dma.hip -- implicit-GEMM convolution (forward and dX) whose K loop issues NO vector-ALU work:
// both operand tiles go global -> LDS with `buffer_load_dword ... lds` (LDS-DMA), addressed by a per-lane
// VGPR offset that is constant for a whole filter tap plus a wave-uniform SGPR offset per reduction row.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate and shares the SIMD's issue with ordinary VALU
// instructions (measured: MFMA-only time + VALU-only time == total time of the register-staged kernel in
// conv_igemm.hip). Every address add / select / ds_write in the K loop is therefore stolen from the MFMAs.
// Here a K-tile costs a wave 16 DMA loads + 32 ds_read_b32 + 32 MFMAs and a handful of SALU instructions.
//
// Reference semantics: bcnn_forward_conv_layer_cpu / bcnn_backward_conv_layer_cpu,
// src/layers/bcnn_conv_layer.c:438-481 and :556-581 (incl. the 1x1 raw-view addressing :445-446, :562-569).
//
//   D[m][col] = sum_{tap t} sum_{j} At[t][j][m] * B_t[j][col]
//   forward : m = f, j = c, col = (n, oh, ow);  B_t[j][col] = x[n][g*Cg + j][oh*s - p + kr][ow*s - p + kc]
//   dX      : m = c, j = f, col = (n, ih, iw) of one stride-parity class; B_t = dy[n][g*Mg + j][qa - kr/s][qb - kc/s]
// The reduction runs tap-major so one K-tile (16 rows) lives inside ONE tap: a lane's gather offset and its
// zero-padding validity are fixed for the tile (invalid -> an out-of-range buffer offset, which the buffer
// unit turns into 0.0 in LDS), and the row only moves the wave-uniform soffset by j * plane size.
// At is the weight tensor re-packed per call to [tap][j (padded to 16)][m (padded to 128)] with zero
// padding, so ragged M / J need no predicates (a zero A row cancels whatever finite B row was fetched).
#include <type_traits>

#include "conv_common.h"
#include "lds_dma.h"

// ---- compile-time tuning knobs (defaults = what is measured best; tools/exp/build_variant.sh DEFS=-D... builds
// an alternative library for A/B runs). The timing-only ablation switches of rounds 1-5 (ABL_NODMA, ABL_NOBAR, ABL_NOSTORE,
// ABL_SETPRIO, ABL_CLOCK) were taken out of the file at the end of round 6; `git log -S ABL_NODMA` has them.
#ifndef DMA_BK
#define DMA_BK 16            // reduction rows per K-tile; 32 halves the barriers at twice the LDS: +-0
#endif
#ifndef DMA_NSTAGE
#define DMA_NSTAGE 2         // LDS ring depth; 3 costs occupancy and is 3-6 % slower on the 14x14 / 7x7 layers
#endif
#ifndef DMA_A_X4
#define DMA_A_X4 1           // A^T rows four at a time with 16-byte-per-lane LDS-DMA (BM == 64 tiles)
#endif
#ifndef DMA_UNROLL_STAGES
#define DMA_UNROLL_STAGES 0  // K loop unrolled by the ring depth: fewer VALU, yet 1-5 % slower on the ResNet shapes
#endif
#ifndef ABL_LB
#define ABL_LB 1             // second argument of __launch_bounds__
#endif

namespace bcnn_hip {

constexpr int kDmaBK = DMA_BK;   // reduction rows per K-tile (one barrier per tile)
constexpr int kDmaMaxTaps = 49;
constexpr int kDmaMaxClasses = 4;


struct DmaClass {
    int ih0, iw0, Hc, Wc;  // dX: first row/col and extent of the stride-parity class (forward: 0,0,OH,OW)
    int ntaps, tap0;       // taps of the class, index of its first tap in the packed At
    int nkx, sgn;          // tap t = (i, j) = (t / nkx, t % nkx) shifts the gathered element by sgn * (i, j)
    unsigned cpi_magic, wc_magic;  // floor(2^32 / columns per image), floor(2^32 / Wc) (0: divisor 1), set by launch_dma
};

struct DmaArgs {
    const float* at;       // packed weights [groups][kk2][Jpad][Mpad]
    const float* b_base;   // gathered tensor (x / dy)
    float* out;
    const float* bias;
    const float* slopes;
    ConvShape s;
    int mode;              // 0 forward, 1 dX
    int act, add_bias;
    int M, J, Jpad, Mpad;  // rows, reduction majors per tap (and their padded sizes)
    int kk2;               // taps in total (all classes)
    unsigned at_bytes, b_bytes;
    int b_major_stride;    // elements between consecutive j in the gathered tensor
    int mtiles;
    int nclass;
    // forward only, optional: per-channel partial sums of the stored values for the batch-norm statistics,
    // stats[((g*Mg + m) * stats_splits + column tile) * 2 + {0: sum, 1: sum of squares}]
    float* stats;
    int stats_splits;
    // 1x1 dX only, optional (DxBnSums, conv_common.h): S1 = sum of the stored values, S2 = sum of stored value * (bs_y - mean)
    // per ROW (input channel), one pair per column tile like `stats`
    const float* bs_y;
    const float* bs_mean;
    float* bs_out;
    // rowmode (few input channels, e.g. the 7x7 RGB stem): the gathered tensor is a zero-padded copy of x and the
    // reduction runs over ALL (c, kr, kc) rows as ONE "tap": row j reads the lane's pixel at the wave-uniform offset
    // c * plane + kr * pitch + kc, (c, kr, kc) = (j / kk, (j % kk) / ks, j % ks) by magic multiplies on the scalar unit
    int rowmode, row_kk, row_ks, row_plane, row_pitch;
    unsigned row_kk_magic, row_ks_magic;
    DmaClass cls[kDmaMaxClasses];
};

// BS: the variant that can emit the batch-norm backward sums of DxBnSums (its epilogue needs ~35 more registers, which would
// cost every other launch three of its eight waves per SIMD)
template <int WM, int WN, int TM, int TN, bool BS = false>
__global__ __launch_bounds__(64 * WM * WN, ABL_LB) void conv_igemm_dma_kernel(const DmaArgs a) {
    constexpr int BK = kDmaBK;
    constexpr int NW = WM * WN;         // waves per workgroup (4, or 2 for the 64 x 32 tile)
    constexpr int RPW = BK / NW;        // K rows each wave stages per tile
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // BN == 32: a B row is only half a DMA slab, so one DMA instruction fills TWO consecutive K rows (lanes
    // 0-31 row j, lanes 32-63 row j + 1: the second row's plane offset rides in the lane offset)
    constexpr bool HALF = (BN == 32);
    constexpr int AH = BM / 64, BH = HALF ? 1 : BN / 64;
    static_assert((NW == 4 || NW == 2) && BM % 64 == 0 && (BN % 64 == 0 || BN == 32), "tile");
    // one block of LDS: As[NS][BK][BM] | Bs[NS][BK][BN]; after the K loop the same bytes hold one 32x32
    // transposition pad per wave for the fused statistics
    // LDS ring depth: the DMA runs NS - 1 tiles ahead of the MFMAs. 3 stages measured no faster than 2 on any
    // ResNet shape and 3-6 % slower on the 14x14 / 7x7 layers (24 KB instead of 16 KB of LDS: 6 instead of 8
    // resident workgroups per CU) -- with 8 waves per SIMD the DMA latency is already hidden.
    constexpr int NS = DMA_NSTAGE;
    constexpr int SMEM = NS * BK * (BM + BN);
    static_assert(SMEM >= NW * 1024, "statistics pads");
    __shared__ float smem[SMEM];
    float (*As)[BK][BM] = reinterpret_cast<float (*)[BK][BM]>(smem);
    float (*Bs)[BK][BN] = reinterpret_cast<float (*)[BK][BN]>(smem + NS * BK * BM);

    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int g = blockIdx.y;
    const DmaClass& ci = a.cls[blockIdx.z];
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lb % a.mtiles, pt = lb / a.mtiles;
    const int m0 = mt * BM;
    const bool fwd = (a.mode == 0);

    // epilogue constants of this block's rows, staged once (a global load per stored value would serialise
    // the epilogue on L2 latency). bcnn_add_scalar of the AVX build skips exactly 0 and 1: 1 is stored as 0.
    __shared__ float s_bias[BM], s_slope[BM];
    if (fwd && tid < BM) {
        const int m = m0 + tid;
        float b = 0.f, sl = 0.f;
        if (m < a.M) {
            if (a.add_bias) { b = a.bias[g * s.Mg + m]; if (b == 1.0f) b = 0.f; }
            if (a.act == BCNN_HIP_ACT_PRELU) sl = a.slopes[g * s.Mg + m];
        }
        s_bias[tid] = b; s_slope[tid] = sl;
    }

    const int c_Wc = ci.Wc;
    const int col_per_img = s.pointwise ? s.OHOW : ci.Hc * ci.Wc;
    const int total_cols = s.N * col_per_img;
    const int p0 = pt * BN;
    if (p0 >= total_cols) return;  // class smaller than the grid (uniform per block)

    const int lim_y = s.pointwise ? 1 : (fwd ? s.H : s.OH);
    const int lim_x = s.pointwise ? 1 : (fwd ? s.W : s.OW);
    const int row_stride = fwd ? s.W : s.OW;

    // n / d for any 32-bit n with m = floor(2^32 / d): the multiply-high is at most one short, one compare puts it right
    // (five vector instructions where the compiler's 32-bit division takes ~35: every thread decodes two columns per tile,
    // and on a SIMD those instructions are taken from the MFMAs)
    auto div_exact = [](unsigned n, unsigned d, unsigned m) -> unsigned {
        if (m == 0u) return n;  // d == 1
        unsigned q = __umulhi(n, m);
        q += (n - q * d >= d) ? 1u : 0u;
        return q;
    };
    // decode one column -> element offset of tap (0,0) in the gathered tensor, its coordinates there, output offset
    auto decode = [&](int col, unsigned& pbase, int& cy, int& cx, unsigned& obase) -> bool {
        if (col >= total_cols) { pbase = 0; cy = -(1 << 20); cx = -(1 << 20); obase = 0; return false; }
        const unsigned n = div_exact((unsigned)col, (unsigned)col_per_img, ci.cpi_magic);
        const unsigned pix = (unsigned)col - n * (unsigned)col_per_img;
        const unsigned in_img = n * (unsigned)s.C + (unsigned)(g * s.Cg), out_img = n * (unsigned)s.F + (unsigned)(g * s.Mg);
        if (s.pointwise) {  // raw [Cg][OH*OW] / [Mg][OH*OW] views on both sides
            cy = 0; cx = 0;
            pbase = (fwd ? in_img * (unsigned)s.HW : out_img * (unsigned)s.OHOW) + pix;
            obase = (fwd ? out_img * (unsigned)s.OHOW : in_img * (unsigned)s.HW) + pix;
            return true;
        }
        const unsigned u = div_exact(pix, (unsigned)c_Wc, ci.wc_magic), v = pix - u * (unsigned)c_Wc;
        if (fwd) {
            cy = (int)u * s.stride - s.pad; cx = (int)v * s.stride - s.pad;
            pbase = (in_img * (unsigned)s.H + (unsigned)cy) * (unsigned)s.W + (unsigned)cx;
            obase = out_img * (unsigned)s.OHOW + pix;
        } else {
            const int ih = ci.ih0 + (int)u * s.stride, iw = ci.iw0 + (int)v * s.stride;
            cy = (ih + s.pad) / s.stride; cx = (iw + s.pad) / s.stride;  // exact for this class's taps
            pbase = (out_img * (unsigned)s.OH + (unsigned)cy) * (unsigned)s.OW + (unsigned)cx;
            obase = (in_img * (unsigned)s.H + (unsigned)ih) * (unsigned)s.W + (unsigned)iw;
        }
        return true;
    };

    // ---- staging columns of this lane: one per 64-column slab of the B tile ---------------------------
    unsigned pb[BH];
    int cy[BH], cx[BH];
#pragma unroll
    for (int h = 0; h < BH; ++h) {
        unsigned ob;
        decode(p0 + (HALF ? (lane & 31) : h * 64 + lane), pb[h], cy[h], cx[h], ob);
        if (HALF) pb[h] += (unsigned)(lane >> 5) * (unsigned)a.b_major_stride;  // upper half: the next K row
    }
    unsigned voff[BH];
    const int c_nkx = ci.nkx, c_sgn = ci.sgn;
    int tap_i = 0, tap_j = 0;  // wave-uniform tap counters (SALU)
    auto set_tap = [&]() {
        const int sdy = c_sgn * tap_i, sdx = c_sgn * tap_j;
        const int shift = sdy * row_stride + sdx;
#pragma unroll
        for (int h = 0; h < BH; ++h) {
            const bool ok = (unsigned)(cy[h] + sdy) < (unsigned)lim_y && (unsigned)(cx[h] + sdx) < (unsigned)lim_x;
            voff[h] = ok ? (pb[h] + (unsigned)shift) * 4u : kOOB;
        }
    };

    const rsrc_i4 rs_b = make_rsrc(a.b_base, a.b_bytes), rs_a = make_rsrc(a.at, a.at_bytes);
    const unsigned lds_a0 = lds_offset(&As[0][0][0]), lds_b0 = lds_offset(&Bs[0][0][0]);
    unsigned a_voff[AH];
#pragma unroll
    for (int h = 0; h < AH; ++h) a_voff[h] = (unsigned)(h * 64 + lane) * 4u;
    // BM == 64: four consecutive A^T rows are one 16-byte-per-lane DMA (lane -> row lane / 16, columns 4 * (lane % 16) ..)
    constexpr bool AX4 = DMA_A_X4 && (AH == 1) && (RPW % 4 == 0);
    const unsigned a_voff4 = ((unsigned)(lane >> 4) * (unsigned)a.Mpad + (unsigned)(lane & 15) * 4u) * 4u;
    const int JB = a.Jpad / BK;
    const int ntiles = ci.ntaps * JB;
    const unsigned a_tile0 = ((unsigned)(g * a.kk2 + ci.tap0) * (unsigned)a.Jpad) * (unsigned)a.Mpad + (unsigned)m0;

    // wave `wid` stages rows RPW*wid .. RPW*wid + RPW - 1 of both tiles
    auto stage = [&](int t, int jb, int buf) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = RPW * wid + r;
            const int j = jb * BK + row;
            const unsigned sa = (a_tile0 + (unsigned)(t * a.Jpad + j) * (unsigned)a.Mpad) * 4u;
            if (!HALF || (r & 1) == 0) {  // HALF: rows (row, row + 1) travel together (J is even: launch precondition)
                const int jc = j < a.J ? j : a.J - (HALF ? 2 : 1);  // padded rows meet a zero A row: any legal row(s)
                unsigned sb;
                if (a.rowmode) {
                    const unsigned c = magic_div((unsigned)jc, a.row_kk_magic), r = (unsigned)jc - c * (unsigned)a.row_kk;
                    const unsigned kr = magic_div(r, a.row_ks_magic), kc = r - kr * (unsigned)a.row_ks;
                    sb = (c * (unsigned)a.row_plane + kr * (unsigned)a.row_pitch + kc) * 4u;
                } else {
                    sb = (unsigned)jc * (unsigned)a.b_major_stride * 4u;
                }
#pragma unroll
                for (int h = 0; h < BH; ++h)
                    dma_row(rs_b, lds_b0 + (unsigned)(((buf * BK + row) * BN + h * 64) * 4), voff[h], sb);
            }
            if (AX4) {
                if ((r & 3) == 0) dma_row_x4(rs_a, lds_a0 + (unsigned)(((buf * BK + row) * BM) * 4), a_voff4, sa);
            } else {
#pragma unroll
                for (int h = 0; h < AH; ++h)
                    dma_row(rs_a, lds_a0 + (unsigned)(((buf * BK + row) * BM + h * 64) * 4), a_voff[h], sa);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    // ---- K loop: ring of NS LDS stages. Tile it + NS - 1 is requested before the MFMAs of tile it; at the
    // end of the iteration only tile it + 1 has to have landed, i.e. the wave waits until at most the loads
    // of the NS - 2 younger tiles are still in flight (vmcnt is in-order) and then meets the barrier.
    constexpr int LOADS_PER_TILE = (AX4 ? RPW / 4 : RPW * AH) + (HALF ? RPW / 2 : RPW * BH);  // DMA instructions per wave and tile
    int t_next = 0, jb_next = 0, issued = 0;       // (tap, major block) of the next tile to request
    auto request_next = [&](int buf) {
        if (issued > 0 && ++jb_next == JB) {
            jb_next = 0; ++t_next;
            if (++tap_j == c_nkx) { tap_j = 0; ++tap_i; }
            set_tap();
        }
        stage(t_next, jb_next, buf);
        ++issued;
    };
    if (ntiles > 0) set_tap();  // a class may own no tap at all (stride > kernel size): it then stores zeros
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < ntiles) request_next(p);
    if (NS == 3 && ntiles > 1) dma_wait_n<LOADS_PER_TILE>(); else dma_wait();
    __syncthreads();
    // One K-tile: request tile it + NS - 1, multiply tile it, make tile it + 1 visible. CURT can carry the LDS
    // stage as a compile-time constant (loop unrolled by the ring depth: fragment reads become base register +
    // immediate offset, saving 5 VALU per tile) -- see DMA_UNROLL_STAGES below for why that is not the default.
    int cur = 0, fill = NS - 1;  // run-time stage being multiplied / refilled (generic ring depth only)
    auto tile_body = [&](int it, auto CURT) {
        constexpr int CUR = decltype(CURT)::value;
        const int c = CUR >= 0 ? CUR : cur;
        const int f = CUR >= 0 ? (CUR + NS - 1) % NS : fill;
        if (it + NS - 1 < ntiles) request_next(f);  // DMA in flight under the MFMAs
        // fragments of k-step ks+1 are fetched from LDS before the MFMAs of k-step ks are issued
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = As[c][lhi][(wm * TM + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bs[c][lhi][(wn * TN + j) * 32 + l31];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int fc = ks & 1, fn = fc ^ 1;
            if (ks + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[fn][i] = As[c][2 * ks + 2 + lhi][(wm * TM + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[fn][j] = Bs[c][2 * ks + 2 + lhi][(wn * TN + j) * 32 + l31];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (the scheduler sinks it otherwise)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[fc][i], bf[fc][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NS == 3 && it + 2 < ntiles) dma_wait_n<LOADS_PER_TILE>(); else dma_wait();
        __syncthreads();
        if (CUR < 0) {
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            fill = (fill + 1 == NS) ? 0 : fill + 1;
        }
    };
    if (NS == 2 && DMA_UNROLL_STAGES) {
        int it = 0;
        for (; it + 1 < ntiles; it += 2) {
            tile_body(it, std::integral_constant<int, 0>{});
            tile_body(it + 1, std::integral_constant<int, 1>{});
        }
        if (it < ntiles) tile_body(it, std::integral_constant<int, 0>{});
    } else {
        for (int it = 0; it < ntiles; ++it) tile_body(it, std::integral_constant<int, -1>{});
    }

    // ---- batch-norm statistics of this tile (fused: saves the separate read of the whole output) -----
    // Columns past the end hold exact zeros (their B columns were zero-filled), so no masking is needed.
    // Each wave transposes its 32x32 accumulator tile through a private XOR-swizzled LDS pad so that a lane
    // owns half a ROW (16 values): ~50 VALU per tile instead of the ~600 a shuffle butterfly over the
    // accumulator layout costs -- fp32 MFMAs and VALU share the SIMD, epilogue VALU is not free.
    __shared__ float s_stat[BM][WN * TN][2];  // shared by the two statistics epilogues below (never both in one launch)
    if (a.stats != nullptr) {
        float* pad = smem + wid * 1024;
        const int prow = lane & 31, phalf = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
                __syncthreads();  // K-loop reads (first tile) / previous tile's pad reads are done
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma_row(r, lane);
                    pad[row * 32 + (l31 ^ row)] = v[r];
                }
                __syncthreads();
                float sv = 0.f, sq = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float t = pad[prow * 32 + ((phalf * 16 + c) ^ prow)];
                    sv += t;
                    sq += t * t;
                }
                sv += __shfl_xor(sv, 32);
                sq += __shfl_xor(sq, 32);
                if (phalf == 0) {
                    const int row = (wm * TM + i) * 32 + prow;
                    s_stat[row][wn * TN + j][0] = sv;
                    s_stat[row][wn * TN + j][1] = sq;
                }
            }
        __syncthreads();
        if (tid < BM && m0 + tid < a.M) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < WN * TN; ++k) { s0 += s_stat[tid][k][0]; s1 += s_stat[tid][k][1]; }
            float* dst = a.stats + ((size_t)(g * s.Mg + m0 + tid) * a.stats_splits + pt) * 2;
            dst[0] = s0; dst[1] = s1;
        }
    }

    // ---- the same transposition for the batch-norm node in front of a 1x1 convolution (dX): S1 and S2 of its backward --
    if (BS && a.bs_out != nullptr) {
        float (*s_bs)[WN * TN][2] = s_stat;
        float* pad = smem + wid * 1024;
        const int prow = lane & 31, phalf = lane >> 5;
        const bool vec4 = (col_per_img & 3) == 0 && (reinterpret_cast<uintptr_t>(a.bs_y) & 15) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // the lane's 16 partners y - mean are requested first: they arrive while the tile is transposed
                const int row = m0 + (wm * TM + i) * 32 + prow;  // input channel of this lane's half row
                const bool rok = row < a.M;
                const float mu = rok ? a.bs_mean[row] : 0.f;
                const int q0 = p0 + (wn * TN + j) * 32 + phalf * 16;  // first of the lane's 16 columns
                unsigned n = (unsigned)q0 / (unsigned)col_per_img, pix = (unsigned)q0 - n * (unsigned)col_per_img;
                float yv[16];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    // columns past the end hold exact zeros in the tile: any finite partner will do
                    yv[c4 * 4] = yv[c4 * 4 + 1] = yv[c4 * 4 + 2] = yv[c4 * 4 + 3] = mu;
                    const int q = q0 + c4 * 4;
                    if (rok && q < total_cols) {
                        const float* yp = a.bs_y + ((size_t)n * (size_t)s.C + (size_t)row) * (size_t)s.HW + pix;
                        if (vec4) {  // four columns stay inside one image and are 16-byte aligned
                            const float4 t4 = *reinterpret_cast<const float4*>(yp);
                            yv[c4 * 4] = t4.x; yv[c4 * 4 + 1] = t4.y; yv[c4 * 4 + 2] = t4.z; yv[c4 * 4 + 3] = t4.w;
                        } else {
                            unsigned nn = n, pp = pix;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if (q + k < total_cols)
                                    yv[c4 * 4 + k] = a.bs_y[((size_t)nn * (size_t)s.C + (size_t)row) * (size_t)s.HW + pp];
                                if (++pp == (unsigned)col_per_img) { pp = 0; ++nn; }
                            }
                        }
                    }
                    pix += 4;
                    if (pix >= (unsigned)col_per_img) { pix -= (unsigned)col_per_img; ++n; }
                }
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int prw = mfma_row(r, lane);
                    pad[prw * 32 + (l31 ^ prw)] = v[r];
                }
                __syncthreads();
                float sv = 0.f, sq = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float t = pad[prow * 32 + ((phalf * 16 + c) ^ prow)];
                    sv += t;
                    sq += t * (yv[c] - mu);
                }
                sv += __shfl_xor(sv, 32);
                sq += __shfl_xor(sq, 32);
                if (phalf == 0) {
                    const int trow = (wm * TM + i) * 32 + prow;
                    s_bs[trow][wn * TN + j][0] = sv;
                    s_bs[trow][wn * TN + j][1] = sq;
                }
            }
        __syncthreads();
        if (tid < BM && m0 + tid < a.M) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < WN * TN; ++k) { s0 += s_bs[tid][k][0]; s1 += s_bs[tid][k][1]; }
            float* dst = a.bs_out + ((size_t)(m0 + tid) * a.stats_splits + pt) * 2;
            dst[0] = s0; dst[1] = s1;
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    // 32-bit element offsets against the output base (tensors are < 2 GiB here), the 16 rows of an
    // accumulator at compile-time multiples of the wave-uniform row stride, and no per-value predicate or
    // bias / activation code when the tile is full and the store is plain (dX, and the raw forward that
    // feeds a batch-norm): epilogue VALU competes with the other resident waves' MFMAs.
    const unsigned o_row_stride = fwd ? (unsigned)s.OHOW : (s.pointwise ? (unsigned)s.OHOW : (unsigned)s.HW);
    const bool plain = !fwd || (!a.add_bias && a.act == BCNN_HIP_ACT_NONE);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        unsigned pbx, ob;
        int y, x;
        if (!decode(p0 + (wn * TN + j) * 32 + l31, pbx, y, x, ob)) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
            const int row0 = (wm * TM + i) * 32 + 4 * lhi;          // row of accumulator register 0, tile-relative
            const bool full = m0 + (wm * TM + i) * 32 + 32 <= a.M;  // wave-uniform
            const unsigned off0 = ob + (unsigned)(m0 + row0) * o_row_stride;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                constexpr int kRowOf[16] = {0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19, 24, 25, 26, 27};
                const int mr = kRowOf[r];
                if (!full && m0 + row0 + mr >= a.M) continue;
                float o = v[r];
                if (!plain) {
                    const float b = s_bias[row0 + mr];
                    if (b != 0.0f) o += b;
                    if (a.act != BCNN_HIP_ACT_NONE) o = act_fwd_cheap(o, a.act, s_slope[row0 + mr]);
                }
#ifdef NT_STORES   // experiment: the result tensor as a streaming store (tools/micro/mall_probe.hip)
                __builtin_nontemporal_store(o, &a.out[off0 + (unsigned)mr * o_row_stride]);
#else
                a.out[off0 + (unsigned)mr * o_row_stride] = o;
#endif
            }
        }
    }
}

// ---- weight re-pack: At[g][tap][j][m] = W[g*Mg + f][c][kr][kc] with (m, j) = (f, c) forward, (c, f) dX ----
static_assert(kPackMaxTaps == kDmaMaxTaps, "IgemmPackJob::tapoff holds one entry per tap");

// block b of a job's (gx, gy, gz) grid: 64 m x 4 j of one (g, tap); m fastest so writes are coalesced
__device__ __forceinline__ void pack_weights_block(const IgemmPackJob& a, int bx, int by, int gt) {
    const int m = bx * 64 + (threadIdx.x & 63);
    const int j = by * 4 + (threadIdx.x >> 6);
    const int g = gt / a.kk2, t = gt - g * a.kk2;
    if (m >= a.Mpad || j >= a.Jpad) return;
    float v = 0.f;
    if (m < a.M && j < a.J) {
        const int f = a.mode == 0 ? m : j, c = a.mode == 0 ? j : m;
        v = a.w[((size_t)(g * a.Mg + f) * a.Cg + c) * a.kk2 + a.tapoff[t]];
        if (a.fold_var) v *= bnfold_a(a.fold_var, a.fold_scales, g * a.Cg + c);  // a batch-norm in front of the layer, folded in
    }
    a.at[((size_t)gt * a.Jpad + j) * a.Mpad + m] = v;
}

__global__ __launch_bounds__(256) void conv_pack_weights_kernel(const IgemmPackJob a) {
    pack_weights_block(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// the packs of many layers in one launch (bcnn_hip_conv_prepack): blockIdx.y = job, blockIdx.x = its linear block
__global__ __launch_bounds__(256) void conv_pack_weights_multi_kernel(const IgemmPackJob* __restrict__ jobs) {
    const IgemmPackJob& a = jobs[blockIdx.y];
    const int b = blockIdx.x;
    if (b >= a.gx * a.gy * a.gz) return;
    const int bx = b % a.gx, r = b / a.gx;
    pack_weights_block(a, bx, r % a.gy, r / a.gy);
}

// ---- host side ------------------------------------------------------------------------------------------
static int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct DmaScratch {
    float* p = nullptr;
    size_t cap = 0;
    int dev = -1;
};
// two blocks: [0] forward / data gradient (weight packs, the padded input of the few-channel forward), [1] the padded input of
// the few-channel WEIGHT gradient, which may run on a side stream next to another layer's data gradient
static thread_local DmaScratch g_dma_scratch[2];

static float* dma_scratch(size_t floats, int which = 0) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    DmaScratch& sc = g_dma_scratch[which];
    if (sc.p == nullptr || sc.cap < floats || sc.dev != dev) {
        if (sc.p && sc.dev == dev) HIP_CHECK(hipFree(sc.p));  // hipFree synchronises the device
        const size_t cap = floats < (1u << 20) ? (1u << 20) : floats + floats / 2;
        HIP_CHECK(hipMalloc((void**)&sc.p, cap * sizeof(float)));
        sc.cap = cap;
        sc.dev = dev;
    }
    return sc.p;
}

// Shapes the DMA kernel takes; everything else stays on conv_igemm.hip.
static bool dma_supported(const ConvShape& s, int M, int J, size_t b_elems) {
    if (s.ksz > 7 && !s.pointwise) return false;
    if (M <= 32 || J < 8) return false;                       // tiny GEMM-M / reduction: padding waste
    if (b_elems * 4 >= 0x7ffffff0ull) return false;           // buffer range / OOB marker
    const int kk2 = s.pointwise ? 1 : s.ksz * s.ksz;
    const size_t at = (size_t)s.groups * kk2 * round_up(J, kDmaBK) * round_up(M, 128) * 4;
    if (at >= 0x7ffffff0ull) return false;
    if ((long long)s.N * (s.OHOW > s.HW ? s.OHOW : s.HW) >= 0x7fffffffLL) return false;
    return true;
}

// Tile shapes. Measured on the ResNet-18 shapes (tools/exp/tile_sweep.sh, TFLOP/s forward / dX):
//                      128x128   64x256   128x64   64x128   64x64
//   64ch  56x56 3x3      53/56    87/92    56/57    95/97   95/98
//   128ch 28x28 3x3      82/85    82/85    91/93    92/95   98/100
//   256ch 14x14 3x3      89/91    87/90    90/91    89/91   99/100
//   512ch  7x7  3x3      69/71    65/66    84/86    85/87   88/90
// A 64x32 tile (two waves, B rows as half slabs: config 5) quantises better still on the 7x7 layers but is
// 5-15 % slower everywhere (85/88, 90/91, 89/90, 72/84): the A tile is then re-fetched for half as many columns.
// The 64x64 tile (four waves of one 32x32 accumulator) wins everywhere: it keeps 8+ waves per SIMD resident
// (16 accumulator registers, 16 KB of LDS) so DMA latency, barriers and epilogues of one workgroup hide under
// the MFMAs of the others, and N*OH*OW = 2^k * 49 quantises onto the 256 CUs far better in small tiles (a
// 128x128 grid lands on 3.06 / 1.53 workgroups per CU: a quarter of the chip idles in the last round).
static int pick_dma_tile(const DmaArgs& a, int max_cols) {
    static const char* forced = BCNN_EXP_ENV("BCNN_HIP_IGEMM_TILE");  // experiments: 0..4
    if (forced && a.bs_out != nullptr) return 4;  // batch-norm sums come from the 64 x 64 tile only
    if (forced && forced[0] >= '0' && forced[0] <= '5' && (forced[0] != '5' || (a.J & 1) == 0)) return forced[0] - '0';
    (void)max_cols;
    return 4;
}

template <int WM, int WN, int TM, int TN>
static void launch_dma_cfg(DmaArgs& a, int max_cols) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    a.mtiles = ceil_div(a.M, BM);
    a.stats_splits = ceil_div(max_cols, BN);
    dim3 grid((unsigned)(a.mtiles * a.stats_splits), (unsigned)a.s.groups, (unsigned)a.nclass);
    trace_kernel(a.bs_out ? "conv_igemm_dma_kernel:dx+bnsums" : a.mode ? "conv_igemm_dma_kernel:dx" : "conv_igemm_dma_kernel:fwd");
    if (a.bs_out != nullptr) {
        if (WM == 2 && WN == 2 && TM == 1 && TN == 1) {
            conv_igemm_dma_kernel<2, 2, 1, 1, true><<<grid, 256, 0, current_stream()>>>(a);
            KERNEL_CHECK();
            return;
        }
        fprintf(stderr, "[bcnn_hip] conv_igemm_dma: batch-norm sums are emitted by the 64 x 64 tile only\n");
        exit(1);
    }
    conv_igemm_dma_kernel<WM, WN, TM, TN><<<grid, 64 * WM * WN, 0, current_stream()>>>(a);
    KERNEL_CHECK();
}

static void launch_dma(DmaArgs& a, int max_cols) {
    for (int c = 0; c < a.nclass && c < kDmaMaxClasses; ++c) {  // the divisors of the kernel's column decode
        DmaClass& ci = a.cls[c];
        const unsigned cpi = (unsigned)(a.s.pointwise ? a.s.OHOW : ci.Hc * ci.Wc), wc = (unsigned)ci.Wc;
        ci.cpi_magic = cpi > 1u ? (unsigned)(0x100000000ULL / cpi) : 0u;
        ci.wc_magic = wc > 1u ? (unsigned)(0x100000000ULL / wc) : 0u;
    }
    switch (pick_dma_tile(a, max_cols)) {
        case 0: launch_dma_cfg<2, 2, 2, 2>(a, max_cols); break;  // 128 x 128
        case 1: launch_dma_cfg<1, 4, 2, 2>(a, max_cols); break;  //  64 x 256
        case 2: launch_dma_cfg<2, 2, 2, 1>(a, max_cols); break;  // 128 x  64
        case 3: launch_dma_cfg<1, 4, 2, 1>(a, max_cols); break;  //  64 x 128
        case 5: launch_dma_cfg<2, 1, 1, 1>(a, max_cols); break;  //  64 x  32 (two waves, half-slab B rows)
        default: launch_dma_cfg<2, 2, 1, 1>(a, max_cols); break; //  64 x  64
    }
}

static void fill_pack_job(IgemmPackJob& p, const float* w, float* at, const ConvShape& s, int mode, int M, int J, int Jpad,
                          int Mpad, int kk2, const unsigned char* tapoff) {
    p.w = w; p.at = at; p.Mg = s.Mg; p.Cg = s.Cg; p.kk2 = kk2; p.ksz = s.ksz;
    p.M = M; p.J = J; p.Jpad = Jpad; p.Mpad = Mpad; p.mode = mode; p.groups = s.groups;
    for (int t = 0; t < kPackMaxTaps; ++t) p.tapoff[t] = t < kk2 ? tapoff[t] : (unsigned char)0;
    p.gx = Mpad / 64; p.gy = ceil_div(Jpad, 4); p.gz = s.groups * kk2;
    p.fold_var = nullptr; p.fold_scales = nullptr;
}

static void pack_weights(const float* w, float* at, const ConvShape& s, int mode, int M, int J, int Jpad, int Mpad,
                         int kk2, const unsigned char* tapoff, const BnFold* fold = nullptr) {
    IgemmPackJob p;
    fill_pack_job(p, w, at, s, mode, M, J, Jpad, Mpad, kk2, tapoff);
    if (fold) { p.fold_var = fold->var; p.fold_scales = fold->scales; }
    dim3 grid((unsigned)p.gx, (unsigned)p.gy, (unsigned)p.gz);
    conv_pack_weights_kernel<<<grid, 256, 0, current_stream()>>>(p);
    KERNEL_CHECK();
}

// packed tap order of the data-gradient form: pointwise = the one tap; otherwise stride-parity classes, class-major
static int dx_tap_order(const ConvShape& s, unsigned char* tapoff) {
    if (s.pointwise) { tapoff[0] = 0; return 1; }
    const int st = s.stride;
    int n = 0;
    for (int ra = 0; ra < st; ++ra)
        for (int rb = 0; rb < st; ++rb)
            for (int kr = ra; kr < s.ksz; kr += st)
                for (int kc = rb; kc < s.ksz; kc += st) tapoff[n++] = (unsigned char)(kr * s.ksz + kc);
    return n;
}

// bcnn_hip_conv_prepack: the A^T this layer's forward (dx_mode 0) / data-gradient (1) kernel will ask prepack_take for
bool dma_pack_plan(const ConvShape& s, int dx_mode, IgemmPackJob* job, size_t* floats) {
    const int M = dx_mode ? s.Cg : s.Mg, J = dx_mode ? s.Mg : s.Cg;
    if (!dma_supported(s, M, J, dx_mode ? (size_t)s.N * s.F * s.OHOW : (size_t)s.N * s.C * s.HW)) return false;
    const int kk2 = s.pointwise ? 1 : s.ksz * s.ksz;
    const int Jpad = round_up(J, kDmaBK), Mpad = round_up(M, 128);
    unsigned char tapoff[kDmaMaxTaps];
    if (dx_mode) dx_tap_order(s, tapoff);
    else for (int t = 0; t < kk2; ++t) tapoff[t] = (unsigned char)t;
    fill_pack_job(*job, nullptr, nullptr, s, dx_mode, M, J, Jpad, Mpad, kk2, tapoff);
    *floats = (size_t)s.groups * kk2 * Jpad * Mpad;
    return true;
}

void dma_pack_launch(const IgemmPackJob* jobs_dev, int n, int max_blocks) {
    conv_pack_weights_multi_kernel<<<dim3((unsigned)max_blocks, (unsigned)n), 256, 0, current_stream()>>>(jobs_dev);
    KERNEL_CHECK();
}

// Returns false when the shape is not covered (caller falls back to the register-staged kernel).
// stats (optional, raw mode only): in  -> partials buffer with room for F * ceil(N*OH*OW / 64) * 2 floats
//                                   out -> splits = number of column tiles written per channel (0: none)
bool conv_forward_dma_supported(const ConvShape& s) { return dma_supported(s, s.Mg, s.Cg, (size_t)s.N * s.C * s.HW); }

// fold: the batch-norm in front of the layer whose per-channel factors go into the packed weights (BnFold)
bool conv_forward_dma(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                      const ConvShape& s, int act, int raw, ConvStats* stats, const BnFold* fold) {
    if (stats) stats->splits = 0;
    if (!dma_supported(s, s.Mg, s.Cg, (size_t)s.N * s.C * s.HW)) return false;
    const int kk2 = s.pointwise ? 1 : s.ksz * s.ksz;
    DmaArgs a;
    a.rowmode = 0;
    a.b_base = x; a.out = y; a.bias = bias; a.slopes = slopes; a.s = s;
    a.mode = 0; a.act = raw ? BCNN_HIP_ACT_NONE : act; a.add_bias = raw ? 0 : 1;
    a.M = s.Mg; a.J = s.Cg; a.Jpad = round_up(a.J, kDmaBK); a.Mpad = round_up(a.M, 128); a.kk2 = kk2;
    const size_t at_floats = (size_t)s.groups * kk2 * a.Jpad * a.Mpad;
    float* at = prepack_take(w, PREPACK_IGEMM, 0, at_floats);  // packed ahead by bcnn_hip_conv_prepack?
    const bool packed = at != nullptr && fold == nullptr;   // a scaled pack depends on this batch's statistics: made here
    if (!packed) at = dma_scratch(at_floats);
    a.at = at; a.at_bytes = (unsigned)(at_floats * 4);
    a.b_bytes = (unsigned)((size_t)s.N * s.C * s.HW * 4);
    a.b_major_stride = s.pointwise ? s.OHOW : s.HW;
    a.nclass = 1;
    DmaClass& ci = a.cls[0];
    ci.ih0 = 0; ci.iw0 = 0; ci.Hc = s.OH; ci.Wc = s.OW; ci.ntaps = kk2; ci.tap0 = 0;
    unsigned char tapoff[kDmaMaxTaps];
    ci.nkx = s.pointwise ? 1 : s.ksz; ci.sgn = 1;
    for (int t = 0; t < kk2; ++t) tapoff[t] = (unsigned char)t;
    if (!packed) pack_weights(w, at, s, 0, a.M, a.J, a.Jpad, a.Mpad, kk2, tapoff, fold);
    a.stats = (stats && raw) ? stats->partials : nullptr;
    a.bs_out = nullptr; a.bs_y = nullptr; a.bs_mean = nullptr;
    launch_dma(a, (int)s.total_q);
    if (a.stats) stats->splits = a.stats_splits;
    return true;
}

// ---- few input channels (the RGB stem): padded-plane GEMM over all (c, kr, kc) rows ---------------------------
// xp[n][c][ih + pad][iw + pad] = x[n][c][ih][iw], zero border: every tap of every output pixel is then a plain,
// always-valid offset from the pixel's base, so the whole reduction is one "tap" for the LDS-DMA kernel.
__global__ __launch_bounds__(256) void conv_pad_input_kernel(const float* __restrict__ x, float* __restrict__ xp, int H, int W,
                                                             int Hp, int Wp, int pad, unsigned total) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned iw = i % (unsigned)Wp, t = i / (unsigned)Wp;
        const unsigned ih = t % (unsigned)Hp, plane = t / (unsigned)Hp;
        const int sh = (int)ih - pad, sw = (int)iw - pad;
        xp[i] = ((unsigned)sh < (unsigned)H && (unsigned)sw < (unsigned)W) ? x[((size_t)plane * H + sh) * W + sw] : 0.f;
    }
}

static unsigned magic_u32(int d) { return d > 1 ? (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d) : 0u; }

bool conv_small_c_applicable(const ConvShape& s) {
    if (s.groups != 1 || s.pointwise || s.ksz > 7 || s.Cg >= 8 || s.Mg <= 32) return false;
    if (s.K < 64) return false;  // K <= 32 has the LDS-free kernels of conv_direct.hip
    const long long Hp = s.H + 2 * s.pad, Wp = s.W + 2 * s.pad;
    if ((long long)s.N * s.C * Hp * Wp * 4 >= 0x7ffffff0LL || (long long)s.N * s.F * s.OHOW * 4 >= 0x7ffffff0LL) return false;
    return true;
}

// the zero-padded copy, shared by the forward pass and the weight gradient of the step (same thread, same x)
float* conv_small_c_padded_input(const float* x, const ConvShape& s, size_t extra_floats, float** extra, int for_dw) {
    const int Hp = s.H + 2 * s.pad, Wp = s.W + 2 * s.pad;
    const size_t xp_floats = (size_t)s.N * s.C * Hp * Wp;
    float* base = dma_scratch(xp_floats + extra_floats + 64, for_dw ? 1 : 0);
    float* xp = base;
    if (extra) *extra = base + ((xp_floats + 63) & ~(size_t)63);
    conv_pad_input_kernel<<<stream_grid(xp_floats / 4 + 1, 256), 256, 0, current_stream()>>>(x, xp, s.H, s.W, Hp, Wp, s.pad,
                                                                                          (unsigned)xp_floats);
    KERNEL_CHECK();
    return xp;
}

bool conv_forward_small_c(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                          const ConvShape& s, int act, int raw, ConvStats* stats) {
    if (stats) stats->splits = 0;
    if (!conv_small_c_applicable(s)) return false;
    const int Hp = s.H + 2 * s.pad, Wp = s.W + 2 * s.pad;
    DmaArgs a;
    a.J = s.K; a.M = s.Mg; a.Jpad = round_up(a.J, kDmaBK); a.Mpad = round_up(a.M, 128); a.kk2 = 1;
    const size_t at_floats = (size_t)a.Jpad * a.Mpad;
    float* at = nullptr;
    float* xp = conv_small_c_padded_input(x, s, at_floats, &at, /*for_dw=*/0);
    // the problem as the kernel sees it: the padded tensor, no padding left, one tap
    const ConvShape sp = make_conv_shape(s.N, s.C, Hp, Wp, s.F, s.ksz, s.stride, 0, 1);
    a.s = sp;
    a.b_base = xp; a.out = y; a.bias = bias; a.slopes = slopes;
    a.mode = 0; a.act = raw ? BCNN_HIP_ACT_NONE : act; a.add_bias = raw ? 0 : 1;
    a.at = at; a.at_bytes = (unsigned)(at_floats * 4);
    a.b_bytes = (unsigned)((size_t)s.N * s.C * Hp * Wp * 4);
    a.b_major_stride = 0;
    a.rowmode = 1; a.row_kk = s.ksz * s.ksz; a.row_ks = s.ksz; a.row_plane = Hp * Wp; a.row_pitch = Wp;
    a.row_kk_magic = magic_u32(a.row_kk); a.row_ks_magic = magic_u32(a.row_ks);
    a.nclass = 1;
    DmaClass& ci = a.cls[0];
    ci.ih0 = 0; ci.iw0 = 0; ci.Hc = sp.OH; ci.Wc = sp.OW; ci.ntaps = 1; ci.tap0 = 0; ci.nkx = 1; ci.sgn = 1;
    // A^T[j][m] = W[m][j]: the weight tensor [F][C*k*k] read as a 1x1 filter bank over J = C*k*k "channels"
    ConvShape ws = s;
    ws.Cg = s.K; ws.Mg = s.Mg; ws.ksz = 1;
    unsigned char tapoff[kDmaMaxTaps];
    tapoff[0] = 0;
    pack_weights(w, at, ws, 0, a.M, a.J, a.Jpad, a.Mpad, 1, tapoff);
    a.stats = (stats && raw) ? stats->partials : nullptr;
    a.bs_out = nullptr; a.bs_y = nullptr; a.bs_mean = nullptr;
    launch_dma(a, (int)s.total_q);
    if (a.stats) stats->splits = a.stats_splits;
    return true;
}

bool conv_backward_data_dma(const float* w, const float* dy, float* dx, const ConvShape& s, DxBnSums* bs) {
    if (bs) bs->splits = 0;
    if (!dma_supported(s, s.Cg, s.Mg, (size_t)s.N * s.F * s.OHOW)) return false;
    const int kk2 = s.pointwise ? 1 : s.ksz * s.ksz;
    DmaArgs a;
    a.rowmode = 0;
    a.b_base = dy; a.out = dx; a.bias = nullptr; a.slopes = nullptr; a.s = s;
    a.mode = 1; a.act = BCNN_HIP_ACT_NONE; a.add_bias = 0; a.stats = nullptr; a.stats_splits = 0;
    a.bs_out = nullptr; a.bs_y = nullptr; a.bs_mean = nullptr;
    a.M = s.Cg; a.J = s.Mg; a.Jpad = round_up(a.J, kDmaBK); a.Mpad = round_up(a.M, 128); a.kk2 = kk2;
    const size_t at_floats = (size_t)s.groups * kk2 * a.Jpad * a.Mpad;
    float* at = prepack_take(w, PREPACK_IGEMM, 1, at_floats);  // packed ahead by bcnn_hip_conv_prepack?
    const bool packed = at != nullptr;
    if (!packed) at = dma_scratch(at_floats);
    a.at = at; a.at_bytes = (unsigned)(at_floats * 4);
    a.b_bytes = (unsigned)((size_t)s.N * s.F * s.OHOW * 4);
    a.b_major_stride = s.OHOW;
    unsigned char tapoff[kDmaMaxTaps];
    dx_tap_order(s, tapoff);
    if (s.pointwise) {
        a.nclass = 1;
        DmaClass& ci = a.cls[0];
        ci.ih0 = 0; ci.iw0 = 0; ci.Hc = s.OH; ci.Wc = s.OW; ci.ntaps = 1; ci.tap0 = 0; ci.nkx = 1; ci.sgn = -1;
        if (!packed) pack_weights(w, at, s, 1, a.M, a.J, a.Jpad, a.Mpad, kk2, tapoff);
        // one group, stride 1 (the raw-view quirk then is the identity): the stored tile IS the gradient of the tensor
        // the batch-norm node in front wrote
        if (bs && bs->partials && pick_dma_tile(a, (int)s.total_q) == 4 && s.groups == 1 && s.HW == s.OHOW &&
            s.total_q == s.total_p &&
            bs->capacity >= (size_t)s.C * (size_t)ceil_div(s.total_q, 64) * 2) {
            a.bs_out = bs->partials; a.bs_y = bs->y; a.bs_mean = bs->mean;
        }
        launch_dma(a, (int)s.total_q);
        if (a.bs_out) bs->splits = a.stats_splits;
        return true;
    }
    // stride-parity classes; the packed tap order is class-major
    const int st = s.stride;
    struct Pending { DmaClass c; int cols; };
    Pending all[49];
    int ncls = 0, tap0 = 0;
    for (int ra = 0; ra < st; ++ra)
        for (int rb = 0; rb < st; ++rb) {
            DmaClass ci;
            ci.ih0 = ((ra - s.pad) % st + st) % st;  // first row with (ih + pad) % st == ra
            ci.iw0 = ((rb - s.pad) % st + st) % st;
            ci.Hc = ci.ih0 < s.H ? (s.H - ci.ih0 + st - 1) / st : 0;
            ci.Wc = ci.iw0 < s.W ? (s.W - ci.iw0 + st - 1) / st : 0;
            ci.ntaps = 0; ci.tap0 = tap0; ci.sgn = -1;
            ci.nkx = rb < s.ksz ? (s.ksz - rb + st - 1) / st : 0;  // kc = rb, rb + st, ... ; tap (i, j) <-> (kr/st, kc/st)
            for (int kr = ra; kr < s.ksz; kr += st)
                for (int kc = rb; kc < s.ksz; kc += st) ++ci.ntaps;  // dx_tap_order lists them in this order
            tap0 += ci.ntaps;
            all[ncls].c = ci;
            all[ncls].cols = s.N * ci.Hc * ci.Wc;
            ++ncls;
        }
    if (!packed) pack_weights(w, at, s, 1, a.M, a.J, a.Jpad, a.Mpad, kk2, tapoff);
    for (int c0 = 0; c0 < ncls; c0 += kDmaMaxClasses) {
        int nc = 0, max_cols = 0;
        for (int c = c0; c < ncls && nc < kDmaMaxClasses; ++c, ++nc) {
            a.cls[nc] = all[c].c;
            if (all[c].cols > max_cols) max_cols = all[c].cols;
        }
        a.nclass = nc;
        if (max_cols > 0) launch_dma(a, max_cols);
    }
    return true;
}

}  // namespace bcnn_hip
