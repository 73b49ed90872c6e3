// conv_dw_dma.hip -- convolution weight gradient as a per-tap GEMM whose operand tiles are staged global -> LDS
// by LDS-DMA, with no vector-ALU work inside the MFMA loop (see conv_igemm_dma.hip for why that matters).
//
// Reference semantics: bcnn_backward_conv_layer_cpu, src/layers/bcnn_conv_layer.c:556-561
//   per image i, group j:  dW_j += dY_ij [Mg x OH*OW] * im2col(x_ij)^T      (beta = 1, accumulates)
// incl. the 1x1 case where the "col" matrix is the raw source buffer viewed as [Cg][OH*OW] (:562-569).
//
// For one filter tap t = (kr, kc):   dW[f][c][t] = sum_q dY[f][q] * X_t[c][q],   q = (n, oh, ow),
//   X_t[c][q] = x[n][c][oh*s - p + kr][ow*s - p + kc]   (0 outside the image)
// i.e. a GEMM with M = Mg, N = Cg and the reduction over all output pixels of the batch. A workgroup owns a
// (f-tile, c-tile, tap, q-range); partial tiles go to the workspace and a second kernel adds them to dW in a
// fixed order (deterministic; keeps the `+=` onto the momentum carry).
//
// Data movement: both tiles are [row][32 q] with q contiguous in memory, so one DMA instruction fills TWO
// rows (lanes 0-31 / 32-63): the per-lane VGPR offset carries (n, pixel, tap shift, validity, row parity),
// the wave-uniform SGPR offset carries the row (f or c). Row pairs are laid 66 floats apart in LDS, which
// makes the MFMA fragment reads (ds_read_b64: lane (row, k-pair)) hit all 64 banks exactly once (the
// compiler pairs them into ds_read2_b64, which is 2-way conflicted on its 32-bank view; LDS is ~20 % busy so
// that is harmless, and the alternative -- interleaving the two rows lane by lane so that ds_read2_b32 is
// conflict-free -- measured 6 % SLOWER because the global side of the DMA then gathers 8-byte pieces).
#include "conv_common.h"
#include "lds_dma.h"

namespace bcnn_hip {

struct DwDmaArgs {
    const float* x;
    const float* dy;
    float* partials;       // [qsplits][groups][kk2][Mpad][Npad]
    ConvShape s;
    int mtiles, ntiles;    // f tiles, c tiles
    int qsplits, q_per_split;  // q_per_split is a multiple of 32
    int Mpad, Npad;
    unsigned x_bytes, dy_bytes;
    unsigned ow_magic;     // ceil(2^32 / OW)
    int b_row_stride;      // elements between consecutive c rows of the gathered operand (HW; OH*OW for 1x1)
    int kk2;
    // rowmode (few input channels, the RGB stem): x is a zero-padded copy and the GEMM-N rows are ALL (c, kr, kc)
    // of ONE "tap": row j sits at the wave-uniform offset c * plane + kr * pitch + kc from the lane's pixel
    int rowmode, nrows, row_kk, row_ks, row_plane, row_pitch;
    unsigned row_kk_magic, row_ks_magic;
};

constexpr int DWQ = 32;    // q per K-tile
constexpr int DWPAIR = 66; // floats per LDS row pair (2 x 32 + 2 pad)

template <int WM, int WN, int WTM, int WTN>
__global__ __launch_bounds__(64 * WM * WN) void conv_dw_dma_kernel(const DwDmaArgs a) {
    constexpr int NW = WM * WN;  // waves per workgroup, arranged WM x WN, each owning WTM x WTN accumulators
    constexpr int BM = 32 * WM * WTM, BN = 32 * WN * WTN;
    static_assert((BM / 2) % NW == 0 && (BN / 2) % NW == 0, "row pairs must split evenly over the waves");
    constexpr int APAIRS = BM / 2, BPAIRS = BN / 2;
    constexpr int BUF = (APAIRS + BPAIRS) * DWPAIR;  // floats per stage buffer
    __shared__ float lds[2 * BUF];

    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int g = blockIdx.y;
    // taps vary fastest so the workgroups that re-read the same dY / x range run together (L2 reuse)
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tap = lb % a.kk2;
    const int rest = lb / a.kk2;
    const int tiles = a.mtiles * a.ntiles;
    const int tile = rest % tiles, qs = rest / tiles;
    const int mt = tile % a.mtiles, nt = tile / a.mtiles;
    const int f0 = mt * BM, c0 = nt * BN;
    const int kr = s.pointwise ? 0 : tap / s.ksz, kc = s.pointwise ? 0 : tap - (tap / s.ksz) * s.ksz;

    const unsigned qbeg = (unsigned)qs * (unsigned)a.q_per_split;
    unsigned qend = qbeg + (unsigned)a.q_per_split;
    if (qend > (unsigned)s.total_q) qend = (unsigned)s.total_q;
    const int nkt = qbeg < qend ? (int)((qend - qbeg + DWQ - 1) / DWQ) : 0;

    // ---- per-lane gather state: lane = (q within the K-tile, row parity) -------------------------------
    unsigned q = qbeg + (unsigned)l31;
    unsigned n = q / (unsigned)s.OHOW;
    unsigned pix = q - n * (unsigned)s.OHOW;
    const unsigned row_a = (unsigned)lhi * (unsigned)s.OHOW, row_b = a.rowmode ? 0u : (unsigned)lhi * (unsigned)a.b_row_stride;
    unsigned va = kOOB, vb = kOOB;
    auto lane_offsets = [&]() {
        const bool qv = q < qend;
        const unsigned offa = n * (unsigned)(s.F * s.OHOW) + pix + row_a;
        unsigned offb;
        bool ok = qv;
        if (s.pointwise) {
            offb = n * (unsigned)(s.C * s.HW) + pix + row_b;
        } else {
            const unsigned oh = magic_div(pix, a.ow_magic), ow = pix - oh * (unsigned)s.OW;
            const int ih = (int)oh * s.stride - s.pad + kr, iw = (int)ow * s.stride - s.pad + kc;
            ok = ok && (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
            offb = (n * (unsigned)(s.C * s.H) + (unsigned)ih) * (unsigned)s.W + (unsigned)iw + row_b;
        }
        va = qv ? offa * 4u : kOOB;
        vb = ok ? offb * 4u : kOOB;
    };
    auto advance = [&]() {  // q += 32 (OH*OW >= 32 is a launch precondition)
        q += DWQ; pix += DWQ;
        if (pix >= (unsigned)s.OHOW) { pix -= (unsigned)s.OHOW; ++n; }
    };

    const rsrc_i4 rs_a = make_rsrc(a.dy, a.dy_bytes), rs_b = make_rsrc(a.x, a.x_bytes);
    const unsigned lds0 = lds_offset(&lds[0]);
    const unsigned a_row0 = (unsigned)(g * s.Mg), b_grp = (unsigned)(g * s.Cg) * (unsigned)s.HW;

    // wave `wid` stages APAIRS/NW + BPAIRS/NW row pairs; rows past the end are fetched from row 0 (finite
    // values that only reach accumulator rows / columns which are never published)
    auto stage = [&](int buf) {
        const unsigned base = lds0 + (unsigned)(buf * BUF * 4);
#pragma unroll
        for (int i = 0; i < APAIRS / NW; ++i) {
            const int p = wid * (APAIRS / NW) + i;
            const int f = f0 + 2 * p;
            const unsigned soff = (a_row0 + (unsigned)(f < s.Mg ? f : 0)) * (unsigned)s.OHOW * 4u;
            dma_row(rs_a, base + (unsigned)(p * DWPAIR * 4), va, soff);
        }
#pragma unroll
        for (int i = 0; i < BPAIRS / NW; ++i) {
            const int p = wid * (BPAIRS / NW) + i;
            const int c = c0 + 2 * p;
            if (a.rowmode) {
                // rows (c, c + 1) of the pair sit at unrelated offsets: the upper half-wave adds the difference
                auto row_off = [&](int j) -> unsigned {
                    if (j >= a.nrows) j = 0;  // finite values into columns that are never published
                    const unsigned ch = magic_div((unsigned)j, a.row_kk_magic), r = (unsigned)j - ch * (unsigned)a.row_kk;
                    const unsigned kr = magic_div(r, a.row_ks_magic), kc = r - kr * (unsigned)a.row_ks;
                    return (ch * (unsigned)a.row_plane + kr * (unsigned)a.row_pitch + kc) * 4u;
                };
                const unsigned o0 = row_off(c), o1 = row_off(c + 1);
                const unsigned vbp = vb == kOOB ? kOOB : vb + (unsigned)lhi * (o1 - o0);
                dma_row(rs_b, base + (unsigned)((APAIRS + p) * DWPAIR * 4), vbp, o0);
            } else {
                const unsigned soff = (b_grp + (unsigned)(c < s.Cg ? c : 0) * (unsigned)a.b_row_stride) * 4u;
                dma_row(rs_b, base + (unsigned)((APAIRS + p) * DWPAIR * 4), vb, soff);
            }
        }
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses (floats): row rr -> (rr >> 1) * 66 + (rr & 1) * 32, k-pair offset 2 * lhi
    int a_off[WTM], b_off[WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i) a_off[i] = ((wm * WTM + i) * 16 + (l31 >> 1)) * DWPAIR + (l31 & 1) * 32 + 2 * lhi;
#pragma unroll
    for (int j = 0; j < WTN; ++j) b_off[j] = (APAIRS + (wn * WTN + j) * 16 + (l31 >> 1)) * DWPAIR + (l31 & 1) * 32 + 2 * lhi;

    if (nkt > 0) {
        lane_offsets();
        stage(0);
    }
    dma_wait();
    __syncthreads();
    for (int it = 0; it < nkt; ++it) {
        const int cur = it & 1;
        if (it + 1 < nkt) {
            advance();
            lane_offsets();
            stage(cur ^ 1);  // DMA in flight under the MFMAs
        }
        const float* buf = lds + cur * BUF;
#pragma unroll
        for (int kp = 0; kp < DWQ / 4; ++kp) {
            float2 av[WTM], bv[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) av[i] = *reinterpret_cast<const float2*>(buf + a_off[i] + 4 * kp);
#pragma unroll
            for (int j = 0; j < WTN; ++j) bv[j] = *reinterpret_cast<const float2*>(buf + b_off[j] + 4 * kp);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) acc[i][j] = mfma32(av[i].x, bv[j].x, acc[i][j]);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) acc[i][j] = mfma32(av[i].y, bv[j].y, acc[i][j]);
        }
        dma_wait();
        __syncthreads();
    }

    // ---- publish the partial tile ---------------------------------------------------------------------
    float* out = a.partials + ((((size_t)qs * s.groups + g) * a.kk2 + tap) * a.Mpad) * (size_t)a.Npad;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + (wm * WTM + i) * 32 + mfma_row(r, lane);
                const int c = c0 + (wn * WTN + j) * 32 + l31;
                out[(size_t)f * a.Npad + c] = v[r];
            }
        }
}

// dW[g][f][c][tap] += sum_qs partials[qs][g][tap][f][c]   (fixed order => deterministic)
// 64 outputs x NSUB interleaved sub-sums per workgroup; threads run along c so the (qsplits x larger) partial
// reads are coalesced and only the single dW read-modify-write is strided by the tap count. NSUB = 4 for the
// 3x3 layers (7-8 splits); NSUB = 16 with four independent loads in flight per thread for the pointwise layers,
// whose 32-64 splits lie 1 MB apart (a single dependent chain per output ran at 1.3 TB/s: 52 -> 26 us per launch).
template <int NSUB>
__global__ __launch_bounds__(64 * NSUB) void conv_dw_dma_finalize_kernel(const float* __restrict__ partials,
                                                                        int qsplits, int groups, int Mg, int Cg,
                                                                        int kk2, int Mpad, int Npad,
                                                                        float* __restrict__ dw,
                                                                        const float* __restrict__ fold_var,
                                                                        const float* __restrict__ fold_scales) {
    __shared__ float red[NSUB][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t total = (size_t)groups * kk2 * Mg * Cg;
    const size_t i = (size_t)blockIdx.x * 64 + tx;
    const size_t plane = (size_t)Mpad * Npad;
    const size_t qstride = (size_t)groups * kk2 * plane;
    float sum = 0.f;
    size_t o = 0;
    if (i < total) {
        const int c = (int)(i % Cg);
        size_t t = i / Cg;
        const int f = (int)(t % Mg);
        t /= Mg;
        const int tap = (int)(t % kk2), g = (int)(t / kk2);
        const float* p = partials + ((size_t)g * kk2 + tap) * plane + (size_t)f * Npad + c;
        int qs = ty;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; qs + 3 * NSUB < qsplits; qs += 4 * NSUB) {
            const float v0 = p[(size_t)qs * qstride], v1 = p[(size_t)(qs + NSUB) * qstride];
            const float v2 = p[(size_t)(qs + 2 * NSUB) * qstride], v3 = p[(size_t)(qs + 3 * NSUB) * qstride];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; qs < qsplits; qs += NSUB) s0 += p[(size_t)qs * qstride];
        sum = (s0 + s1) + (s2 + s3);
        o = (((size_t)g * Mg + f) * Cg + c) * kk2 + tap;
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && i < total) {
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < NSUB; ++r) tot += red[r][tx];
        if (fold_var) tot *= bnfold_a(fold_var, fold_scales, (int)((i / ((size_t)Mg * Cg * kk2)) * Cg + (i % Cg)));  // BnFold: d/dW of W diag(a)
        dw[o] += tot;
    }
}

// The same sums (same association per element: bit-identical results) for ONE tap -- the pointwise layers, where dw[f][c] is
// contiguous in c -- with 16 bytes per thread: a wave row reads 1 KB contiguous pieces of every split instead of 256 bytes
// (the 4-byte version read the 32 MB of a 512 x 512 layer's 32 splits at 1.5 TB/s, 22 us per launch, 12 launches per
// MobileNet step). Requires Cg % 4 == 0 and 16-byte aligned dw / partials.
template <int NSUB>
__global__ __launch_bounds__(64 * NSUB) void conv_dw_dma_finalize_x4_kernel(const float* __restrict__ partials, int qsplits,
                                                                           int groups, int Mg, int Cg, int Mpad, int Npad,
                                                                           float* __restrict__ dw,
                                                                           const float* __restrict__ fold_var,
                                                                           const float* __restrict__ fold_scales) {
    __shared__ float4 red[NSUB][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t total4 = (size_t)groups * Mg * Cg / 4;
    const size_t i4 = (size_t)blockIdx.x * 64 + tx;
    const size_t plane = (size_t)Mpad * Npad;
    const size_t qstride = (size_t)groups * plane;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < total4) {
        const size_t i = i4 * 4;
        const int c = (int)(i % Cg);
        const size_t t = i / Cg;
        const int f = (int)(t % Mg), g = (int)(t / Mg);
        const float* p = partials + (size_t)g * plane + (size_t)f * Npad + c;
        auto ld = [&](int qs) { return *reinterpret_cast<const float4*>(p + (size_t)qs * qstride); };
        auto acc = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
        float4 s0 = sum, s1 = sum, s2 = sum, s3 = sum;
        int qs = ty;
        for (; qs + 3 * NSUB < qsplits; qs += 4 * NSUB) {
            const float4 v0 = ld(qs), v1 = ld(qs + NSUB), v2 = ld(qs + 2 * NSUB), v3 = ld(qs + 3 * NSUB);
            acc(s0, v0); acc(s1, v1); acc(s2, v2); acc(s3, v3);
        }
        for (; qs < qsplits; qs += NSUB) acc(s0, ld(qs));
        sum = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                          (s0.w + s1.w) + (s2.w + s3.w));
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && i4 < total4) {
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < NSUB; ++r) { tot.x += red[r][tx].x; tot.y += red[r][tx].y; tot.z += red[r][tx].z; tot.w += red[r][tx].w; }
        float4* o = reinterpret_cast<float4*>(dw) + i4;  // (((g * Mg + f) * Cg + c) * 1 + 0) / 4 == i4
        if (fold_var) {  // BnFold: d/dW of W diag(a); c .. c + 3 are four consecutive input channels of group g
            const size_t i = i4 * 4;
            const int c = (int)(i % Cg) + (int)(i / ((size_t)Mg * Cg)) * Cg;
            tot.x *= bnfold_a(fold_var, fold_scales, c); tot.y *= bnfold_a(fold_var, fold_scales, c + 1);
            tot.z *= bnfold_a(fold_var, fold_scales, c + 2); tot.w *= bnfold_a(fold_var, fold_scales, c + 3);
        }
        float4 d = *o;
        d.x += tot.x; d.y += tot.y; d.z += tot.z; d.w += tot.w;
        *o = d;
    }
}

// ---- host side ------------------------------------------------------------------------------------------
struct DwTile { int bm, bn, threads; };
constexpr int kNumDwTiles = 6;
static const DwTile kDwTiles[kNumDwTiles] = {
    {64, 64, 256},    // 0: 2x2 waves of 32x32
    {128, 64, 256},   // 1: 2x2 waves of 64x32
    {128, 64, 512},   // 2: 4x2 waves of 32x32
    {128, 128, 256},  // 3: 2x2 waves of 64x64
    {128, 128, 1024}, // 4: 4x4 waves of 32x32
    {64, 128, 512},   // 5: 2x4 waves of 32x32
};

struct DwDmaPlan {
    bool ok;
    int cfg;  // index into kDwTiles
    int mtiles, ntiles, qsplits, q_per_split, Mpad, Npad, kk2;
    size_t partial_floats;
};

static DwDmaPlan plan_dw_dma(const ConvShape& s) {
    DwDmaPlan p;
    p.ok = false;
    p.partial_floats = 0;
    if (s.ksz > 7 && !s.pointwise) return p;
    if (s.Mg < 32 || s.Cg < 16 || (s.Mg & 1) || (s.Cg & 1)) return p;  // row pairs; tiny GEMMs stay on conv_bwd.hip
    if (s.OHOW < DWQ || s.total_q < 4 * DWQ) return p;
    if ((size_t)s.N * s.C * s.HW * 4 >= 0x7ffffff0ull || (size_t)s.N * s.F * s.OHOW * 4 >= 0x7ffffff0ull) return p;
    p.kk2 = s.pointwise ? 1 : s.ksz * s.ksz;
    p.cfg = s.Mg > 64 ? 1 : 0;
    static const char* forced = BCNN_EXP_ENV("BCNN_HIP_DW_TILE");  // experiments: index into kDwTiles
    if (forced && forced[0] >= '0' && forced[0] < '0' + kNumDwTiles) p.cfg = forced[0] - '0';
    static const char* wantenv = BCNN_EXP_ENV("BCNN_HIP_DW_WANT");
    const int want_per_cu = wantenv ? atoi(wantenv) : 8;
    const int BM = kDwTiles[p.cfg].bm, BN = kDwTiles[p.cfg].bn;
    p.mtiles = ceil_div(s.Mg, BM); p.ntiles = ceil_div(s.Cg, BN);
    p.Mpad = p.mtiles * BM; p.Npad = p.ntiles * BN;
    const long long tiles = (long long)p.mtiles * p.ntiles * p.kk2 * s.groups;
    long long want = ((long long)want_per_cu * kCUs + tiles - 1) / tiles;  // workgroups per CU in total
    const long long maxs = (s.total_q + 8 * DWQ - 1) / (8 * DWQ);      // >= 8 K-tiles per workgroup
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    long long per = (s.total_q + want - 1) / want;
    per = (per + DWQ - 1) / DWQ * DWQ;
    p.q_per_split = (int)per;
    p.qsplits = (int)((s.total_q + per - 1) / per);
    p.partial_floats = (size_t)p.qsplits * s.groups * p.kk2 * (size_t)p.Mpad * p.Npad;
    p.ok = true;
    return p;
}

size_t conv_dw_dma_workspace_floats(const ConvShape& s) { return plan_dw_dma(s).partial_floats; }

static unsigned magic_of_u(int d) { return d > 1 ? (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d) : 0u; }

// Returns false when the shape is not covered (caller falls back to conv_bwd.hip's kernel).
// fold: the layer ran on W diag(a) (BnFold, conv_common.h): the weight gradient's columns take the same factors
bool conv_backward_weights_dma(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                               size_t workspace_floats, const BnFold* fold) {
    const DwDmaPlan p = plan_dw_dma(s);
    if (!p.ok) return false;
    if (workspace == nullptr || workspace_floats < p.partial_floats) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n",
                workspace_floats, p.partial_floats);
        exit(1);
    }
    DwDmaArgs a;
    a.rowmode = 0; a.nrows = 0;
    a.x = x; a.dy = dy; a.partials = workspace; a.s = s;
    a.mtiles = p.mtiles; a.ntiles = p.ntiles; a.qsplits = p.qsplits; a.q_per_split = p.q_per_split;
    a.Mpad = p.Mpad; a.Npad = p.Npad; a.kk2 = p.kk2;
    a.x_bytes = (unsigned)((size_t)s.N * s.C * s.HW * 4);
    a.dy_bytes = (unsigned)((size_t)s.N * s.F * s.OHOW * 4);
    a.ow_magic = magic_of_u(s.OW);
    a.b_row_stride = s.pointwise ? s.OHOW : s.HW;
    dim3 grid((unsigned)(p.mtiles * p.ntiles * p.kk2 * p.qsplits), (unsigned)s.groups);
    const unsigned threads = (unsigned)kDwTiles[p.cfg].threads;
    trace_kernel("conv_dw_dma_kernel");
    switch (p.cfg) {
        case 0: conv_dw_dma_kernel<2, 2, 1, 1><<<grid, threads, 0, current_stream()>>>(a); break;
        case 1: conv_dw_dma_kernel<2, 2, 2, 1><<<grid, threads, 0, current_stream()>>>(a); break;
        case 2: conv_dw_dma_kernel<4, 2, 1, 1><<<grid, threads, 0, current_stream()>>>(a); break;
        case 3: conv_dw_dma_kernel<2, 2, 2, 2><<<grid, threads, 0, current_stream()>>>(a); break;
        case 4: conv_dw_dma_kernel<4, 4, 1, 1><<<grid, threads, 0, current_stream()>>>(a); break;
        default: conv_dw_dma_kernel<2, 4, 1, 1><<<grid, threads, 0, current_stream()>>>(a); break;
    }
    KERNEL_CHECK();
    const size_t total = (size_t)s.groups * s.Mg * s.Cg * p.kk2;
    static const bool x4_on = BCNN_EXP_ENV("BCNN_HIP_NO_DW_FINALIZE_X4") == nullptr;  // A/B switch (experiment build only)
    if (x4_on && p.kk2 == 1 && p.qsplits > 16 && (s.Cg & 3) == 0 && (p.Npad & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0)
        conv_dw_dma_finalize_x4_kernel<16><<<(unsigned)((total / 4 + 63) / 64), 1024, 0, current_stream()>>>(
            workspace, p.qsplits, s.groups, s.Mg, s.Cg, p.Mpad, p.Npad, dw, fold ? fold->var : nullptr, fold ? fold->scales : nullptr);
    else if (p.qsplits > 16)
        conv_dw_dma_finalize_kernel<16><<<(unsigned)((total + 63) / 64), 1024, 0, current_stream()>>>(
            workspace, p.qsplits, s.groups, s.Mg, s.Cg, p.kk2, p.Mpad, p.Npad, dw, fold ? fold->var : nullptr, fold ? fold->scales : nullptr);
    else
        conv_dw_dma_finalize_kernel<4><<<(unsigned)((total + 63) / 64), 256, 0, current_stream()>>>(
            workspace, p.qsplits, s.groups, s.Mg, s.Cg, p.kk2, p.Mpad, p.Npad, dw, fold ? fold->var : nullptr, fold ? fold->scales : nullptr);
    KERNEL_CHECK();
    return true;
}

// ---- few input channels (the RGB stem): dW[f][(c, kr, kc)] as ONE GEMM over the zero-padded input ----------
bool conv_small_c_applicable(const ConvShape& s);                                                    // conv_igemm_dma.hip
float* conv_small_c_padded_input(const float* x, const ConvShape& s, size_t extra_floats, float** extra, int for_dw);

static DwDmaPlan plan_dw_small_c(const ConvShape& s) {
    DwDmaPlan p;
    p.ok = false; p.partial_floats = 0;
    if (!conv_small_c_applicable(s) || (s.Mg & 1) || s.OHOW < DWQ || s.total_q < 4 * DWQ) return p;
    p.kk2 = 1;
    p.cfg = 0;  // 64 x 64 tiles: 147 rows -> three column tiles
    const int BM = kDwTiles[p.cfg].bm, BN = kDwTiles[p.cfg].bn;
    p.mtiles = ceil_div(s.Mg, BM); p.ntiles = ceil_div(s.K, BN);
    p.Mpad = p.mtiles * BM; p.Npad = p.ntiles * BN;
    const long long tiles = (long long)p.mtiles * p.ntiles;
    long long want = (8LL * kCUs + tiles - 1) / tiles;
    const long long maxs = (s.total_q + 8 * DWQ - 1) / (8 * DWQ);
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    long long per = (s.total_q + want - 1) / want;
    per = (per + DWQ - 1) / DWQ * DWQ;
    p.q_per_split = (int)per;
    p.qsplits = (int)((s.total_q + per - 1) / per);
    p.partial_floats = (size_t)p.qsplits * (size_t)p.Mpad * p.Npad;
    p.ok = true;
    return p;
}

size_t conv_dw_small_c_workspace_floats(const ConvShape& s) { return plan_dw_small_c(s).partial_floats; }

bool conv_backward_weights_small_c(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                   size_t workspace_floats) {
    const DwDmaPlan p = plan_dw_small_c(s);
    if (!p.ok) return false;
    if (workspace == nullptr || workspace_floats < p.partial_floats) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n", workspace_floats,
                p.partial_floats);
        exit(1);
    }
    const int Hp = s.H + 2 * s.pad, Wp = s.W + 2 * s.pad;
    float* xp = conv_small_c_padded_input(x, s, 0, nullptr, /*for_dw=*/1);
    const ConvShape sp = make_conv_shape(s.N, s.C, Hp, Wp, s.F, s.ksz, s.stride, 0, 1);
    DwDmaArgs a;
    a.x = xp; a.dy = dy; a.partials = workspace; a.s = sp;
    a.mtiles = p.mtiles; a.ntiles = p.ntiles; a.qsplits = p.qsplits; a.q_per_split = p.q_per_split;
    a.Mpad = p.Mpad; a.Npad = p.Npad; a.kk2 = 1;
    a.x_bytes = (unsigned)((size_t)s.N * s.C * Hp * Wp * 4);
    a.dy_bytes = (unsigned)((size_t)s.N * s.F * s.OHOW * 4);
    a.ow_magic = magic_of_u(sp.OW);
    a.b_row_stride = 0;
    a.rowmode = 1; a.nrows = s.K; a.row_kk = s.ksz * s.ksz; a.row_ks = s.ksz; a.row_plane = Hp * Wp; a.row_pitch = Wp;
    a.row_kk_magic = magic_of_u(a.row_kk); a.row_ks_magic = magic_of_u(a.row_ks);
    dim3 grid((unsigned)(p.mtiles * p.ntiles * p.qsplits), 1u);
    conv_dw_dma_kernel<2, 2, 1, 1><<<grid, (unsigned)kDwTiles[0].threads, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    // partials [qs][1][1][Mpad][Npad] -> dw[f][j] += ..., j = (c, kr, kc): the weight tensor's own layout
    const size_t total = (size_t)s.Mg * s.K;
    if (p.qsplits > 16)
        conv_dw_dma_finalize_kernel<16><<<(unsigned)((total + 63) / 64), 1024, 0, current_stream()>>>(
            workspace, p.qsplits, 1, s.Mg, s.K, 1, p.Mpad, p.Npad, dw, nullptr, nullptr);
    else
        conv_dw_dma_finalize_kernel<4><<<(unsigned)((total + 63) / 64), 256, 0, current_stream()>>>(
            workspace, p.qsplits, 1, s.Mg, s.K, 1, p.Mpad, p.Npad, dw, nullptr, nullptr);
    KERNEL_CHECK();
    return true;
}

}  // namespace bcnn_hip
