// conv_bwd.hip -- convolution backward: weight gradient and data gradient as batched implicit GEMMs
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Reference semantics: bcnn_backward_conv_layer_cpu, src/layers/bcnn_conv_layer.c:533-585
//   per image i, group j:  dW_j += G_ij [F/g x OH*OW] * im2col(x_ij)^T      (beta = 1, accumulates)
//                          dX_ij = col2im(W_j^T * G_ij)                      (zero-fill => overwrite)
//   1x1 kernels: the "col" matrix is the raw source buffer viewed as [C/g][OH*OW] and dX is written
//   as that same view (prefix of each image-group), regardless of stride/pad (:562-569).
// Here both are single launches over the whole batch:
//   dW: GEMM-M = F/g, GEMM-N = K (= C/g*k*k), reduction over q = (image, output pixel); split over q
//       across workgroups AND across the 4 waves of a workgroup; every wave writes its partial tile to
//       the workspace and a second kernel adds the partials to dW in a fixed order (deterministic;
//       keeps the `+=` onto the momentum carry). One extra all-ones im2col column yields the bias
//       gradient for free when the padded K tile has room (saves a full re-read of dy).
//   dX: gather form, GEMM-M = C/g, GEMM-N = N*H*W input pixels, reduction over (f, kr, kc); every
//       output element is produced by exactly one thread => plain store, no zero-fill pass, no atomics.
#include "conv_common.h"

namespace bcnn_hip {

// ================================================================================================
// dW
// ================================================================================================
struct ConvDwArgs {
    const float* x;
    const float* dy;
    float* partials;  // [nparts][groups][mtiles*BM][ntiles*BN]
    ConvShape s;
    int mtiles, ntiles, qsplits;
    int q_per_split;  // multiple of 64
    int bias_col;     // 1: column index K of the im2col matrix is all ones (=> bias gradient)
};

constexpr int DW_BQ = 64;  // q rows staged per step (16 per wave)

template <int TM, int TN>
__global__ __launch_bounds__(256) void conv_dw_kernel(const ConvDwArgs a) {
    constexpr int BM = TM * 32, BN = TN * 32;
    constexpr int LDA = BM + 1, LDB = BN + 1;  // odd strides: transposing stores and fragment reads conflict-free
    constexpr int A_IT = BM * DW_BQ / 256, B_IT = BN * DW_BQ / 256;
    __shared__ float As[DW_BQ][LDA];  // As[q][f]
    __shared__ float Bs[DW_BQ][LDB];  // Bs[q][k]
    __shared__ int2 ktab[BN];         // per k of this block's k-tile

    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.z;
    const int tile = blockIdx.x, qs = blockIdx.y;
    const int mt = tile % a.mtiles, nt = tile / a.mtiles;
    const int f0 = mt * BM, k0 = nt * BN;
    const long long qbeg = (long long)qs * a.q_per_split;
    long long qend = qbeg + a.q_per_split;
    if (qend > s.total_q) qend = s.total_q;

    if (tid < BN) {
        const int k = k0 + tid;
        int2 e;
        if (k < s.K) {
            if (s.pointwise) { e.x = k * s.OHOW; e.y = 0; }
            else {
                const int kk2 = s.ksz * s.ksz;
                const int c = k / kk2, r = k - c * kk2;
                const int kr = r / s.ksz, kc = r - kr * s.ksz;
                e.x = c * s.HW + kr * s.W + kc;
                e.y = kr | (kc << 16);
            }
        } else if (k == s.K && a.bias_col) {
            e.x = 0; e.y = 0x7fff0000;  // marker: all-ones column
        } else {
            e.x = 0; e.y = 0x4000;      // out of range row => zero
        }
        ktab[tid] = e;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int qi = tid & 63, r0 = tid >> 6;  // this thread stages column qi of the step, rows r0 + 4*i
    const unsigned uH = s.pointwise ? 1u : (unsigned)s.H, uW = s.pointwise ? 1u : (unsigned)s.W;
    const int l31 = lane & 31, lhi = lane >> 5;
    __syncthreads();

    float ra[A_IT], rb[B_IT];
    auto load_step = [&](long long qstep) {
        const long long q = qstep + qi;
        const bool qv = q < qend;
        const long long qq = qv ? q : 0;
        const int n = (int)(qq / s.OHOW), pix = (int)(qq - (long long)n * s.OHOW);
        const float* gp = a.dy + ((long long)n * s.F + (long long)g * s.Mg) * s.OHOW + pix;
        const float* xp = a.x + ((long long)n * s.C + (long long)g * s.Cg) * s.HW;
        int ih0 = 0, iw0 = 0, off = pix;
        if (!s.pointwise) {
            const int oh = pix / s.OW, ow = pix - oh * s.OW;
            ih0 = oh * s.stride - s.pad; iw0 = ow * s.stride - s.pad;
            off = ih0 * s.W + iw0;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int f = f0 + r0 + 4 * i;
            ra[i] = (qv && f < s.Mg) ? gp[(long long)f * s.OHOW] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int2 e = ktab[r0 + 4 * i];
            const int kr = e.y & 0xffff, kc = e.y >> 16;
            float v = 0.f;
            if (qv) {
                if (kc == 0x7fff) v = 1.0f;
                else if ((unsigned)(ih0 + kr) < uH && (unsigned)(iw0 + kc) < uW) v = xp[off + e.x];
            }
            rb[i] = v;
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) As[qi][r0 + 4 * i] = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) Bs[qi][r0 + 4 * i] = rb[i];
    };

    if (qbeg < qend) load_step(qbeg);
    for (long long qstep = qbeg; qstep < qend; qstep += DW_BQ) {
        __syncthreads();  // previous step's fragments consumed
        store_step();
        __syncthreads();
        if (qstep + DW_BQ < qend) load_step(qstep + DW_BQ);  // next step's loads fly under the MFMAs
        const int qrow = wid * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[qrow + 2 * ks + lhi][i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[qrow + 2 * ks + lhi][j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
    }

    // each wave publishes its own partial tile (its quarter of the block's q range)
    const int MP = a.mtiles * BM, NP = a.ntiles * BN;
    const int part = qs * 4 + wid;
    float* out = a.partials + (((long long)part * s.groups + g) * MP) * NP;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + i * 32 + mfma_row(r, lane);
                const int k = k0 + j * 32 + l31;
                out[(long long)f * NP + k] = acc[i][j][r];
            }
}

// dW[g][f][k] += sum_p partials[p][g][f][k]; column K (if bias_col) goes to dbias[g*Mg + f].
__global__ __launch_bounds__(256) void conv_dw_finalize_kernel(const float* __restrict__ partials, int nparts,
                                                               int groups, int Mg, int K, int MP, int NP,
                                                               int bias_col, float* __restrict__ dw,
                                                               float* __restrict__ dbias) {
    const int kcols = K + (bias_col ? 1 : 0);
    const long long total = (long long)groups * Mg * kcols;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int k = (int)(i % kcols);
        const long long t = i / kcols;
        const int f = (int)(t % Mg), g = (int)(t / Mg);
        float sum = 0.f;
        for (int p = 0; p < nparts; ++p)
            sum += partials[(((long long)p * groups + g) * MP + f) * NP + k];
        if (k < K) dw[((long long)g * Mg + f) * K + k] += sum;
        else dbias[g * Mg + f] += sum;
    }
}

struct DwPlan {
    int TM, TN, mtiles, ntiles, qsplits, q_per_split, bias_col;
    size_t partial_floats;
};

static DwPlan plan_dw(const ConvShape& s, bool want_bias_col) {
    DwPlan p;
    p.TM = (s.Mg <= 32) ? 1 : 2;
    p.TN = (s.K + (want_bias_col ? 1 : 0) <= 32) ? 1 : 2;
    const int BM = p.TM * 32, BN = p.TN * 32;
    p.mtiles = ceil_div(s.Mg, BM);
    p.bias_col = (want_bias_col && (s.K % BN) != 0) ? 1 : 0;  // needs a free slot in the padded k tile
    p.ntiles = ceil_div(s.K, BN);
    const long long tiles = (long long)p.mtiles * p.ntiles * s.groups;
    long long want = (4LL * kCUs + tiles - 1) / tiles;       // ~4 workgroups per CU
    const long long maxs = (s.total_q + 4 * DW_BQ - 1) / (4 * DW_BQ);  // >= 4 steps per workgroup
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    long long per = (s.total_q + want - 1) / want;
    per = (per + DW_BQ - 1) / DW_BQ * DW_BQ;
    p.q_per_split = (int)per;
    p.qsplits = (int)((s.total_q + per - 1) / per);
    p.partial_floats = (size_t)p.qsplits * 4 * s.groups * (size_t)(p.mtiles * BM) * (size_t)(p.ntiles * BN);
    return p;
}

size_t conv_dw_workspace_floats(const ConvShape& s) { return plan_dw(s, true).partial_floats; }

// returns true when the bias gradient was produced by the all-ones column
bool conv_backward_weights(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s,
                           float* workspace, size_t workspace_floats, bool want_bias) {
    if (s.total_q == 0 || s.Mg == 0 || s.K == 0) return false;
    const DwPlan p = plan_dw(s, want_bias && dbias != nullptr);
    if (workspace == nullptr || workspace_floats < p.partial_floats) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n",
                workspace_floats, p.partial_floats);
        exit(1);
    }
    KTimer kt(K_CONV_DW, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    ConvDwArgs a;
    a.x = x; a.dy = dy; a.partials = workspace; a.s = s;
    a.mtiles = p.mtiles; a.ntiles = p.ntiles; a.qsplits = p.qsplits; a.q_per_split = p.q_per_split;
    a.bias_col = p.bias_col;
    dim3 grid((unsigned)(p.mtiles * p.ntiles), (unsigned)p.qsplits, (unsigned)s.groups);
    if (p.TM == 1 && p.TN == 1) conv_dw_kernel<1, 1><<<grid, 256, 0, current_stream()>>>(a);
    else if (p.TM == 1 && p.TN == 2) conv_dw_kernel<1, 2><<<grid, 256, 0, current_stream()>>>(a);
    else if (p.TM == 2 && p.TN == 1) conv_dw_kernel<2, 1><<<grid, 256, 0, current_stream()>>>(a);
    else conv_dw_kernel<2, 2><<<grid, 256, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    const long long total = (long long)s.groups * s.Mg * (s.K + p.bias_col);
    conv_dw_finalize_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(
        workspace, p.qsplits * 4, s.groups, s.Mg, s.K, p.mtiles * p.TM * 32, p.ntiles * p.TN * 32, p.bias_col,
        dw, dbias);
    KERNEL_CHECK();
    return p.bias_col != 0;
}

// ================================================================================================
// dX
// ================================================================================================
struct ConvDxArgs {
    const float* w;
    const float* dy;
    float* dx;
    ConvShape s;
    int mtiles, ptiles;
    int KR;  // reduction length Mg*ksz*ksz
};

template <int WM, int WN, int TM, int TN, int BK>
__global__ __launch_bounds__(256) void conv_dx_kernel(const ConvDxArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDA = BM + 1;
    constexpr int B_ROWS = 256 / BN, B_IT = BK / B_ROWS, A_IT = BM * BK / 256;
    static_assert(WM * WN == 4 && BN <= 256 && 256 % BN == 0, "tile");
    __shared__ float As[2][BK][LDA];   // As[kred][c]
    __shared__ float Bs[2][BK][BN];    // Bs[kred][pixel]
    __shared__ int4 ktab[2][BK];       // {f*OHOW, kr | kc<<16, f*Cg*k2 + tap, valid}

    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int g = blockIdx.y;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lb % a.mtiles, pt = lb / a.mtiles;
    const int c0 = mt * BM;
    const long long p0 = (long long)pt * BN;
    const long long total_cols = s.pointwise ? s.total_q : s.total_p;
    const int col_per_img = s.pointwise ? s.OHOW : s.HW;
    const int kk2 = s.ksz * s.ksz;
    const int nk = (a.KR + BK - 1) / BK;

    // this thread's B column = one input pixel (or one output pixel in the 1x1 raw-view case)
    const int bj = tid % BN, bk0 = tid / BN;
    const long long bp = p0 + bj;
    const bool bvalid = bp < total_cols;
    int b_ih = 0, b_iw = 0, b_pix = 0;
    const float* gyb = a.dy;
    {
        const long long pp = bvalid ? bp : 0;
        const int n = (int)(pp / col_per_img), pix = (int)(pp - (long long)n * col_per_img);
        gyb = a.dy + ((long long)n * s.F + (long long)g * s.Mg) * s.OHOW;
        if (s.pointwise) b_pix = pix;
        else { b_ih = pix / s.W + s.pad; b_iw = pix % s.W + s.pad; }
    }
    const int ak = tid % BK, am0 = tid / BK;
    const float* wg = a.w + (long long)g * s.Mg * s.K;

    auto fill_ktab = [&](int kt, int slot) {
        if (tid < BK) {
            const int kr_ = kt * BK + tid;
            int4 e;
            if (kr_ < a.KR) {
                const int f = kr_ / kk2, tap = kr_ - f * kk2;
                const int kr = tap / s.ksz, kc = tap - kr * s.ksz;
                e.x = f * s.OHOW; e.y = kr | (kc << 16); e.z = f * s.K + tap; e.w = 1;
            } else { e.x = 0; e.y = 0; e.z = 0; e.w = 0; }
            ktab[slot][tid] = e;
        }
    };

    float ra[A_IT], rb[B_IT];
    auto load_tile = [&](int slot) {
        {
            const int4 e = ktab[slot][ak];
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int m = am0 + i * (256 / BK);
                const bool ok = e.w && (c0 + m < s.Cg);
                ra[i] = ok ? wg[e.z + (c0 + m) * kk2] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int4 e = ktab[slot][bk0 + i * B_ROWS];
            float v = 0.f;
            if (bvalid && e.w) {
                if (s.pointwise) {
                    v = gyb[e.x + b_pix];
                } else {
                    const int th = b_ih - (e.y & 0xffff), tw = b_iw - (e.y >> 16);
                    if (th >= 0 && tw >= 0) {
                        int oh = th, ow = tw;
                        bool ok = true;
                        if (s.stride != 1) {
                            oh = th / s.stride; ow = tw / s.stride;
                            ok = (oh * s.stride == th) && (ow * s.stride == tw);
                        }
                        if (ok && oh < s.OH && ow < s.OW) v = gyb[e.x + oh * s.OW + ow];
                    }
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) As[buf][ak][am0 + i * (256 / BK)] = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) Bs[buf][bk0 + i * B_ROWS][bj] = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fill_ktab(0, 0);
    __syncthreads();
    load_tile(0);
    store_tile(0);
    if (nk > 1) fill_ktab(1, 1);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(cur ^ 1);
        int kleft = a.KR - kt * BK;
        if (kleft > BK) kleft = BK;
        const int ksteps = (kleft + 1) >> 1;
        for (int ks = 0; ks < ksteps; ++ks) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[cur][2 * ks + lhi][(wm * TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[cur][2 * ks + lhi][(wn * TN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        if (kt + 2 < nk) fill_ktab(kt + 2, cur);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long long p = p0 + (wn * TN + j) * 32 + l31;
        if (p >= total_cols) continue;
        const int n = (int)(p / col_per_img), pix = (int)(p - (long long)n * col_per_img);
        float* ob = a.dx + ((long long)n * s.C + (long long)g * s.Cg) * s.HW + pix;
        const int cstride = s.pointwise ? s.OHOW : s.HW;  // raw [Cg][OH*OW] view for 1x1 (quirk 1)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (wm * TM + i) * 32 + mfma_row(r, lane);
                if (c < s.Cg) ob[(long long)c * cstride] = acc[i][j][r];
            }
    }
}

template <int WM, int WN, int TM, int TN, int BK>
static void launch_dx(ConvDxArgs& a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const long long cols = a.s.pointwise ? a.s.total_q : a.s.total_p;
    a.mtiles = ceil_div(a.s.Cg, BM);
    a.ptiles = ceil_div(cols, BN);
    dim3 grid((unsigned)(a.mtiles * a.ptiles), (unsigned)a.s.groups);
    conv_dx_kernel<WM, WN, TM, TN, BK><<<grid, 256, 0, current_stream()>>>(a);
    KERNEL_CHECK();
}

void conv_backward_data(const float* w, const float* dy, float* dx, const ConvShape& s) {
    if (s.total_p == 0 || s.Cg == 0) return;
    KTimer kt(K_CONV_DX, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    ConvDxArgs a;
    a.w = w; a.dy = dy; a.dx = dx; a.s = s; a.KR = s.Mg * s.ksz * s.ksz;
    const long long cols = s.pointwise ? s.total_q : s.total_p;
    if (s.Cg <= 32) launch_dx<1, 4, 1, 1, 16>(a);
    else if (s.Cg <= 64 || (long long)ceil_div(s.Cg, 128) * ceil_div(cols, 128) * s.groups < 2 * kCUs)
        launch_dx<2, 2, 1, 2, 16>(a);
    else launch_dx<2, 2, 2, 2, 16>(a);
}

}  // namespace bcnn_hip
