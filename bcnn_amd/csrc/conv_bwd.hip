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
//   dX lives in conv_igemm.hip (it is the same gather-GEMM as the forward pass).
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
// OUTS outputs x (256 / OUTS) interleaved sub-sums per workgroup, combined in a fixed order (deterministic): the split
// counts run into the hundreds for small GEMMs, a single thread per output would be a serial latency chain -- and with
// ~1000 partials (the RGB layer of MobileNet: 864 outputs) even 16 sub-sums are one (79 us; 4 outputs x 64 sub-sums: 12 us).
template <int OUTS>
__global__ __launch_bounds__(256) void conv_dw_finalize_kernel(const float* __restrict__ partials, int nparts,
                                                               int groups, int Mg, int K, int MP, int NP,
                                                               int bias_col, float* __restrict__ dw,
                                                               float* __restrict__ dbias) {
    constexpr int SUBS = 256 / OUTS;
    __shared__ float red[SUBS][OUTS + 1];
    const int tx = threadIdx.x % OUTS, ty = threadIdx.x / OUTS;
    const int kcols = K + (bias_col ? 1 : 0);
    const long long total = (long long)groups * Mg * kcols;
    const long long i = (long long)blockIdx.x * OUTS + tx;
    int k = 0, f = 0, g = 0;
    float sum = 0.f;
    if (i < total) {
        k = (int)(i % kcols);
        const long long t = i / kcols;
        f = (int)(t % Mg); g = (int)(t / Mg);
        for (int p = ty; p < nparts; p += SUBS) sum += partials[(((long long)p * groups + g) * MP + f) * NP + k];
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && i < total) {
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < SUBS; ++r) tot += red[r][tx];
        if (k < K) dw[((long long)g * Mg + f) * K + k] += tot;
        else dbias[g * Mg + f] += tot;
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
    if (p.qsplits * 4 >= 256)
        conv_dw_finalize_kernel<4><<<(unsigned)((total + 3) / 4), 256, 0, current_stream()>>>(
            workspace, p.qsplits * 4, s.groups, s.Mg, s.K, p.mtiles * p.TM * 32, p.ntiles * p.TN * 32, p.bias_col, dw, dbias);
    else
        conv_dw_finalize_kernel<16><<<(unsigned)((total + 15) / 16), 256, 0, current_stream()>>>(
            workspace, p.qsplits * 4, s.groups, s.Mg, s.K, p.mtiles * p.TM * 32, p.ntiles * p.TN * 32, p.bias_col, dw, dbias);
    KERNEL_CHECK();
    return p.bias_col != 0;
}

}  // namespace bcnn_hip
