// conv_fwd.hip -- convolution forward as ONE batched implicit GEMM on the fp32 matrix cores.
//
// Reference semantics: bcnn_forward_conv_layer_cpu, src/layers/bcnn_conv_layer.c:367-485
// (per image, per group: im2col -> sgemm(W[M x K], col[K x OH*OW]) -> +bias -> activation).
// Here: GEMM-M = F/groups, GEMM-K = C/groups*k*k, GEMM-N = N*OH*OW (the whole batch), the im2col
// matrix is never materialised: its elements are gathered straight from x while staging the
// B tile into LDS (zero for padding), the bias add and the activation run in the epilogue on the
// accumulator registers, so the output is written exactly once.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32): a 256-thread workgroup = WM x WN waves, each wave owns
// TM x TN accumulators of 32x32. Output pixels run along the MFMA column index (= lane & 31), so
// every accumulator register stores two 128-byte runs of consecutive pixels.
#include "conv_common.h"

namespace bcnn_hip {

struct ConvFwdArgs {
    const float* x;
    const float* w;
    const float* bias;    // may be NULL when !add_bias
    const float* slopes;  // PReLU, else NULL
    float* y;
    ConvShape s;
    int act;
    int add_bias;
    int mtiles, ptiles;
};

template <int WM, int WN, int TM, int TN, int BK>
__global__ __launch_bounds__(256) void conv_fwd_igemm(const ConvFwdArgs a) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int LDA = BM + 1;  // +1: the transposing A store (k along lanes) stays <= 4-way
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(256 % BN == 0 || BN % 256 == 0, "B staging map");
    constexpr int B_ROWS = (256 / BN) > 0 ? (256 / BN) : 1;  // k-rows staged per pass
    constexpr int B_IT = BK / B_ROWS;
    constexpr int A_IT = BM * BK / 256;
    static_assert(BN <= 256 && BK % B_ROWS == 0 && (BM * BK) % 256 == 0, "tile/threads mismatch");

    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][BN];
    __shared__ int2 ktab[2][BK];  // per k of the staged tile: {offset c*H*W + kr*W + kc, kr | kc << 16}

    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int g = blockIdx.y;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lb % a.mtiles, pt = lb / a.mtiles;
    const int f0 = mt * BM;
    const long long q0 = (long long)pt * BN;
    const int nk = (s.K + BK - 1) / BK;

    // ---- this thread's B column (one output pixel of the folded batch) ----
    const int bj = tid % BN, bk0 = tid / BN;
    const long long bq = q0 + bj;
    const bool bvalid = bq < s.total_q;
    int b_ih0 = 0, b_iw0 = 0, b_off = 0;
    const float* xg = a.x;
    {
        const long long qq = bvalid ? bq : 0;
        const int n = (int)(qq / s.OHOW), pix = (int)(qq % s.OHOW);
        xg = a.x + ((long long)n * s.C + (long long)g * s.Cg) * s.HW;
        if (s.pointwise) {
            b_off = pix;  // raw [Cg][OH*OW] view of the image-group (quirk 1)
        } else {
            const int oh = pix / s.OW, ow = pix % s.OW;
            b_ih0 = oh * s.stride - s.pad;
            b_iw0 = ow * s.stride - s.pad;
            b_off = b_ih0 * s.W + b_iw0;
        }
    }
    const unsigned uH = s.pointwise ? 1u : (unsigned)s.H, uW = s.pointwise ? 1u : (unsigned)s.W;

    // ---- this thread's A elements ----
    const int ak = tid % BK, am0 = tid / BK;
    const float* wg = a.w + (long long)g * s.Mg * s.K;

    auto fill_ktab = [&](int kt, int slot) {
        if (tid < BK) {
            const int k = kt * BK + tid;
            int2 e;
            if (k < s.K) {
                if (s.pointwise) {
                    e.x = k * s.OHOW;
                    e.y = 0;
                } else {
                    const int kk2 = s.ksz * s.ksz;
                    const int c = k / kk2, r = k - c * kk2;
                    const int kr = r / s.ksz, kc = r - kr * s.ksz;
                    e.x = c * s.HW + kr * s.W + kc;
                    e.y = kr | (kc << 16);
                }
            } else {
                e.x = 0;
                e.y = 0x4000;  // kr far outside the image: the bounds test fails -> zero
            }
            ktab[slot][tid] = e;
        }
    };

    float ra[A_IT], rb[B_IT];
    auto load_tile = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = am0 + i * (256 / BK);
            const int k = kt * BK + ak;
            const bool ok = (f0 + m < s.Mg) && (k < s.K);
            ra[i] = ok ? wg[(long long)(f0 + m) * s.K + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int kk = bk0 + i * B_ROWS;
            const int2 e = ktab[slot][kk];
            const int kr = e.y & 0xffff, kc = e.y >> 16;
            const bool ok = bvalid && ((unsigned)(b_ih0 + kr) < uH) && ((unsigned)(b_iw0 + kc) < uW);
            rb[i] = ok ? xg[b_off + e.x] : 0.f;
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
    load_tile(0, 0);
    store_tile(0);
    if (nk > 1) fill_ktab(1, 1);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1, cur ^ 1);  // global loads in flight under the MFMAs
        int kleft = s.K - kt * BK;
        if (kleft > BK) kleft = BK;
        const int ksteps = (kleft + 1) >> 1;  // the LDS tile is zero-padded to an even k
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

    // ---- epilogue: bias (quirk 2) + activation on the accumulators, one store per element ----
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long long q = q0 + (wn * TN + j) * 32 + l31;
        if (q >= s.total_q) continue;
        const int n = (int)(q / s.OHOW), pix = (int)(q % s.OHOW);
        float* yb = a.y + ((long long)n * s.F + (long long)g * s.Mg) * s.OHOW + pix;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + (wm * TM + i) * 32 + mfma_row(r, lane);
                if (f >= s.Mg) continue;
                float v = acc[i][j][r];
                const int fc = g * s.Mg + f;
                if (a.add_bias) {
                    const float b = a.bias[fc];
                    // bcnn_add_scalar (AVX build) adds nothing for exactly 0.0f and exactly 1.0f
                    if (b != 0.0f && b != 1.0f) v += b;
                }
                if (a.act != BCNN_HIP_ACT_NONE)
                    v = act_fwd_cheap(v, a.act, a.act == BCNN_HIP_ACT_PRELU ? a.slopes[fc] : 0.f);
                yb[(long long)f * s.OHOW] = v;
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, int BK>
static void launch_fwd(ConvFwdArgs& a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    a.mtiles = ceil_div(a.s.Mg, BM);
    a.ptiles = ceil_div(a.s.total_q, BN);
    dim3 grid((unsigned)(a.mtiles * a.ptiles), (unsigned)a.s.groups);
    conv_fwd_igemm<WM, WN, TM, TN, BK><<<grid, 256, 0, current_stream()>>>(a);
    KERNEL_CHECK();
}

// raw = 1: write the bare convolution (no bias, no activation) -- used by the fused-BN path.
void conv_forward_dispatch(const float* x, const float* w, const float* bias, const float* slopes,
                           float* y, const ConvShape& s, int act, int raw) {
    ConvFwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.slopes = slopes; a.y = y; a.s = s;
    a.act = raw ? BCNN_HIP_ACT_NONE : act;
    a.add_bias = raw ? 0 : 1;
    if (s.total_q == 0 || s.Mg == 0) return;
    KTimer kt(K_CONV_FWD, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    if (s.Mg <= 32) {
        launch_fwd<1, 4, 1, 1, 16>(a);       // 32 x 128
    } else if (s.Mg <= 64 || (long long)ceil_div(s.Mg, 128) * ceil_div(s.total_q, 128) * s.groups < 2 * kCUs) {
        launch_fwd<2, 2, 1, 2, 16>(a);       // 64 x 128
    } else {
        launch_fwd<2, 2, 2, 2, 16>(a);       // 128 x 128
    }
}

}  // namespace bcnn_hip
