// conv_igemm.hip -- the general batched implicit-GEMM kernel of the convolution node, used for BOTH
// the forward pass and the data gradient (they are the same computation: D[m][col] = sum_r A[m][r] *
// B[r][col] where B is gathered on the fly from an NCHW tensor with zero fill).
//
// Reference semantics: bcnn_forward_conv_layer_cpu / bcnn_backward_conv_layer_cpu,
// src/layers/bcnn_conv_layer.c:438-481 and :556-581 (im2col + sgemm, sgemm^T + col2im), incl. the 1x1
// "raw [C/g][OH*OW] view" addressing (:445-446, :562-569).
//
//   forward : m = output channel f, r = (c, kr, kc), col = (n, oh, ow)
//             A = W[f][r], B = x[n][c][oh*s-p+kr][ow*s-p+kc], D -> y (+bias, activation)
//   dX      : m = input channel c, r = (f, tap), col = (n, ih, iw) restricted to ONE stride-parity class
//             A = W[f][c][tap], B = dy[n][f][(ih+p-kr)/s][(iw+p-kc)/s], D -> dx (plain store = overwrite)
//             For stride s the input pixels split into s*s classes by ((ih+p)%s, (iw+p)%s); a class only
//             ever meets the taps with kr%s, kc%s equal to its residues, so each class runs a dense GEMM
//             over its own tap list (3x3/s2: 4+2+2+1 taps instead of 4x9) -- no structural zeros on the MFMAs.
//
// Design for CDNA4: 256 threads = 4 waves (WM x WN), each wave TM x TN accumulators of 32x32
// (v_mfma_f32_32x32x2_f32, exact fp32). Per K-tile the B tile is gathered into LDS and the A tile is
// transposed through a +1-padded LDS row; the next tile's global loads are issued before the current
// tile's MFMAs (register staging, one barrier per K-tile). The K loop is kept lean on the VALU:
//   * a column's tap validity is ONE 64-bit mask computed once per thread (pixels are fixed per thread),
//     so a gathered element costs: table read, bit test, add, clamp -- and an UNCONDITIONAL load;
//   * table entries {A offset, B offset, tap index} are produced with multiply-high "magic" division;
//   * all offsets are 32-bit against wave-uniform bases.
#include "conv_common.h"

namespace bcnn_hip {

struct ClassInfo {        // dX: one stride-parity class
    int ih0, iw0;         // first input row / column of the class
    int Hc, Wc;           // rows / columns of the class per image
    int ntaps;            // taps with kr%s == ra && kc%s == rb
    unsigned char taps[52];  // kr | kc << 4 (ksz <= 7)
};
constexpr int kMaxClassesPerLaunch = 4;  // travels in the kernel arguments (no device table, no races)

struct IgemmArgs {
    const float* a_base;   // weights
    const float* b_base;   // gathered tensor (x for forward, dy for dX)
    float* out;            // y or dx
    const float* bias;     // forward epilogue (may be NULL)
    const float* slopes;   // PReLU (may be NULL)
    ConvShape s;
    int mode;              // 0 forward, 1 dX
    int act, add_bias;
    int M;                 // rows per group: Mg (forward) or Cg (dX)
    int KR;                // reduction length of this launch (per class for dX)
    int a_row_stride;      // A(m, r) = a_base[g*group_stride + m*a_row_stride + aoff(r)]
    long long a_group_stride;
    int mtiles, ptiles;
    // reduction index decode: r -> (major, tap) with major = r / ntaps (c for forward, f for dX)
    int ntaps;
    unsigned ntaps_magic;  // ceil(2^32 / ntaps)
    unsigned ksz_magic;    // ceil(2^32 / ksz)
    // dX only: stride-parity classes handled by this launch (blockIdx.z)
    int nclass;
    ClassInfo cls[kMaxClassesPerLaunch];
};


// n / d for small d via multiply-high with magic = ceil(2^32 / d) (exact while n*d < 2^32); d == 1 has
// magic 2^32, which does not fit: it is encoded as 0 and means "identity".
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

template <int WM, int WN, int TM, int TN, int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IgemmArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDA = BM + 1;
    constexpr int B_ROWS = 256 / BN, B_IT = BK / B_ROWS, A_IT = BM * BK / 256;
    static_assert(WM * WN == 4 && BN <= 256 && 256 % BN == 0 && (BM * BK) % 256 == 0, "tile");
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][BN];
    __shared__ int4 ktab[2][BK];  // {aoff, boff, tap index, valid}
    __shared__ int ctaps[49];     // dX: taps of this block's class

    const ConvShape& s = a.s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int g = blockIdx.y;
    const int cls = blockIdx.z;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lb % a.mtiles, pt = lb / a.mtiles;
    const int m0 = mt * BM;
    const bool fwd = (a.mode == 0);

    // ---- class geometry (dX) / trivial class (forward) -------------------------------------------
    int c_ih0 = 0, c_iw0 = 0, c_Hc = fwd ? s.OH : s.H, c_Wc = fwd ? s.OW : s.W, ntaps = a.ntaps, KR = a.KR;
    unsigned ntaps_magic = a.ntaps_magic;
    if (!fwd && !s.pointwise) {
        const ClassInfo& ci = a.cls[cls];
        c_ih0 = ci.ih0; c_iw0 = ci.iw0; c_Hc = ci.Hc; c_Wc = ci.Wc; ntaps = ci.ntaps;
        KR = s.Mg * ntaps;
        ntaps_magic = ntaps > 1 ? (unsigned)((0x100000000ULL + ntaps - 1) / ntaps) : 0u;
        if (tid < ntaps) ctaps[tid] = ci.taps[tid];
    }
    const int col_per_img = s.pointwise ? s.OHOW : c_Hc * c_Wc;
    const long long total_cols = (long long)s.N * col_per_img;
    const long long p0 = (long long)pt * BN;
    if (p0 >= total_cols) return;  // class smaller than the grid (uniform per block)
    const int nk = (KR + BK - 1) / BK;

    // ---- decode one column: base offset into the gathered tensor, tap validity mask, output offset ----
    // returns false for columns past the end
    auto decode = [&](long long col, bool need_mask, unsigned& bbase, unsigned long long& mask, unsigned& obase) -> bool {
        if (col >= total_cols) { bbase = 0; mask = 0; obase = 0; return false; }
        const unsigned n = (unsigned)(col / col_per_img);
        const unsigned pix = (unsigned)(col - (long long)n * col_per_img);
        if (s.pointwise) {
            // raw views: forward reads x[n][g] as [Cg][OH*OW], dX writes dx[n][g] as [Cg][OH*OW]
            if (fwd) {
                bbase = (n * (unsigned)s.C + (unsigned)(g * s.Cg)) * (unsigned)s.HW + pix;
                obase = (n * (unsigned)s.F + (unsigned)(g * s.Mg)) * (unsigned)s.OHOW + pix;
            } else {
                bbase = (n * (unsigned)s.F + (unsigned)(g * s.Mg)) * (unsigned)s.OHOW + pix;
                obase = (n * (unsigned)s.C + (unsigned)(g * s.Cg)) * (unsigned)s.HW + pix;
            }
            mask = 1ULL;
            return true;
        }
        const unsigned u = pix / (unsigned)c_Wc, v = pix - u * (unsigned)c_Wc;
        unsigned long long m = 0;
        if (fwd) {
            const int ih0 = (int)u * s.stride - s.pad, iw0 = (int)v * s.stride - s.pad;
            if (need_mask) {
                unsigned colm = 0;
                for (int kc = 0; kc < s.ksz; ++kc) colm |= ((unsigned)(iw0 + kc) < (unsigned)s.W ? 1u : 0u) << kc;
                for (int kr = 0; kr < s.ksz; ++kr)
                    if ((unsigned)(ih0 + kr) < (unsigned)s.H) m |= (unsigned long long)colm << (kr * s.ksz);
            }
            bbase = (n * (unsigned)s.C + (unsigned)(g * s.Cg)) * (unsigned)s.HW + (unsigned)(ih0 * s.W + iw0);
            obase = (n * (unsigned)s.F + (unsigned)(g * s.Mg)) * (unsigned)s.OHOW + pix;
        } else {
            const int ih = c_ih0 + (int)u * s.stride, iw = c_iw0 + (int)v * s.stride;
            const int qa = (ih + s.pad) / s.stride, qb = (iw + s.pad) / s.stride;  // exact for valid taps
            for (int t = 0; need_mask && t < ntaps; ++t) {
                const int kr = ctaps[t] & 0xf, kc = ctaps[t] >> 4;
                const int oh = qa - kr / s.stride, ow = qb - kc / s.stride;
                if ((unsigned)oh < (unsigned)s.OH && (unsigned)ow < (unsigned)s.OW) m |= 1ULL << t;
            }
            bbase = (n * (unsigned)s.F + (unsigned)(g * s.Mg)) * (unsigned)s.OHOW + (unsigned)(qa * s.OW + qb);
            obase = (n * (unsigned)s.C + (unsigned)(g * s.Cg)) * (unsigned)s.HW + (unsigned)(ih * s.W + iw);
        }
        mask = m;
        return true;
    };

    if (!fwd && !s.pointwise) __syncthreads();  // ctaps visible before decode()

    // ---- this thread's staging column -------------------------------------------------------------
    const int bj = tid % BN, bk0 = tid / BN;
    unsigned b_base = 0, o_unused = 0;
    unsigned long long b_mask = 0;
    decode(p0 + bj, true, b_base, b_mask, o_unused);

    // ---- this thread's A rows ---------------------------------------------------------------------
    const int ak = tid % BK, am0 = tid / BK;
    const float* abase = a.a_base + (long long)g * a.a_group_stride;
    unsigned a_rowoff[A_IT];
    unsigned a_rowok = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + am0 + i * (256 / BK);
        const bool ok = m < a.M;
        a_rowoff[i] = ok ? (unsigned)m * (unsigned)a.a_row_stride : 0u;
        a_rowok |= (ok ? 1u : 0u) << i;
    }

    // table entry of reduction index r (one thread per entry, magic division)
    auto fill_ktab = [&](int kt, int slot) {
        if (tid < BK) {
            const int r = kt * BK + tid;
            int4 e = make_int4(0, 0, 63, 0);  // tap 63: never set in a mask
            if (r < KR) {
                const unsigned major = fast_div((unsigned)r, ntaps_magic);
                const unsigned tap = (unsigned)r - major * (unsigned)ntaps;
                e.w = 1;
                if (s.pointwise) {
                    e.x = fwd ? r : r * s.K;            // W[f][k] / W[f][c]: forward aoff = k; dX aoff = f*K (+ c via row stride)
                    e.y = r * s.OHOW;                   // k-th (f-th) row of the raw [.][OH*OW] view
                    e.z = 0;
                } else if (fwd) {
                    const unsigned kr = fast_div(tap, a.ksz_magic), kc = tap - kr * (unsigned)s.ksz;
                    e.x = r;                            // W[f][c*kk2 + tap], row stride K
                    e.y = (int)(major * (unsigned)s.HW + kr * (unsigned)s.W + kc);
                    e.z = (int)tap;
                } else {
                    const int kr = ctaps[tap] & 0xf, kc = ctaps[tap] >> 4;
                    e.x = (int)(major * (unsigned)s.K) + kr * s.ksz + kc;  // W[f][c][kr][kc], row (c) stride kk2
                    e.y = (int)(major * (unsigned)s.OHOW) - ((kr / s.stride) * s.OW + kc / s.stride);
                    e.z = (int)tap;
                }
            }
            ktab[slot][tid] = e;
        }
    };

    // Staging registers. Every global load is UNCONDITIONAL from a clamped (always legal) offset; validity
    // is applied when the value is written to LDS, so load_tile is straight-line code.
    float ra[A_IT], rb[B_IT];
    unsigned a_ok = 0, b_ok = 0;
    auto load_tile = [&](int slot) {
        {
            const int4 e = ktab[slot][ak];
            a_ok = e.w ? a_rowok : 0u;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = abase[e.w ? a_rowoff[i] + (unsigned)e.x : 0u];
        }
        int4 e[B_IT];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) e[i] = ktab[slot][bk0 + i * B_ROWS];
        b_ok = 0;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const bool ok = (b_mask >> e[i].z) & 1ULL;
            rb[i] = a.b_base[ok ? b_base + (unsigned)e[i].y : 0u];
            b_ok |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) As[buf][ak][am0 + i * (256 / BK)] = ((a_ok >> i) & 1u) ? ra[i] : 0.f;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) Bs[buf][bk0 + i * B_ROWS][bj] = ((b_ok >> i) & 1u) ? rb[i] : 0.f;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    if (nk > 0) {
        fill_ktab(0, 0);
        __syncthreads();
        load_tile(0);
        store_tile(0);
        if (nk > 1) fill_ktab(1, 1);
        __syncthreads();
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(cur ^ 1);  // global loads in flight under the MFMAs
        // always BK/2 steps: entries past the end of the reduction are zero in LDS (table `valid` = 0)
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
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

    // ---- epilogue ------------------------------------------------------------------------------------
    const unsigned o_row_stride = fwd ? (unsigned)s.OHOW : (s.pointwise ? (unsigned)s.OHOW : (unsigned)s.HW);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        unsigned bb, ob;
        unsigned long long mk;
        if (!decode(p0 + (wn * TN + j) * 32 + l31, false, bb, mk, ob)) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + mfma_row(r, lane);
                if (m >= a.M) continue;
                float v = acc[i][j][r];
                if (fwd) {
                    const int fc = g * s.Mg + m;
                    if (a.add_bias) {
                        const float b = a.bias[fc];
                        if (b != 0.0f && b != 1.0f) v += b;  // bcnn_add_scalar (AVX build) skips exactly 0 and 1
                    }
                    if (a.act != BCNN_HIP_ACT_NONE)
                        v = act_fwd_cheap(v, a.act, a.act == BCNN_HIP_ACT_PRELU ? a.slopes[fc] : 0.f);
                }
                a.out[(size_t)ob + (size_t)m * o_row_stride] = v;
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------
static unsigned magic_of(int d) { return d > 1 ? (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d) : 0u; }

template <int WM, int WN, int TM, int TN, int BK>
static void launch_igemm(IgemmArgs& a, long long max_cols) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    a.mtiles = ceil_div(a.M, BM);
    a.ptiles = ceil_div(max_cols, BN);
    dim3 grid((unsigned)(a.mtiles * a.ptiles), (unsigned)a.s.groups, (unsigned)a.nclass);
    conv_igemm_kernel<WM, WN, TM, TN, BK><<<grid, 256, 0, current_stream()>>>(a);
    KERNEL_CHECK();
}

static void dispatch_igemm(IgemmArgs& a, long long max_cols) {
    const long long big_tiles = (long long)ceil_div(a.M, 128) * ceil_div(max_cols, 128) * a.s.groups * a.nclass;
    if (a.M <= 32) launch_igemm<1, 4, 1, 1, 16>(a, max_cols);        // 32 x 128
    else if (a.M <= 64 || big_tiles < 2 * kCUs) launch_igemm<2, 2, 1, 2, 16>(a, max_cols);  // 64 x 128
    else launch_igemm<2, 2, 2, 2, 16>(a, max_cols);                  // 128 x 128
}

bool conv_forward_dma(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                      const ConvShape& s, int act, int raw, ConvStats* stats, const BnFold* fold = nullptr);        // conv_igemm_dma.hip
bool conv_backward_data_dma(const float* w, const float* dy, float* dx, const ConvShape& s, DxBnSums* bs);
bool conv_forward_small_c(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                          const ConvShape& s, int act, int raw, ConvStats* stats);   // conv_igemm_dma.hip

static bool dma_enabled() {
    static const int on = BCNN_EXP_ENV("BCNN_HIP_NO_DMA") ? 0 : 1;  // A/B switch for profiling
    return on != 0;
}

// raw = 1: write the bare convolution (no bias, no activation) -- used by the fused-BN path.
void conv_forward_dispatch(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                           const ConvShape& s, int act, int raw, ConvStats* stats) {
    if (stats) stats->splits = 0;
    if (s.total_q == 0 || s.Mg == 0) return;
    if (s.ksz > 7 && !s.pointwise) {
        fprintf(stderr, "[bcnn_hip] conv forward: kernel size %d > 7 is not supported\n", s.ksz);
        exit(1);
    }
    KTimer kt(K_CONV_FWD, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    if (dma_enabled() && conv_forward_dma(x, w, bias, slopes, y, s, act, raw, stats)) return;
    if (dma_enabled() && conv_forward_small_c(x, w, bias, slopes, y, s, act, raw, stats)) return;
    IgemmArgs a;
    a.a_base = w; a.b_base = x; a.out = y; a.bias = bias; a.slopes = slopes; a.s = s;
    a.mode = 0;
    a.act = raw ? BCNN_HIP_ACT_NONE : act;
    a.add_bias = raw ? 0 : 1;
    a.M = s.Mg; a.KR = s.K; a.a_row_stride = s.K; a.a_group_stride = (long long)s.Mg * s.K;
    a.ntaps = s.pointwise ? 1 : s.ksz * s.ksz;
    a.ntaps_magic = magic_of(a.ntaps);
    a.ksz_magic = magic_of(s.ksz);
    a.nclass = 1;
    dispatch_igemm(a, s.total_q);
}

// ------------------------------------------------------------------------------------------------
// dX for layers with very few input channels (K = C/g*k*k <= 32, e.g. the RGB layer of configs[1] when its source
// carries a gradient). As an implicit GEMM the output has only C/g rows, so a 32-row MFMA tile idles 29 of them
// (measured 6.9 TFLOP/s, 3.2 ms at configs[1]). The reference's own factorisation fits these shapes instead
// (bcnn_conv_layer.c:563-577): col[K][q] = W^T[K x F] * dy[F][q] -- a GEMM with M = K <= 32 rows, all useful --
// followed by col2im. col is produced for a chunk of images small enough to stay in the Infinity Cache and is
// consumed at once by a gather-form col2im (each dx element sums its <= k*k contributions in ascending (kr,kc)
// order, the order bcnn_col2im applies them in, bcnn_mat.c:935-970), so it never travels to HBM and back.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                int F, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // wt[k][f] = w[f][k]
    if (i < F * K) wt[(i % K) * F + (i / K)] = w[i];
}

__global__ __launch_bounds__(256) void col2im_batch_kernel(const float* __restrict__ col, float* __restrict__ dx,
                                                           int C, int H, int W, int ksz, int pad, int stride, int OH,
                                                           int OW, unsigned total) {
    const unsigned gs = gridDim.x * blockDim.x;
    const int K = C * ksz * ksz, OHOW = OH * OW;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int iw = (int)(i % (unsigned)W);
        unsigned t = i / (unsigned)W;
        const int ih = (int)(t % (unsigned)H);
        t /= (unsigned)H;
        const int c = (int)(t % (unsigned)C), n = (int)(t / (unsigned)C);
        const float* cn = col + ((size_t)n * K + (size_t)c * ksz * ksz) * OHOW;
        float acc = 0.f;
        for (int kr = 0; kr < ksz; ++kr) {
            const int th = ih + pad - kr;
            if (th < 0 || th % stride) continue;
            const int oh = th / stride;
            if (oh >= OH) continue;
            for (int kc = 0; kc < ksz; ++kc) {
                const int tw = iw + pad - kc;
                if (tw < 0 || tw % stride) continue;
                const int ow = tw / stride;
                if (ow >= OW) continue;
                acc += cn[(size_t)(kr * ksz + kc) * OHOW + oh * OW + ow];
            }
        }
        dx[i] = acc;
    }
}

struct ColScratch {
    float* p = nullptr;
    size_t cap = 0;
    int dev = -1;
};
static thread_local ColScratch g_col_scratch;

static float* col_scratch(size_t floats) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    ColScratch& sc = g_col_scratch;
    if (sc.p == nullptr || sc.cap < floats || sc.dev != dev) {
        if (sc.p && sc.dev == dev) HIP_CHECK(hipFree(sc.p));  // hipFree synchronises the device
        HIP_CHECK(hipMalloc((void**)&sc.p, floats * sizeof(float)));
        sc.cap = floats; sc.dev = dev;
    }
    return sc.p;
}

static bool conv_backward_data_small_c(const float* w, const float* dy, float* dx, const ConvShape& s) {
    static const bool off = BCNN_EXP_ENV("BCNN_HIP_NO_SMALLC_DX") != nullptr;
    if (off || s.pointwise || s.groups != 1 || s.K > 32 || s.Mg < 32) return false;
    if ((long long)s.N * s.C * s.HW >= (1LL << 31)) return false;
    const size_t per_image = (size_t)s.K * s.OHOW;                      // col floats per image
    int chunk = (int)(((size_t)96 << 20) / (per_image * sizeof(float)));  // <= 96 MB of col in flight
    if (chunk < 1) chunk = 1;
    if (chunk > s.N) chunk = s.N;
    float* scratch = col_scratch((size_t)chunk * per_image + (size_t)s.K * s.Mg);
    float* wt = scratch + (size_t)chunk * per_image;
    transpose_weights_kernel<<<ceil_div(s.Mg * s.K, 256), 256, 0, current_stream()>>>(w, wt, s.Mg, s.K);
    KERNEL_CHECK();
    for (int n0 = 0; n0 < s.N; n0 += chunk) {
        const int nb = (s.N - n0 < chunk) ? s.N - n0 : chunk;
        // col = Wt * dy as a bare 1x1 "forward" over the chunk: source dy [nb][F][OH*OW], K output rows
        const ConvShape g = make_conv_shape(nb, s.F, s.OH, s.OW, s.K, 1, 1, 0, 1);
        IgemmArgs a;
        a.a_base = wt; a.b_base = dy + (size_t)n0 * s.F * s.OHOW; a.out = scratch; a.bias = nullptr; a.slopes = nullptr;
        a.s = g; a.mode = 0; a.act = BCNN_HIP_ACT_NONE; a.add_bias = 0;
        a.M = g.Mg; a.KR = g.K; a.a_row_stride = g.K; a.a_group_stride = (long long)g.Mg * g.K;
        a.ntaps = 1; a.ntaps_magic = magic_of(1); a.ksz_magic = magic_of(1); a.nclass = 1;
        dispatch_igemm(a, g.total_q);
        const long long total = (long long)nb * s.C * s.HW;
        col2im_batch_kernel<<<stream_grid((size_t)total, 256), 256, 0, current_stream()>>>(
            scratch, dx + (size_t)n0 * s.C * s.HW, s.C, s.H, s.W, s.ksz, s.pad, s.stride, s.OH, s.OW, (unsigned)total);
        KERNEL_CHECK();
    }
    return true;
}

void conv_backward_data(const float* w, const float* dy, float* dx, const ConvShape& s, DxBnSums* bs) {
    if (bs) bs->splits = 0;
    if (s.total_p == 0 || s.Cg == 0) return;
    if (s.ksz > 7 && !s.pointwise) {
        fprintf(stderr, "[bcnn_hip] conv backward: kernel size %d > 7 is not supported\n", s.ksz);
        exit(1);
    }
    KTimer kt(K_CONV_DX, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    if (conv_backward_data_small_c(w, dy, dx, s)) return;
    if (dma_enabled() && conv_backward_data_dma(w, dy, dx, s, bs)) return;
    IgemmArgs a;
    a.a_base = w; a.b_base = dy; a.out = dx; a.bias = nullptr; a.slopes = nullptr; a.s = s;
    a.mode = 1; a.act = BCNN_HIP_ACT_NONE; a.add_bias = 0;
    a.M = s.Cg; a.a_group_stride = (long long)s.Mg * s.K;
    a.ksz_magic = magic_of(s.ksz);
    if (s.pointwise) {
        a.KR = s.Mg; a.a_row_stride = 1;  // W[f][c]: A(c, f) = w[f*K + c]
        a.ntaps = 1; a.ntaps_magic = magic_of(1); a.nclass = 1;
        dispatch_igemm(a, s.total_q);
        return;
    }
    a.a_row_stride = s.ksz * s.ksz;
    a.ntaps = 0; a.ntaps_magic = 0; a.KR = 0;
    // stride-parity classes, up to kMaxClassesPerLaunch per launch
    const int st = s.stride;
    int nc = 0;
    long long max_cols = 0;
    for (int ra = 0; ra < st; ++ra)
        for (int rb = 0; rb < st; ++rb) {
            ClassInfo& ci = a.cls[nc];
            ci.ih0 = ((ra - s.pad) % st + st) % st;  // first row with (ih + pad) % st == ra
            ci.iw0 = ((rb - s.pad) % st + st) % st;
            ci.Hc = ci.ih0 < s.H ? (s.H - ci.ih0 + st - 1) / st : 0;
            ci.Wc = ci.iw0 < s.W ? (s.W - ci.iw0 + st - 1) / st : 0;
            ci.ntaps = 0;
            for (int kr = 0; kr < s.ksz; ++kr)
                for (int kc = 0; kc < s.ksz; ++kc)
                    if (kr % st == ra && kc % st == rb) ci.taps[ci.ntaps++] = (unsigned char)(kr | (kc << 4));
            const long long cols = (long long)s.N * ci.Hc * ci.Wc;
            if (cols > max_cols) max_cols = cols;
            if (++nc == kMaxClassesPerLaunch || (ra == st - 1 && rb == st - 1)) {
                a.nclass = nc;
                if (max_cols > 0) dispatch_igemm(a, max_cols);
                nc = 0;
                max_cols = 0;
            }
        }
}

}  // namespace bcnn_hip
