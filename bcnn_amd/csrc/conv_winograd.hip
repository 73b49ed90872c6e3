// conv_winograd.hip -- Winograd F(2x2, 3x3) convolution for 3x3 / stride 1 / pad 1 / one-group layers.
//
// Reference: the PREDICT-mode path of bcnn_forward_conv_layer_cpu (src/layers/bcnn_conv_layer.c:388-436) on the
// kernels of src/kernels/bcnn_mat.c:1403-2138 (weight transform G g G^T, input transform B^T d B over 4x4 patches
// with stride 2, 16 element-wise-position GEMMs, output transform A^T m A). The reference keeps its operands in an
// NC4HW4 layout for 4-wide SIMD; that layout is a CPU artefact and is not reproduced -- the arithmetic is:
//
//     V[xi][c][t]  = (B^T d_{c,t} B)[xi]           t = (n, th, tw): 2x2 output tile, d = 4x4 input patch at (2th-1, 2tw-1)
//     U[xi][f][c]  = (G  g_{f,c} G^T)[xi]          xi = 4*i + j in 0..15
//     M[xi][f][t]  = sum_c U[xi][f][c] * V[xi][c][t]              16 independent GEMMs  [F x C] x [C x T]
//     y[n][f][2th+a][2tw+b] = (A^T M[.][f][t] A)[a][b] + bias[f]  -> activation
//
// 16 multiplies per 2x2 outputs and channel pair instead of 36: 2.25x fewer MACs on the fp32 matrix pipe, which is
// what bounds every 3x3 layer of the benchmark.
//
// MI355X mapping: the 16 GEMMs are ONE launch of the LDS-DMA implicit-GEMM kernel (conv_igemm_dma.hip) on the shape
// "1x1 convolution, 16 groups, one image of T pixels": V is exactly a [16*C][T] raw matrix, U the grouped weights,
// M the grouped output -- no vector-ALU work in the MFMA loop, as for every other layer. The transforms are three
// streaming kernels (16-byte-free but fully coalesced along t). V and M live in a library scratch that stays in the
// 256 MB Infinity Cache for the 14x14 / 7x7 stages, which is where the path is enabled (wino_profitable below).
// The same three steps give dX (a stride-1 3x3 convolution of dY with the 180-degree rotated, transposed filter).
#include "conv_common.h"

namespace bcnn_hip {

bool conv_forward_dma(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                      const ConvShape& s, int act, int raw, ConvStats* stats, const BnFold* fold = nullptr);  // conv_igemm_dma.hip
size_t conv_dw_dma_workspace_floats(const ConvShape& s);                        // conv_dw_dma.hip
bool conv_backward_weights_dma(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                               size_t workspace_floats, const BnFold* fold = nullptr);

// ---- transforms --------------------------------------------------------------------------------------------
// B^T d B for one 4x4 patch (rows first, then columns): 32 additions.
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
__device__ __forceinline__ void wino_input_4x4(const float (&d)[4][4], float (&v)[16]) {
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[4 * i + 0] = t[i][0] - t[i][2];
        v[4 * i + 1] = t[i][1] + t[i][2];
        v[4 * i + 2] = t[i][2] - t[i][1];
        v[4 * i + 3] = t[i][1] - t[i][3];
    }
}

struct WinoGeom {
    int N, C, H, W;      // tensor being transformed (x for forward, dy for dX)
    int TH, TW;          // 2x2 tiles per image: ceil(H/2) x ceil(W/2) (output extent == input extent: stride 1, pad 1)
    unsigned T;          // N * TH * TW
};

// V[xi][c][t]: one thread per (c, t); lanes run along t so every one of the 16 stores is a coalesced run.
__global__ __launch_bounds__(256) void wino_input_transform_kernel(const float* __restrict__ x, float* __restrict__ v,
                                                                   const WinoGeom g) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= g.T) return;
    const unsigned per_img = (unsigned)(g.TH * g.TW);
    const unsigned n = t / per_img, r = t - n * per_img;
    const int th = (int)(r / (unsigned)g.TW), tw = (int)(r - (unsigned)th * (unsigned)g.TW);
    const float* p = x + ((size_t)n * g.C + c) * (size_t)(g.H * g.W);
    const int ih0 = 2 * th - 1, iw0 = 2 * tw - 1;
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ih = ih0 + i;
        const bool rok = (unsigned)ih < (unsigned)g.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = iw0 + j;
            d[i][j] = (rok && (unsigned)iw < (unsigned)g.W) ? p[ih * g.W + iw] : 0.f;
        }
    }
    float o[16];
    wino_input_4x4(d, o);
    float* dst = v + (size_t)c * g.T + t;
    const size_t plane = (size_t)g.C * g.T;
#pragma unroll
    for (int k = 0; k < 16; ++k) dst[(size_t)k * plane] = o[k];
}

// U[xi][m][j] = (G g G^T)[xi], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].
//   forward: m = f, j = c, g = w[f][c]            dX: m = c, j = f, g = w[f][c] rotated by 180 degrees
__global__ __launch_bounds__(256) void wino_weight_transform_kernel(const float* __restrict__ w, float* __restrict__ u,
                                                                    int F, int C, int dx_mode) {
    const int M = dx_mode ? C : F, J = dx_mode ? F : C;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * J) return;
    const int m = idx / J, j = idx - m * J;
    const int f = dx_mode ? j : m, c = dx_mode ? m : j;
    const float* p = w + ((size_t)f * C + c) * 9;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = dx_mode ? p[(2 - a) * 3 + (2 - b)] : p[a * 3 + b];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const size_t plane = (size_t)M * J;
    float* dst = u + (size_t)m * J + j;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        dst[(size_t)(4 * a + 0) * plane] = t[a][0];
        dst[(size_t)(4 * a + 1) * plane] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        dst[(size_t)(4 * a + 2) * plane] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        dst[(size_t)(4 * a + 3) * plane] = t[a][2];
    }
}

// y = A^T m A (+ bias, activation), A^T = [1 1 1 0; 0 1 -1 -1]. One thread per (f, t); 16 coalesced loads along t.
// stats != nullptr (raw output feeding a fused batch-norm): the workgroup also publishes the sum and the sum of
// squares of the values it stores, stats[(f * gridDim.x + blockIdx.x) * 2 + {0, 1}] -- the layout
// bn_stats_finalize consumes (ConvStats), so the batch-norm needs no statistics pass of its own.
template <bool PLAIN>
__global__ __launch_bounds__(256) void wino_output_transform_kernel(const float* __restrict__ mm, float* __restrict__ y,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ slopes, int act,
                                                                    const WinoGeom g, int F, float* __restrict__ stats) {
    __shared__ float red[2][4];
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const int f = blockIdx.y;
    const bool live = t < g.T;
    if (!live && stats == nullptr) return;
    const size_t plane = (size_t)F * g.T;
    const float* src = mm + (size_t)f * g.T + (live ? t : 0);
    float m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = live ? src[(size_t)k * plane] : 0.f;
    float s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = m[j] + m[4 + j] + m[8 + j];
        s[1][j] = m[4 + j] - m[8 + j] - m[12 + j];
    }
    float o[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        o[a][0] = s[a][0] + s[a][1] + s[a][2];
        o[a][1] = s[a][1] - s[a][2] - s[a][3];
    }
    if (!PLAIN) {
        float b = bias ? bias[f] : 0.f;
        if (b == 1.0f) b = 0.f;  // bcnn_add_scalar of the AVX build adds nothing for exactly 1.0f (quirk 2)
        const float sl = (act == BCNN_HIP_ACT_PRELU && slopes) ? slopes[f] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float vv = o[a][c];
                if (b != 0.0f) vv += b;
                if (act != BCNN_HIP_ACT_NONE) vv = act_fwd_cheap(vv, act, sl);
                o[a][c] = vv;
            }
    }
    const unsigned per_img = (unsigned)(g.TH * g.TW);
    const unsigned n = t / per_img, r = t - n * per_img;
    const int th = (int)(r / (unsigned)g.TW), tw = (int)(r - (unsigned)th * (unsigned)g.TW);
    float* dst = y + ((size_t)n * F + f) * (size_t)(g.H * g.W);
    const int oh = 2 * th, ow = 2 * tw;
    const bool two_cols = ow + 1 < g.W;
    float sv = 0.f, sq = 0.f;
    if (live) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (oh + a >= g.H) break;
            float* row = dst + (oh + a) * g.W + ow;
            row[0] = o[a][0];
            sv += o[a][0];
            sq += o[a][0] * o[a][0];
            if (two_cols) {
                row[1] = o[a][1];
                sv += o[a][1];
                sq += o[a][1] * o[a][1];
            }
        }
    }
    if (stats != nullptr) {  // fixed-order reduction: 64-lane shuffle tree, then the four waves in order
        sv = wave_sum(sv);
        sq = wave_sum(sq);
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        if (lane == 0) { red[0][wid] = sv; red[1][wid] = sq; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* d = stats + ((size_t)f * gridDim.x + blockIdx.x) * 2;
            d[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            d[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// Weight gradient in the transformed domain:  dU[xi][f][c] = sum_t dM[xi][f][t] * V[xi][c][t],
//   dM = A dy_tile A^T (the adjoint of the output transform, A = [1 0; 1 1; 1 -1; 0 -1]),  dw[f][c] += G^T dU[.][f][c] G.
__global__ __launch_bounds__(256) void wino_dy_transform_kernel(const float* __restrict__ dy, float* __restrict__ dm,
                                                                const WinoGeom g) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const int f = blockIdx.y;
    if (t >= g.T) return;
    const unsigned per_img = (unsigned)(g.TH * g.TW);
    const unsigned n = t / per_img, r = t - n * per_img;
    const int th = (int)(r / (unsigned)g.TW), tw = (int)(r - (unsigned)th * (unsigned)g.TW);
    const float* p = dy + ((size_t)n * g.C + f) * (size_t)(g.H * g.W);
    const int oh = 2 * th, ow = 2 * tw;
    float d[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) d[a][b] = (oh + a < g.H && ow + b < g.W) ? p[(oh + a) * g.W + ow + b] : 0.f;
    float q[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        q[0][b] = d[0][b];
        q[1][b] = d[0][b] + d[1][b];
        q[2][b] = d[0][b] - d[1][b];
        q[3][b] = -d[1][b];
    }
    float* dst = dm + (size_t)f * g.T + t;
    const size_t plane = (size_t)g.C * g.T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[(size_t)(4 * i + 0) * plane] = q[i][0];
        dst[(size_t)(4 * i + 1) * plane] = q[i][0] + q[i][1];
        dst[(size_t)(4 * i + 2) * plane] = q[i][0] - q[i][1];
        dst[(size_t)(4 * i + 3) * plane] = -q[i][1];
    }
}

// dw[f][c][3][3] += G^T dU[.][f][c] G,  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]  (beta = 1: onto the momentum carry)
__global__ __launch_bounds__(256) void wino_dw_finish_kernel(const float* __restrict__ du, float* __restrict__ dw, int F,
                                                             int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= F * C) return;
    const size_t plane = (size_t)F * C;
    float u[4][4];
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k >> 2][k & 3] = du[(size_t)k * plane + idx];
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        t[1][j] = 0.5f * (u[1][j] - u[2][j]);
        t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
    float* out = dw + (size_t)idx * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        out[a * 3 + 0] += t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        out[a * 3 + 1] += 0.5f * (t[a][1] - t[a][2]);
        out[a * 3 + 2] += 0.5f * (t[a][1] + t[a][2]) + t[a][3];
    }
}

// ---- scratch: V, M and U of the layer in flight (grow-only; sized by the first step, Infinity-Cache resident
// for the shapes wino_profitable admits) ----------------------------------------------------------------------
struct WinoScratch {
    float* p = nullptr;
    size_t cap = 0;
    int dev = -1;
};
// two blocks: [0] forward / data gradient, [1] weight gradient -- the weight gradient may run on a side stream next to
// another layer's data gradient (bcnn_hip_conv_side_stream_mode)
static thread_local WinoScratch g_wino_scratch[2];

static float* wino_scratch(size_t floats, int which = 0) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    WinoScratch& sc = g_wino_scratch[which];
    if (sc.p == nullptr || sc.cap < floats || sc.dev != dev) {
        if (sc.p && sc.dev == dev) {
            HIP_CHECK(hipStreamSynchronize(current_stream()));  // launches still reading the old block
            HIP_CHECK(hipFree(sc.p));
        }
        HIP_CHECK(hipMalloc((void**)&sc.p, floats * sizeof(float)));
        sc.cap = floats;
        sc.dev = dev;
    }
    return sc.p;
}

// Shapes the path takes at all (the algorithm needs 3x3 / s1 / p1 / one group).
static bool wino_applicable(const ConvShape& s) {
    if (s.ksz != 3 || s.stride != 1 || s.pad != 1 || s.groups != 1) return false;
    if (s.C < 16 || s.F < 64) return false;  // grouped-GEMM preconditions of the LDS-DMA kernel (M > 32, J >= 8)
    const long long T = (long long)s.N * ((s.H + 1) / 2) * ((s.W + 1) / 2);
    const long long big = (long long)(s.C > s.F ? s.C : s.F) * 16 * T;
    return T > 0 && big * 4 < 0x7ffffff0LL;  // V / M addressed through one < 2 GiB buffer descriptor
}

// ... and the ones where it is faster than the direct LDS-DMA kernel. The three-kernel form moves V and M
// (16/4 = 4x the activation each) through the memory system once more, so it pays where those stay on chip and
// the GEMM is deep: measured on the ResNet-18 shapes at N = 128 (tools/exp/wino_sweep.sh).
static int g_wino_force = -1;  // experiment build: BCNN_HIP_WINOGRAD=0/1 overrides the rule
static bool wino_profitable(const ConvShape& s) {
    if (g_wino_force < 0) {
        const char* e = BCNN_EXP_ENV("BCNN_HIP_WINOGRAD");
        g_wino_force = e ? (e[0] == '0' ? 0 : 1) : 2;
    }
    if (g_wino_force != 2) return g_wino_force == 1;
    const double T = (double)s.N * ((s.H + 1) / 2) * ((s.W + 1) / 2);
    const double vm_bytes = 16.0 * T * (s.C + s.F) * 4.0;
    return s.C >= 128 && s.F >= 128 && vm_bytes <= 230e6;
}

static void wino_run(const float* src, const float* w, float* dst, const ConvShape& s, int dx_mode, const float* bias,
                     const float* slopes, int act, bool plain, ConvStats* stats = nullptr) {
    // src: x [N][C][H][W] (forward) or dy [N][F][H][W] (dX); J = reduction channels, M = produced channels
    const int J = dx_mode ? s.F : s.C, M = dx_mode ? s.C : s.F;
    WinoGeom g;
    g.N = s.N; g.C = J; g.H = s.H; g.W = s.W;
    g.TH = (s.H + 1) / 2; g.TW = (s.W + 1) / 2;
    g.T = (unsigned)((long long)s.N * g.TH * g.TW);
    const size_t v_floats = (size_t)16 * J * g.T, m_floats = (size_t)16 * M * g.T, u_floats = (size_t)16 * M * J;
    float* V = wino_scratch(v_floats + m_floats + u_floats);
    float* Mm = V + v_floats;
    float* U = Mm + m_floats;
    wino_weight_transform_kernel<<<ceil_div((long long)M * J, 256), 256, 0, current_stream()>>>(w, U, s.F, s.C, dx_mode);
    KERNEL_CHECK();
    dim3 gi((unsigned)ceil_div(g.T, 256), (unsigned)J);
    wino_input_transform_kernel<<<gi, 256, 0, current_stream()>>>(src, V, g);
    KERNEL_CHECK();
    // 16 GEMMs [M x J] x [J x T] as one grouped 1x1 convolution over a single "image" of T pixels
    const ConvShape gs = make_conv_shape(1, 16 * J, 1, (int)g.T, 16 * M, 1, 1, 0, 16);
    if (!conv_forward_dma(V, U, nullptr, nullptr, Mm, gs, BCNN_HIP_ACT_NONE, /*raw=*/1, nullptr)) {
        fprintf(stderr, "[bcnn_hip] winograd: grouped GEMM shape rejected (J=%d M=%d T=%u)\n", J, M, g.T);
        exit(1);
    }
    WinoGeom go = g;
    go.C = M;
    dim3 gout((unsigned)ceil_div(g.T, 256), (unsigned)M);
    float* st = (stats && stats->partials && plain) ? stats->partials : nullptr;
    if (plain)
        wino_output_transform_kernel<true><<<gout, 256, 0, current_stream()>>>(Mm, dst, nullptr, nullptr, 0, go, M, st);
    else
        wino_output_transform_kernel<false><<<gout, 256, 0, current_stream()>>>(Mm, dst, bias, slopes, act, go, M, nullptr);
    KERNEL_CHECK();
    if (stats) stats->splits = st ? (int)gout.x : 0;
}

// Algorithmic figures of the class timers: FLOPs the MFMAs really execute (transformed domain, 16 instead of 36
// multiplies per 2x2 outputs), bytes = the layer's tensors once (same as the direct kernels).
static double wino_flops(const ConvShape& s) {
    const double T = (double)s.N * ((s.H + 1) / 2) * ((s.W + 1) / 2);
    return 2.0 * 16.0 * T * s.C * s.F;
}
// the same without the tiles' overhang on odd-sized planes (7 x 7: 16 tiles cover 8 x 8)
static double wino_useful_flops(const ConvShape& s) { return 2.0 * 16.0 * ((double)s.N * s.H * s.W / 4.0) * s.C * s.F; }
static double wino_bytes(const ConvShape& s) {
    return 4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW);
}

// Returns false when the layer stays on the direct kernels.
bool conv_winograd_unfused_takes(const ConvShape& s) { return wino_applicable(s) && wino_profitable(s); }

bool conv_forward_winograd(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                           const ConvShape& s, int act, int raw, ConvStats* stats) {
    if (!wino_applicable(s) || !wino_profitable(s)) return false;
    KTimer kt(K_CONV_FWD_WINO, wino_flops(s), wino_bytes(s), wino_useful_flops(s));
    const bool plain = raw || (bias == nullptr && act == BCNN_HIP_ACT_NONE);
    // raw output for a fused batch-norm: the output transform also emits the per-channel statistics partials
    // (ceil(T / 256) <= ceil(N*OH*OW / 64) entries per channel: inside the buffer conv.hip sized)
    wino_run(x, w, y, s, /*dx_mode=*/0, bias, slopes, raw ? BCNN_HIP_ACT_NONE : act, plain, raw ? stats : nullptr);
    if (stats && !raw) stats->splits = 0;
    return true;
}

bool conv_backward_data_winograd(const float* w, const float* dy, float* dx, const ConvShape& s) {
    if (!wino_applicable(s) || !wino_profitable(s)) return false;
    // dX is itself a 3x3 / s1 / p1 convolution of dy [N][F][H][W] with F and C swapped
    if (s.F < 16 || s.C < 64) return false;
    KTimer kt(K_CONV_DX_WINO, wino_flops(s), wino_bytes(s), wino_useful_flops(s));
    wino_run(dy, w, dx, s, /*dx_mode=*/1, nullptr, nullptr, BCNN_HIP_ACT_NONE, true);
    return true;
}

// ---- weight gradient -----------------------------------------------------------------------------------------
static ConvShape wino_dw_gemm_shape(const ConvShape& s) {
    const int T = (int)((long long)s.N * ((s.H + 1) / 2) * ((s.W + 1) / 2));
    return make_conv_shape(1, 16 * s.C, 1, T, 16 * s.F, 1, 1, 0, 16);
}

static bool wino_dw_applicable(const ConvShape& s) {
    return wino_applicable(s) && wino_profitable(s) && s.C >= 16 && !(s.C & 1) && !(s.F & 1);
}

// split partials of the 16 grouped GEMMs (the caller's conv workspace, bcnn_hip_conv_workspace_size)
size_t conv_dw_winograd_workspace_floats(const ConvShape& s) {
    if (!wino_dw_applicable(s)) return 0;
    return conv_dw_dma_workspace_floats(wino_dw_gemm_shape(s));
}

bool conv_backward_weights_winograd(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                    size_t workspace_floats) {
    if (!wino_dw_applicable(s)) return false;
    const ConvShape gs = wino_dw_gemm_shape(s);
    const size_t need = conv_dw_dma_workspace_floats(gs);
    if (need == 0) return false;
    KTimer kt(K_CONV_DW_WINO, wino_flops(s), wino_bytes(s), wino_useful_flops(s));
    WinoGeom g;
    g.N = s.N; g.C = s.C; g.H = s.H; g.W = s.W;
    g.TH = (s.H + 1) / 2; g.TW = (s.W + 1) / 2;
    g.T = (unsigned)((long long)s.N * g.TH * g.TW);
    const size_t v_floats = (size_t)16 * s.C * g.T, m_floats = (size_t)16 * s.F * g.T, u_floats = (size_t)16 * s.F * s.C;
    float* V = wino_scratch(v_floats + m_floats + u_floats, 1);
    float* dM = V + v_floats;
    float* dU = dM + m_floats;
    dim3 gi((unsigned)ceil_div(g.T, 256), (unsigned)s.C);
    wino_input_transform_kernel<<<gi, 256, 0, current_stream()>>>(x, V, g);
    KERNEL_CHECK();
    WinoGeom gy = g;
    gy.C = s.F;
    dim3 gd((unsigned)ceil_div(g.T, 256), (unsigned)s.F);
    wino_dy_transform_kernel<<<gd, 256, 0, current_stream()>>>(dy, dM, gy);
    KERNEL_CHECK();
    HIP_CHECK(hipMemsetAsync(dU, 0, u_floats * sizeof(float), current_stream()));
    if (!conv_backward_weights_dma(V, dM, dU, gs, workspace, workspace_floats)) {
        fprintf(stderr, "[bcnn_hip] winograd dW: grouped GEMM shape rejected (C=%d F=%d T=%u)\n", s.C, s.F, g.T);
        exit(1);
    }
    wino_dw_finish_kernel<<<ceil_div((long long)s.F * s.C, 256), 256, 0, current_stream()>>>(dU, dw, s.F, s.C);
    KERNEL_CHECK();
    return true;
}

}  // namespace bcnn_hip
