// batchnorm.hip -- batch normalisation forward/backward (HBM-bound streaming + wave64 reductions).
//
// Reference semantics: src/layers/bcnn_batchnorm_layer.c:196-242 (forward), :301-332 (backward):
//   mean = sum(x)/M, var = sum(x^2)/M - mean^2 (biased, one pass), M = N*H*W;
//   running = 0.9*running + 0.1*batch; y = ((x-mean)/sqrtf(var+1e-6)) * scale + bias;
//   backward uses eps = 1e-5 and writes the input gradient over dy in place.
// Passes over the activation tensor: TRAIN forward = 1 read (statistics) + 1 read + 1 write (apply,
// with the activation fused); backward = 1 pass of reads (sums) + 1 read/write pass (apply). The
// reference CPU path makes ~10 forward sweeps (two copies, x_norm, scale and bias passes).
#include "chan_reduce.h"
#include "conv_common.h"
#include "bn_math.h"

namespace bcnn_hip {

#ifdef NT_STORES   // experiment: streaming stores for the sweeps' results
#define BN_ST4(p, v) do { typedef float f4_ __attribute__((ext_vector_type(4))); const f4_ t_ = {(v).x, (v).y, (v).z, (v).w}; \
                          __builtin_nontemporal_store(t_, reinterpret_cast<f4_*>(p)); } while (0)
#else
#define BN_ST4(p, v) (*reinterpret_cast<float4*>(p) = (v))
#endif


// ---- per-channel constants of an apply sweep ---------------------------------------------------------------------
// The apply bodies need, per channel, sqrt(var + eps), its correctly rounded reciprocal, dmean / M ...: two square roots
// and up to four IEEE divisions. Evaluated inside the map kernels (once per 16 bytes in the flat kernel, whose lanes
// change channel every step) that was more vector-ALU work than the sweep's own arithmetic. The finalize kernel that
// runs right before the sweep -- one thread per channel has mean, var, dmean, dvar in hand -- now leaves them in a small
// table (same operations, same roundings: bit-identical results), and the bodies load three float4 per call.
//   forward  [2c]   = {mean, rs6, 1 / rs6, scale}   [2c + 1] = {bias, -, -, -}                  rs6 = sqrt(var + 1e-6)
//   backward [3c]   = {mean, scale, rs5, 1 / rs5}   [3c + 1] = {rs6, 1 / rs6, fwd bias, dmean / M}   [3c + 2] = {dvar, -, -, -}
struct ConstScratch {
    float4* p = nullptr;
    size_t cap = 0;
    int dev = -1;
};
// One table per host thread AND device (a thread that alternates devices keeps each device's table). A thread drives
// one stream at a time (runtime.hip: the current stream is per thread), and finalize -> apply run back to back on it: two nets
// on different streams need two host threads, like every per-thread scratch of this library.
constexpr int kMaxDevices = 64;
static thread_local ConstScratch g_bn_consts[kMaxDevices];
float4* bn_consts_scratch(int channels, bool required);
float4* bn_consts_scratch(int channels, bool required) {  // grow-only: finalize -> apply only
    if (!required && BCNN_EXP_ENV("BCNN_HIP_BN_NO_CONSTS")) return nullptr;  // A/B switch (experiment build): constants evaluated in the bodies
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) { fprintf(stderr, "[bcnn_hip] device ordinal %d out of range\n", dev); exit(1); }
    ConstScratch& sc = g_bn_consts[dev];
    const size_t need = (size_t)channels * 3;
    if (sc.p == nullptr || sc.cap < need) {
        if (sc.p) HIP_CHECK(hipFree(sc.p));  // hipFree syncs the device
        const size_t cap = need < 4096 ? 4096 : need * 2;
        HIP_CHECK(hipMalloc((void**)&sc.p, cap * sizeof(float4)));
        sc.cap = cap;
        sc.dev = dev;
    }
    return sc.p;
}
__device__ __forceinline__ void bn_fwd_consts_store(float4* consts, int c, float mean, float var, const float* scale,
                                                    const float* bias) {
    if (!consts) return;
    const BnDiv rs = bn_divisor(sqrtf(var + 0.000001f));
    consts[2 * c] = make_float4(mean, rs.d, rs.r, scale[c]);
    consts[2 * c + 1] = make_float4(bias[c], 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void bn_bwd_consts_store(float4* consts, int c, const float* mean, float var, float sc,
                                                    const float* fwd_bias, float dmean, float dvar, float fM) {
    if (!consts) return;
    const BnDiv rs5 = bn_divisor(sqrtf(var + 0.00001f)), rs6 = bn_divisor(sqrtf(var + 0.000001f));
    consts[3 * c] = make_float4(mean[c], sc, rs5.d, rs5.r);
    consts[3 * c + 1] = make_float4(rs6.d, rs6.r, fwd_bias ? fwd_bias[c] : 0.f, __fdiv_rn(dmean, fM));
    consts[3 * c + 2] = make_float4(dvar, 0.f, 0.f, 0.f);
}

// ---- forward statistics --------------------------------------------------------------------------
struct StatsF {
    static constexpr int kInFlight = 4;  // chan_reduce_partial's unroll
    const float* x;
    __device__ void operator()(long long off, int, float (&acc)[2]) const {
        const float v = x[off];
        acc[0] += v;
        acc[1] += v * v;
    }
    __device__ void vec4(long long off, int, float (&acc)[2]) const {
        const float4 v = *reinterpret_cast<const float4*>(x + off);
        acc[0] += (v.x + v.y) + (v.z + v.w);
        acc[1] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
};

__global__ void bn_stats_finalize_kernel(const float* __restrict__ partials, int C, int splits, int M,
                                         float* __restrict__ saved_mean, float* __restrict__ saved_var,
                                         float* __restrict__ run_mean, float* __restrict__ run_var,
                                         float4* __restrict__ consts, const float* __restrict__ scale,
                                         const float* __restrict__ bias, const float* __restrict__ mean_shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int i = 0; i < splits; ++i) {
        s += (double)partials[((long long)c * splits + i) * 2 + 0];
        ss += (double)partials[((long long)c * splits + i) * 2 + 1];
    }
    const float inv = 1.0f / (float)M;
    const float mean = __fmul_rn((float)s, inv);                                        // bcnn_scal(c, scale, mean)
    const float var = __fsub_rn(__fmul_rn((float)ss, inv), __fmul_rn(mean, mean));      // bcnn_varmean
    saved_mean[c] = mean;
    saved_var[c] = var;
    // mean_shift (BnFold, conv_common.h): a per-channel constant the producer left out of x; the RUNNING mean is the one
    // place where it shows (saved_mean stays in the frame of the stored x, which is what apply and backward pair it with)
    const float vis = mean_shift ? __fadd_rn(mean, mean_shift[c]) : mean;
    run_mean[c] = __fadd_rn(__fmul_rn(vis, 0.1f), __fmul_rn(run_mean[c], 0.9f));        // scal 0.9, axpy 0.1
    run_var[c] = __fadd_rn(__fmul_rn(var, 0.1f), __fmul_rn(run_var[c], 0.9f));
    bn_fwd_consts_store(consts, c, mean, var, scale, bias);
}

// Same result for MANY partials per channel (the convolution epilogue emits one per column tile, up to a
// few thousand): one workgroup per channel, threads stride over the partials in double, fixed-shape
// butterfly + a fixed-order cross-wave step.
constexpr int kFinalizeThreads = 1024;
__global__ __launch_bounds__(kFinalizeThreads) void bn_stats_finalize_wide_kernel(
    const float* __restrict__ partials, int C, int splits, int M, float* __restrict__ saved_mean,
    float* __restrict__ saved_var, float* __restrict__ run_mean, float* __restrict__ run_var,
    float4* __restrict__ consts, const float* __restrict__ scale, const float* __restrict__ bias,
    const float* __restrict__ mean_shift) {
    constexpr int NW = kFinalizeThreads / 64;
    __shared__ double red[NW][2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const float2* p = reinterpret_cast<const float2*>(partials) + (long long)c * splits;
    double s = 0.0, ss = 0.0;
    // four independent loads per round: with one, a channel of 50 176 partials (the first pointwise layer of
    // MobileNet) was a chain of 196 dependent round trips per thread = 80 us
    int i = threadIdx.x;
    for (; i + 3 * kFinalizeThreads < splits; i += 4 * kFinalizeThreads) {
        const float2 v0 = p[i], v1 = p[i + kFinalizeThreads], v2 = p[i + 2 * kFinalizeThreads], v3 = p[i + 3 * kFinalizeThreads];
        s += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        ss += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; i < splits; i += kFinalizeThreads) {
        const float2 v = p[i];
        s += (double)v.x;
        ss += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    if (lane == 0) { red[wid][0] = s; red[wid][1] = ss; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = 0.0;
    ss = 0.0;
    for (int w = 0; w < NW; ++w) { s += red[w][0]; ss += red[w][1]; }
    const float inv = 1.0f / (float)M;
    const float mean = __fmul_rn((float)s, inv);
    const float var = __fsub_rn(__fmul_rn((float)ss, inv), __fmul_rn(mean, mean));
    saved_mean[c] = mean;
    saved_var[c] = var;
    const float vis = mean_shift ? __fadd_rn(mean, mean_shift[c]) : mean;  // see bn_stats_finalize_kernel
    run_mean[c] = __fadd_rn(__fmul_rn(vis, 0.1f), __fmul_rn(run_mean[c], 0.9f));
    run_var[c] = __fadd_rn(__fmul_rn(var, 0.1f), __fmul_rn(run_var[c], 0.9f));
    bn_fwd_consts_store(consts, c, mean, var, scale, bias);
}

// ---- forward apply ---------------------------------------------------------------------------------
// mode TRAIN/VALID: y = act(scale*((x - mean)/sqrtf(var + 1e-6)) + bias);  PREDICT: y = act(x*scale + bias)
// Optional side outputs: ws (copy of x, when ws != x) and xn (normalised values).
struct BnApplyArgs {
    const float* x;
    float* y;
    float* ws;
    float* xn;
    const float* mean;
    const float* var;
    const float* scale;
    const float* bias;
    int C, HW, predict, act;
    long long total;
    // a following eltwise node folded into this pass (bcnn_eltwise_layer.c:82-113): y = act2(act(bn(x)) + res[off]) for
    // off < res_count (the reference adds its second operand to the first res_count elements only), act2(act(bn(x))) behind
    const float* res;
    unsigned res_count;
    int act2;
    const float4* consts;  // per-channel constants left by the statistics finalize kernel (TRAIN), or NULL
};

struct BnApplyBody {
    BnApplyArgs a;
    bool al;
    __device__ void operator()(unsigned off, int c, int cnt) const {
        float mean, sc, b;
        BnDiv rs;
        if (a.consts) {
            const float4 k0 = a.consts[2 * c], k1 = a.consts[2 * c + 1];
            mean = k0.x; rs.d = k0.y; rs.r = k0.z; sc = k0.w; b = k1.x;
        } else {
            mean = a.predict ? 0.f : a.mean[c];
            rs = bn_divisor(a.predict ? 1.f : sqrtf(a.var[c] + 0.000001f));
            sc = a.scale[c]; b = a.bias[c];
        }
        const bool side_ws = a.ws && a.ws != a.x, side_xn = a.xn && !a.predict;
        if (cnt == 4 && al && (off & 3u) == 0) {
            const float4 xv = *reinterpret_cast<const float4*>(a.x + off);
            float4 yv, nv;
            yv.x = bn_one(xv.x, mean, rs, sc, b, a.predict, a.act, &nv.x);
            yv.y = bn_one(xv.y, mean, rs, sc, b, a.predict, a.act, &nv.y);
            yv.z = bn_one(xv.z, mean, rs, sc, b, a.predict, a.act, &nv.z);
            yv.w = bn_one(xv.w, mean, rs, sc, b, a.predict, a.act, &nv.w);
            if (side_ws) *reinterpret_cast<float4*>(a.ws + off) = xv;
            if (side_xn) *reinterpret_cast<float4*>(a.xn + off) = nv;
            if (a.res) {
                if (off + 4 <= a.res_count) {
                    const float4 r = *reinterpret_cast<const float4*>(a.res + off);
                    yv.x += r.x; yv.y += r.y; yv.z += r.z; yv.w += r.w;
                } else if (off < a.res_count) {
                    yv.x += a.res[off];
                    if (off + 1 < a.res_count) yv.y += a.res[off + 1];
                    if (off + 2 < a.res_count) yv.z += a.res[off + 2];
                }
                yv.x = act_fwd_cheap(yv.x, a.act2, 0.f); yv.y = act_fwd_cheap(yv.y, a.act2, 0.f);
                yv.z = act_fwd_cheap(yv.z, a.act2, 0.f); yv.w = act_fwd_cheap(yv.w, a.act2, 0.f);
            }
            BN_ST4(a.y + off, yv);
        } else {
            for (int k = 0; k < cnt; ++k) {
                const float xv = a.x[off + k];
                float nv = 0.f;
                float yv = bn_one(xv, mean, rs, sc, b, a.predict, a.act, &nv);
                if (a.res) yv = act_fwd_cheap(yv + (off + k < a.res_count ? a.res[off + k] : 0.f), a.act2, 0.f);
                if (side_ws) a.ws[off + k] = xv;
                if (side_xn) a.xn[off + k] = nv;
                a.y[off + k] = yv;
            }
        }
    }
};

// ---- backward sums ---------------------------------------------------------------------------------
// g' = dy * act'(y) (optional fused activation backward);  S1 = sum g',  S2 = sum g' * (x - mean)
// The value the forward pass stored for pre-normalisation input x at flat offset `off`, recomputed with the forward's
// operations (bit-identical): act(bn(x)), or with a folded eltwise node (res != NULL) act(bn(x) + res[off]) for
// off < res_count.
__device__ __forceinline__ float bn_recompute_y(float x, float m, const BnDiv& rs, float sc, float b, int act, const float* res,
                                                unsigned res_count, unsigned long long off) {
    float dummy;
    if (!res) return bn_one(x, m, rs, sc, b, 0, act, &dummy);
    float v = bn_one(x, m, rs, sc, b, 0, BCNN_HIP_ACT_NONE, &dummy);
    if (off < res_count) v += res[off];
    return act_fwd_cheap(v, act, 0.f);
}

struct BwdSumsF {
    static constexpr int kInFlight = 2;  // chan_reduce_partial's unroll: <= 48 registers, see there
    const float* dy;
    const float* y;   // post-activation output, used only when act != NONE and fwd_bias == NULL
    const float* x;   // pre-normalisation input (workspace)
    const float* mean;
    // when fwd_bias != NULL the forward output is RECOMPUTED from x (same operations, same rounding as the
    // forward apply => bit-identical y) instead of being read: one full-tensor read less per pass
    const float* fwd_bias;
    const float* var;
    const float* scale;
    int act, C, HW;
    const float* res;     // folded eltwise node (with fwd_bias): its second operand and how much of it is added
    unsigned res_count;
    __device__ float fwd_y(float xv, int c, long long off) const {
        return bn_recompute_y(xv, mean[c], bn_divisor(sqrtf(var[c] + 0.000001f)), scale[c], fwd_bias[c], act, res, res_count,
                              (unsigned long long)off);
    }
    __device__ void operator()(long long off, int c, float (&acc)[2]) const {
        float g = dy[off];
        const float xv = x[off];
        if (act != BCNN_HIP_ACT_NONE) g *= act_bwd_cheap(fwd_bias ? fwd_y(xv, c, off) : y[off], act, 0.f);
        acc[0] += g;
        acc[1] += g * (xv - mean[c]);
    }
    __device__ void vec4(long long off, int c, float (&acc)[2]) const {
        const float m = mean[c];
        float4 g = *reinterpret_cast<const float4*>(dy + off);
        const float4 xv = *reinterpret_cast<const float4*>(x + off);
        if (act != BCNN_HIP_ACT_NONE) {
            float4 yv;
            if (fwd_bias) {
                const BnDiv rs = bn_divisor(sqrtf(var[c] + 0.000001f));
                const float sc = scale[c], b = fwd_bias[c];
                const unsigned long long o = (unsigned long long)off;
                yv.x = bn_recompute_y(xv.x, m, rs, sc, b, act, res, res_count, o);
                yv.y = bn_recompute_y(xv.y, m, rs, sc, b, act, res, res_count, o + 1);
                yv.z = bn_recompute_y(xv.z, m, rs, sc, b, act, res, res_count, o + 2);
                yv.w = bn_recompute_y(xv.w, m, rs, sc, b, act, res, res_count, o + 3);
            } else {
                yv = *reinterpret_cast<const float4*>(y + off);
            }
            g.x *= act_bwd_cheap(yv.x, act, 0.f); g.y *= act_bwd_cheap(yv.y, act, 0.f);
            g.z *= act_bwd_cheap(yv.z, act, 0.f); g.w *= act_bwd_cheap(yv.w, act, 0.f);
        }
        acc[0] += (g.x + g.y) + (g.z + g.w);
        acc[1] += (g.x * (xv.x - m) + g.y * (xv.y - m)) + (g.z * (xv.z - m) + g.w * (xv.w - m));
    }
};

// dbias += S1 ; dscales += S2/sqrt(var+1e-6) ; dmean = scale*S1 * (-1/sqrt(var+1e-5)) ;
// dvar = scale*S2 * (-0.5/(var*sqrt(var)+1e-5))          (bcnn_batchnorm_layer.c:263-281, bcnn_mat.c:692-727)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partials, int C, int splits,
                                       const float* __restrict__ scale, const float* __restrict__ var,
                                       float* __restrict__ dbias, float* __restrict__ dscales,
                                       float* __restrict__ dmean, float* __restrict__ dvar,
                                       float4* __restrict__ consts, const float* __restrict__ mean,
                                       const float* __restrict__ fwd_bias, float fM) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < splits; ++i) {
        s1 += (double)partials[((long long)c * splits + i) * 2 + 0];
        s2 += (double)partials[((long long)c * splits + i) * 2 + 1];
    }
    const float v = var[c], sc = scale[c];
    dbias[c] += (float)s1;
    dscales[c] += (float)(s2 / (double)sqrtf(v + 0.000001f));
    float md = (float)(s1 * (double)sc);
    float vd = (float)(s2 * (double)sc);
    md *= (-1.0f / sqrtf(v + 0.00001f));
    vd *= -0.5f / (v * sqrtf(v) + 0.00001f);
    dmean[c] = md;
    dvar[c] = vd;
    bn_bwd_consts_store(consts, c, mean, v, sc, fwd_bias, md, vd, fM);
}

// The same for MANY partials per channel (a 1x1 convolution's data-gradient epilogue emits one per 64 pixels): one workgroup
// per channel, four loads in flight per thread, fixed-shape reduction in double.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_wide_kernel(const float* __restrict__ partials, int C, int splits,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ var, float* __restrict__ dbias,
                                                                   float* __restrict__ dscales, float* __restrict__ dmean,
                                                                   float* __restrict__ dvar, float4* __restrict__ consts,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ fwd_bias, float fM) {
    __shared__ double red[16][2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = blockIdx.x;
    const float2* p = reinterpret_cast<const float2*>(partials) + (long long)c * splits;
    double s1 = 0.0, s2 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < splits; i += 4 * 1024) {
        const float2 v0 = p[i], v1 = p[i + 1024], v2 = p[i + 2048], v3 = p[i + 3072];
        s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; i < splits; i += 1024) {
        const float2 v = p[i];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) { red[wid][0] = s1; red[wid][1] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s1 = 0.0;
    s2 = 0.0;
    for (int w = 0; w < 16; ++w) { s1 += red[w][0]; s2 += red[w][1]; }
    const float v = var[c], sc = scale[c];
    dbias[c] += (float)s1;
    dscales[c] += (float)(s2 / (double)sqrtf(v + 0.000001f));
    float md = (float)(s1 * (double)sc);
    float vd = (float)(s2 * (double)sc);
    md *= (-1.0f / sqrtf(v + 0.00001f));
    vd *= -0.5f / (v * sqrtf(v) + 0.00001f);
    dmean[c] = md;
    dvar[c] = vd;
    bn_bwd_consts_store(consts, c, mean, v, sc, fwd_bias, md, vd, fM);
}

struct BnBwdApplyArgs {
    float* dy;        // in/out
    float* dx;        // optional copy
    const float* y;   // post-activation (act != NONE)
    const float* x;   // workspace
    const float* mean;
    const float* var;
    const float* scale;
    const float* dmean;
    const float* dvar;
    const float* fwd_bias;  // != NULL: recompute the forward output from x instead of reading y
    int C, HW, act, M;
    long long total;
    int keep_dy;            // dy is only read; the result goes to dx alone
    const float* res;       // folded eltwise node (with fwd_bias), see bn_recompute_y
    unsigned res_count;
    const float4* consts;   // per-channel constants left by the backward finalize kernel, or NULL
    float rM;               // 1 / M, correctly rounded (host division)
};

struct BnBwdApplyBody {
    BnBwdApplyArgs a;
    bool al;
    __device__ void operator()(unsigned off, int c, int cnt) const {
        const BnDiv fM{(float)a.M, a.rM};
        const bool use_act = a.act != BCNN_HIP_ACT_NONE, use_y = use_act && a.fwd_bias == nullptr;
        float mean, sc, fb, dmm, dv;
        BnDiv rs, rs_fwd;
        if (a.consts) {
            const float4 k0 = a.consts[3 * c], k1 = a.consts[3 * c + 1], k2 = a.consts[3 * c + 2];
            mean = k0.x; sc = k0.y; rs.d = k0.z; rs.r = k0.w;
            rs_fwd.d = k1.x; rs_fwd.r = k1.y; fb = k1.z; dmm = k1.w; dv = k2.x;
        } else {
            mean = a.mean[c]; sc = a.scale[c];
            rs = bn_divisor(sqrtf(a.var[c] + 0.00001f));
            rs_fwd = bn_divisor(use_act && !use_y ? sqrtf(a.var[c] + 0.000001f) : 1.0f);
            fb = a.fwd_bias ? a.fwd_bias[c] : 0.f;
            dmm = __fdiv_rn(a.dmean[c], fM.d); dv = a.dvar[c];
        }
        if (cnt == 4 && al && (off & 3u) == 0) {
            const float4 g = *reinterpret_cast<const float4*>(a.dy + off);
            const float4 xv = *reinterpret_cast<const float4*>(a.x + off);
            float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (use_y) yv = *reinterpret_cast<const float4*>(a.y + off);
            else if (use_act) {
                yv.x = bn_recompute_y(xv.x, mean, rs_fwd, sc, fb, a.act, a.res, a.res_count, off);
                yv.y = bn_recompute_y(xv.y, mean, rs_fwd, sc, fb, a.act, a.res, a.res_count, off + 1ull);
                yv.z = bn_recompute_y(xv.z, mean, rs_fwd, sc, fb, a.act, a.res, a.res_count, off + 2ull);
                yv.w = bn_recompute_y(xv.w, mean, rs_fwd, sc, fb, a.act, a.res, a.res_count, off + 3ull);
            }
            float4 o;
            o.x = bn_bwd_one(g.x, yv.x, xv.x, mean, rs, sc, dmm, dv, fM, a.act);
            o.y = bn_bwd_one(g.y, yv.y, xv.y, mean, rs, sc, dmm, dv, fM, a.act);
            o.z = bn_bwd_one(g.z, yv.z, xv.z, mean, rs, sc, dmm, dv, fM, a.act);
            o.w = bn_bwd_one(g.w, yv.w, xv.w, mean, rs, sc, dmm, dv, fM, a.act);
            if (!a.keep_dy) BN_ST4(a.dy + off, o);
            if (a.dx) BN_ST4(a.dx + off, o);
        } else {
            for (int k = 0; k < cnt; ++k) {
                const float xk = a.x[off + k];
                const float yk = use_y ? a.y[off + k]
                                       : (use_act ? bn_recompute_y(xk, mean, rs_fwd, sc, fb, a.act, a.res, a.res_count, off + k) : 0.f);
                const float o = bn_bwd_one(a.dy[off + k], yk, xk, mean, rs, sc, dmm, dv, fM, a.act);
                if (!a.keep_dy) a.dy[off + k] = o;
                if (a.dx) a.dx[off + k] = o;
            }
        }
    }
};


// pre: statistics partials already produced by the convolution epilogue (pre->splits > 0), else NULL
void batchnorm_forward_impl(const float* x, float* y, float* run_mean, float* run_var, const float* scales,
                            const float* bias, float* saved_mean, float* saved_var, float* x_norm, float* workspace,
                            int n, int c, int hw, int mode, int act, const ConvStats* pre, const BnResidual* res,
                            bool stats_only, const float* mean_shift) {
    const long long M = (long long)n * hw, total = M * c;
    if (!total) return;
    const bool have_pre = pre && pre->splits > 0 && mode == BCNN_HIP_MODE_TRAIN;
    // read x twice (statistics, apply) + write y; one read less with fused statistics
    // (stats_only: no apply sweep -- with fused statistics only the partials are read)
    KTimer kt(K_BN_FWD, 0.0, 4.0 * ((have_pre ? 0.0 : 1.0) + (stats_only ? 0.0 : 2.0)) * (double)total);
    const int want_act = act;
    if (!act_is_cheap(act)) act = BCNN_HIP_ACT_NONE;  // tanh/softplus/logistic: separate pass below
    BnApplyArgs a;
    a.x = x; a.y = y; a.ws = workspace; a.xn = x_norm; a.scale = scales; a.bias = bias;
    a.C = c; a.HW = hw; a.act = act; a.total = total;
    a.res = nullptr; a.res_count = 0; a.act2 = BCNN_HIP_ACT_NONE;
    if (res) {  // the caller checked: both activations cheap, 16-byte aligned operand
        a.res = res->res;
        a.res_count = (unsigned)(res->count < (size_t)total ? res->count : (size_t)total);
        a.act2 = res->act;
    }
    a.predict = (mode == BCNN_HIP_MODE_PREDICT);
    a.mean = run_mean; a.var = run_var;
    a.consts = nullptr;
    float4* consts = (mode == BCNN_HIP_MODE_TRAIN && !stats_only) ? bn_consts_scratch(c, false) : nullptr;
    if (mode == BCNN_HIP_MODE_PREDICT) a.ws = nullptr;  // the reference keeps no copy in PREDICT mode
    if (mode == BCNN_HIP_MODE_VALID) a.xn = nullptr;    // x_norm is only written in TRAIN mode (:230)
    if (have_pre) {
        bn_stats_finalize_wide_kernel<<<c, kFinalizeThreads, 0, current_stream()>>>(
            pre->partials, c, pre->splits, (int)M, saved_mean, saved_var, run_mean, run_var, consts, scales, bias, mean_shift);
        KERNEL_CHECK();
        a.mean = saved_mean; a.var = saved_var; a.consts = consts;
    } else if (mode == BCNN_HIP_MODE_TRAIN) {
        const int splits = chan_splits(c, M);
        float* part = reduce_scratch((size_t)c * splits * 2);
        launch_chan_reduce<2>(StatsF{x}, c, hw, M, splits, part);
        bn_stats_finalize_kernel<<<ceil_div(c, 256), 256, 0, current_stream()>>>(
            part, c, splits, (int)M, saved_mean, saved_var, run_mean, run_var, consts, scales, bias, mean_shift);
        KERNEL_CHECK();
        a.mean = saved_mean; a.var = saved_var; a.consts = consts;
    }
    if (stats_only) return;  // the consumer normalises on the fly (bcnn_hip_maxpool_forward_bn)
    auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
#ifdef BCNN_HIP_EXPERIMENT
    static const int skip_fw = getenv("BCNN_HIP_SKIP_SWEEPS") ? atoi(getenv("BCNN_HIP_SKIP_SWEEPS")) : 0;
    if (!(skip_fw & 4))
#endif
    launch_chan_map(BnApplyBody{a, al16(x) && al16(y) && al16(a.ws) && al16(a.xn) && al16(a.res)}, n, c, hw);
    if (want_act != act) bcnn_hip_activation_forward(y, (size_t)total, want_act, nullptr, hw, c);
}

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_batchnorm_forward(const float* x, float* y, float* run_mean, float* run_var,
                                const float* scales, const float* bias, float* saved_mean,
                                float* saved_var, float* x_norm, float* workspace, int n, int c, int hw,
                                int mode, int act) {
    batchnorm_forward_impl(x, y, run_mean, run_var, scales, bias, saved_mean, saved_var, x_norm, workspace, n, c, hw,
                           mode, act, nullptr, nullptr, false, nullptr);
}

void bcnn_hip_batchnorm_forward_stats(const float* x, float* y, float* run_mean, float* run_var, const float* scales,
                                      const float* bias, float* saved_mean, float* saved_var, float* x_norm,
                                      float* workspace, int n, int c, int hw, int mode, int act, const float* stats,
                                      int splits) {
    ConvStats st;
    st.partials = const_cast<float*>(stats); st.splits = stats ? splits : 0; st.capacity = 0;
    batchnorm_forward_impl(x, y, run_mean, run_var, scales, bias, saved_mean, saved_var, x_norm, workspace, n, c, hw,
                           mode, act, &st, nullptr, false, nullptr);
}

// TRAIN-mode batch statistics (saved and running) WITHOUT the apply sweep: the 1x1 convolution behind the node takes the
// node's input and folds the affine map into its weights (bcnn_hip_conv_set_input_bnfold); `stats` as above
void bcnn_hip_batchnorm_forward_stats_only(const float* x, float* run_mean, float* run_var, const float* scales,
                                           const float* bias, float* saved_mean, float* saved_var, int n, int c, int hw,
                                           const float* stats, int splits) {
    ConvStats st;
    st.partials = const_cast<float*>(stats); st.splits = stats ? splits : 0; st.capacity = 0;
    batchnorm_forward_impl(x, nullptr, run_mean, run_var, scales, bias, saved_mean, saved_var, nullptr, nullptr, n, c, hw,
                           BCNN_HIP_MODE_TRAIN, BCNN_HIP_ACT_NONE, &st, nullptr, true, nullptr);
}

}  // extern "C"

namespace bcnn_hip {
// S1 = sum g', S2 = sum g' (x - mean) per channel -> dbias, dscales, dmean, dvar (the first sweep of the backward pass)
// (also called by pool.hip: the stem's sums over the pooled tensors)
void batchnorm_backward_sums(const float* dy, const float* y, int act, const float* scales, float* dscales,
                             float* dbias, const float* saved_mean, const float* saved_var, float* dmean,
                             float* dvar, const float* workspace, int n, int c, int hw, const float* fwd_bias,
                             const float* res = nullptr, unsigned res_count = 0, float4* consts = nullptr,
                             float consts_fM = 0.f /* divisor of dmean in the table; 0: N * hw */) {
    const long long M = (long long)n * hw;
    const int splits = chan_splits(c, M);
    float* part = reduce_scratch((size_t)c * splits * 2);
    BwdSumsF f;
    f.dy = dy; f.y = y; f.x = workspace; f.mean = saved_mean; f.act = act; f.C = c; f.HW = hw;
    f.fwd_bias = fwd_bias; f.var = saved_var; f.scale = scales; f.res = res; f.res_count = res_count;
#ifdef BCNN_HIP_EXPERIMENT
    // timing experiment (wrong results): what the step gains if the sums of the conv1 batch-norms of the 56 x 56 / 28 x 28 blocks
    // came from the data-gradient kernel behind them instead of this sweep
    static const int skip_exp = getenv("BCNN_HIP_SKIP_C1_SUMS") ? 1 : 0;
    static const int skip_all = getenv("BCNN_HIP_SKIP_SWEEPS") ? atoi(getenv("BCNN_HIP_SKIP_SWEEPS")) : 0;  // 1 sums, 2 bwd apply, 4 fwd apply
    if (!(skip_all & 1))
    if (!(skip_exp && res == nullptr && act == BCNN_HIP_ACT_RELU && ((hw == 3136 && c == 64) || (hw == 784 && c == 128))))
#endif
    launch_chan_reduce<2>(f, c, hw, M, splits, part);
    bn_bwd_finalize_kernel<<<ceil_div(c, 256), 256, 0, current_stream()>>>(part, c, splits, scales, saved_var,
                                                                          dbias, dscales, dmean, dvar, consts, saved_mean,
                                                                          fwd_bias, consts_fM > 0.f ? consts_fM : (float)M);
    KERNEL_CHECK();
}

// dy <- scale * g' / sqrt(var + 1e-5) + dvar * 2 (x - mean) / M + dmean / M, copied to dx (the second sweep)
static void batchnorm_backward_apply(float* dy, float* dx, const float* y, int act, const float* scales,
                                     const float* saved_mean, const float* saved_var, const float* dmean,
                                     const float* dvar, const float* workspace, int n, int c, int hw,
                                     const float* fwd_bias, int keep_dy = 0, const float* res = nullptr,
                                     unsigned res_count = 0, const float4* consts = nullptr) {
    const long long M = (long long)n * hw, total = M * c;
    BnBwdApplyArgs a;
    a.consts = consts;
    a.rM = 1.0f / (float)M;  // IEEE, round to nearest: what __fdiv_rn(1.0f, M) gives on the device
    a.dy = dy; a.dx = (dx && dx != dy) ? dx : nullptr; a.y = y; a.x = workspace;
    a.mean = saved_mean; a.var = saved_var; a.scale = scales; a.dmean = dmean; a.dvar = dvar;
    a.C = c; a.HW = hw; a.act = act; a.M = (int)M; a.total = total; a.fwd_bias = fwd_bias;
    a.keep_dy = keep_dy; a.res = res; a.res_count = res_count;
    auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
#ifdef BCNN_HIP_EXPERIMENT
    static const int skip_bw = getenv("BCNN_HIP_SKIP_SWEEPS") ? atoi(getenv("BCNN_HIP_SKIP_SWEEPS")) : 0;
    if (skip_bw & 2) return;
#endif
    launch_chan_map(BnBwdApplyBody{a, al16(dy) && al16(a.dx) && al16(workspace) && al16(y)}, n, c, hw);
}

// fwd_bias (optional): the bias the forward pass added; with it the forward output is recomputed from the
// workspace copy of the input instead of read from y (fused activation backward only).
void batchnorm_backward_impl(float* dy, float* dx, const float* y, int act, const float* scales, float* dscales,
                             float* dbias, const float* saved_mean, const float* saved_var, float* dmean,
                             float* dvar, const float* workspace, int n, int c, int hw, const float* fwd_bias) {
    const long long M = (long long)n * hw, total = M * c;
    if (!total) return;
    if (act == BCNN_HIP_ACT_NONE || !act_bwd_is_cheap(act)) fwd_bias = nullptr;
    // 2 passes over (dy, x[, y]) + write dy
    KTimer kt(K_BN_BWD, 0.0, 4.0 * ((act != BCNN_HIP_ACT_NONE && !fwd_bias) ? 7.0 : 5.0) * (double)total);
    if (!act_bwd_is_cheap(act)) {  // softplus: its derivative needs exp() -> separate in-place pass first
        bcnn_hip_activation_backward(y, dy, (size_t)total, act, nullptr, nullptr, hw, c);
        act = BCNN_HIP_ACT_NONE;
    }
    float4* consts = bn_consts_scratch(c, false);
    batchnorm_backward_sums(dy, y, act, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar, workspace, n, c, hw,
                            fwd_bias, nullptr, 0, consts);
    batchnorm_backward_apply(dy, dx, y, act, scales, saved_mean, saved_var, dmean, dvar, workspace, n, c, hw, fwd_bias, 0,
                             nullptr, 0, consts);
}
// batchnorm_backward_impl whose sums a producer of dy already left as partials[(channel * splits + i) * 2 + {S1, S2}]
// (the depthwise kernel that wrote dy: depthwise_lds.hip): finalize + the apply sweep, no read-only sweep
void batchnorm_backward_presummed(float* dy, const float* y, int act, const float* scales, float* dscales, float* dbias,
                                  const float* saved_mean, const float* saved_var, float* dmean, float* dvar,
                                  const float* workspace, int n, int c, int hw, const float* fwd_bias, const float* sums,
                                  int splits) {
    const long long M = (long long)n * hw, total = M * c;
    if (!total) return;
    if (act == BCNN_HIP_ACT_NONE) fwd_bias = nullptr;
    KTimer kt(K_BN_BWD, 0.0, 4.0 * ((act != BCNN_HIP_ACT_NONE && !fwd_bias) ? 4.0 : 3.0) * (double)total);
    float4* consts = bn_consts_scratch(c, false);
    bn_bwd_finalize_wide_kernel<<<c, 1024, 0, current_stream()>>>(sums, c, splits, scales, saved_var, dbias, dscales, dmean,
                                                                  dvar, consts, saved_mean, fwd_bias, (float)M);
    KERNEL_CHECK();
    batchnorm_backward_apply(dy, nullptr, y, act, scales, saved_mean, saved_var, dmean, dvar, workspace, n, c, hw, fwd_bias, 0,
                             nullptr, 0, consts);
}
// d(res)[i] += dout[i] * act'(out[i]) for the first `count` elements (the partial operand of the folded eltwise node)
__global__ __launch_bounds__(256) void bn_residual_grad_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                               float* __restrict__ dres, unsigned count, int act) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u)
        dres[i] += dout[i] * act_bwd_cheap(out[i], act, 0.f);  // `out` is read here only: one image's worth
}

// Backward of a batch-norm (no activation of its own) whose output went through a folded eltwise node: the incoming
// gradient is dout * act'(out) (bcnn_eltwise_layer.c:124-127), read from the eltwise node's tensors and NOT rewritten; the
// gradient w.r.t. the batch-norm input goes to dx. dres (optional) accumulates the partial operand's gradient.
// The eltwise output needed by act' is RECOMPUTED from x, the forward bias and res (bn_recompute_y: the forward's own
// operations, bit-identical) instead of read: the two sweeps read (dout, x) like a plain batch-norm backward.
void batchnorm_backward_residual(const float* dout, const float* out, int act_res, const float* res, float* dres,
                                 size_t res_count, float* dx, const float* scales, float* dscales, float* dbias,
                                 const float* fwd_bias, const float* saved_mean, const float* saved_var, float* dmean,
                                 float* dvar, const float* workspace, int n, int c, int hw) {
    const long long total = (long long)n * hw * c;
    if (!total) return;
    KTimer kt(K_BN_BWD, 0.0, 4.0 * 5.0 * (double)total);  // (dout, x) twice + dx
    const unsigned cnt = (unsigned)(res_count < (size_t)total ? res_count : (size_t)total);
    if (dres && cnt) {
        bn_residual_grad_kernel<<<stream_grid(cnt, 256), 256, 0, current_stream()>>>(out, dout, dres, cnt, act_res);
        KERNEL_CHECK();
    }
    float4* consts = bn_consts_scratch(c, false);
    batchnorm_backward_sums(dout, out, act_res, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar, workspace, n, c,
                            hw, fwd_bias, res, cnt, consts);
    batchnorm_backward_apply(const_cast<float*>(dout), dx, out, act_res, scales, saved_mean, saved_var, dmean, dvar,
                             workspace, n, c, hw, fwd_bias, /*keep_dy=*/1, res, cnt, consts);
}

}  // namespace bcnn_hip

extern "C" {

void bcnn_hip_batchnorm_apply(const float* x, float* y, const float* scales, const float* bias, const float* saved_mean,
                              const float* saved_var, int n, int c, int hw, int act) {
    const long long total = (long long)n * hw * c;
    if (!total) return;
    BnApplyArgs a;
    a.x = x; a.y = y; a.ws = nullptr; a.xn = nullptr; a.scale = scales; a.bias = bias; a.mean = saved_mean; a.var = saved_var;
    a.C = c; a.HW = hw; a.act = act_is_cheap(act) ? act : BCNN_HIP_ACT_NONE; a.total = total; a.predict = 0;
    a.res = nullptr; a.res_count = 0; a.act2 = BCNN_HIP_ACT_NONE; a.consts = nullptr;
    auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    launch_chan_map(BnApplyBody{a, al16(x) && al16(y)}, n, c, hw);
    if (a.act != act) bcnn_hip_activation_forward(y, (size_t)total, act, nullptr, hw, c);
}

void bcnn_hip_batchnorm_backward_finalize(const float* sums, int splits, const float* scales, float* dscales, float* dbias,
                                          const float* saved_var, float* dmean, float* dvar, int c) {
    if (c <= 0 || splits <= 0) return;
    bn_bwd_finalize_wide_kernel<<<c, 1024, 0, current_stream()>>>(sums, c, splits, scales, saved_var, dbias, dscales, dmean,
                                                                  dvar, nullptr, nullptr, nullptr, 1.0f);
    KERNEL_CHECK();
}

void bcnn_hip_batchnorm_backward_sums(const float* dy, const float* scales, float* dscales, float* dbias,
                                      const float* saved_mean, const float* saved_var, float* dmean, float* dvar,
                                      const float* x, int n, int c, int hw) {
    const long long total = (long long)n * hw * c;
    if (!total) return;
    KTimer kt(K_BN_BWD, 0.0, 4.0 * 2.0 * (double)total);  // one pass over (dy, x)
    batchnorm_backward_sums(dy, nullptr, BCNN_HIP_ACT_NONE, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar, x,
                            n, c, hw, nullptr);
}

void bcnn_hip_batchnorm_backward_apply(float* dy, float* dx, const float* scales, const float* saved_mean,
                                       const float* saved_var, const float* dmean, const float* dvar, const float* x,
                                       int n, int c, int hw) {
    if (!(long long)n * hw * c) return;
    batchnorm_backward_apply(dy, dx, nullptr, BCNN_HIP_ACT_NONE, scales, saved_mean, saved_var, dmean, dvar, x, n, c, hw,
                             nullptr);
}

void bcnn_hip_batchnorm_backward(float* dy, float* dx, const float* y, int act, const float* scales,
                                 float* dscales, float* dbias, const float* saved_mean,
                                 const float* saved_var, float* dmean, float* dvar, const float* x_norm,
                                 const float* workspace, int n, int c, int hw) {
    (void)x_norm;  // recomputed from workspace/mean/var: saves a full-tensor read (and its write in forward)
    batchnorm_backward_impl(dy, dx, y, act, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar, workspace, n,
                            c, hw, nullptr);
}

}  // extern "C"
