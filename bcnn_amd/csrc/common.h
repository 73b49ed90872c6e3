// common.h -- shared device/host helpers for the gfx950 kernels (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/bcnn_hip.h"

// Reference convention (src/bcnn_utils.h:174-195): print and exit on any device error.
#define HIP_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            fprintf(stderr, "[bcnn_hip] %s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr,    \
                    hipGetErrorString(_e));                                                    \
            exit((int)_e);                                                                     \
        }                                                                                      \
    } while (0)

#define KERNEL_CHECK() HIP_CHECK(hipGetLastError())

// A/B switches for kernel experiments (alternative tiles, forcing the fallback kernels, ...) exist only in the
// EXPERIMENT build of the library (make exp -> ../lib/libbcnn_hip_exp.so, -DBCNN_HIP_EXPERIMENT; used by tools/exp
// and by tests/test_fallback_paths.py through BCNN_HIP_LIB). The product build reads no environment variable.
#ifdef BCNN_HIP_EXPERIMENT
#define BCNN_EXP_ENV(name) getenv(name)
#else
#define BCNN_EXP_ENV(name) ((const char*)nullptr)
#endif

namespace bcnn_hip {

hipStream_t current_stream();
void set_current_stream(hipStream_t st);  // runtime.hip

// Optional per-kernel-class timing with HIP events on the launch stream (off by default; bench.py turns it
// on to report the roofline of the dominant kernel from inside the timed region). runtime.hip.
enum KClass { K_CONV_FWD = 0, K_CONV_DW, K_CONV_DX, K_BN_FWD, K_BN_BWD, K_POOL, K_ELTWISE_ACT, K_GEMM, K_SGD,
              K_DEPTHWISE_FWD, K_DEPTHWISE_BWD,
              K_CONV_FWD_WINO, K_CONV_DX_WINO, K_CONV_DW_WINO,  // Winograd F(2x2,3x3) layers: FLOPs = what the MFMAs execute
              K_CONV_FWD_WINO43, K_CONV_DX_WINO43,              // Winograd F(4x4,3x3) layers, likewise (36 positions per 4 x 4 outputs)
              K_CONV_DW_WINO43,
              K_NUM };
struct KTimer {
    int idx;
    // useful_flops: the part of `flops` that is not padding (Winograd tiles hanging over an odd-sized plane); < 0: all of it
    KTimer(int cls, double flops, double bytes, double useful_flops = -1.0);
    ~KTimer();
};

// Dispatch trace (bcnn_hip_trace_*, off by default): dispatchers name the kernel family they launch. runtime.hip
extern thread_local bool g_trace_on;
void trace_kernel_slow(const char* name);
inline void trace_kernel(const char* name) { if (g_trace_on) trace_kernel_slow(name); }

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kCUs = 256;      // MI355X
constexpr int kXCDs = 8;

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

#ifndef STREAM_GRID_PER_CU
#define STREAM_GRID_PER_CU 8
#endif
// grid for a grid-stride streaming kernel: enough blocks to fill the chip, capped (guide G11).
inline int stream_grid(size_t work_items, int block) {
    size_t need = (work_items + block - 1) / block;
    size_t cap = (size_t)kCUs * STREAM_GRID_PER_CU;
    if (need < 1) need = 1;
    return (int)(need < cap ? need : cap);
}

#ifdef __HIPCC__
// XCD-aware block remap: the dispatcher places block b on XCD b % 8; give each XCD a contiguous run
// of logical tiles so neighbours (which share operand panels / halos) hit the same L2.
// Bijective for any nblk (guide section 5 "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / kXCDs, r = nblk % kXCDs;
    const int xcd = bid % kXCDs, idx = bid / kXCDs;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Wave-wide sum on the DPP path (no LDS crossbar round trips, ~8 vector-ALU instructions); result valid in LANE 63 only.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror: 16-lane sums
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2, 3
    return v;
}

// block-wide sum of one value; result valid in thread 0. `red` holds >= blockDim/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) t += red[i];
    }
    return t;
}

// The activation map, bcnn_activation_layer.c:90-146. exp/log are evaluated in double exactly
// where the reference does, so results agree to rounding of the final float conversion.
__device__ __forceinline__ float act_fwd(float x, int act, float slope) {
    switch (act) {
        case BCNN_HIP_ACT_TANH: {
            const float t = 2 * x;
            const double e = exp((double)t);
            return (float)(e - 1) / ((float)e + 1);
        }
        case BCNN_HIP_ACT_RELU: return x * (float)(x > 0);  // a multiply: -0.0f for negatives, NaN propagates
        case BCNN_HIP_ACT_LRELU: return x > 0 ? x : 0.1f * x;
        case BCNN_HIP_ACT_RAMP: return x * (float)(x > 0) + 0.1f * x;
        case BCNN_HIP_ACT_SOFTPLUS: return (float)log((double)(1.0f + (float)exp((double)x)));
        case BCNN_HIP_ACT_ABS: return fabsf(x);
        case BCNN_HIP_ACT_CLAMP: return (x < 0) ? 0.f : ((x > 1) ? 1.f : x);
        case BCNN_HIP_ACT_LOGISTIC: return 1.0f / (1.0f + (float)exp((double)(-x)));
        case BCNN_HIP_ACT_PRELU: return x > 0 ? x : slope * x;
        default: return x;
    }
}

// Activations cheap enough to fuse into a producer's epilogue (no double-precision exp/log). The
// other three (tanh, softplus, logistic) run as a separate in-place pass (activation.hip) so that
// the MFMA kernels keep their register budget.
__host__ __device__ __forceinline__ bool act_is_cheap(int act) {
    return act != BCNN_HIP_ACT_TANH && act != BCNN_HIP_ACT_SOFTPLUS && act != BCNN_HIP_ACT_LOGISTIC;
}
__device__ __forceinline__ float act_fwd_cheap(float x, int act, float slope) {
    switch (act) {
        case BCNN_HIP_ACT_RELU: return x * (float)(x > 0);
        case BCNN_HIP_ACT_LRELU: return x > 0 ? x : 0.1f * x;
        case BCNN_HIP_ACT_RAMP: return x * (float)(x > 0) + 0.1f * x;
        case BCNN_HIP_ACT_ABS: return fabsf(x);
        case BCNN_HIP_ACT_CLAMP: return (x < 0) ? 0.f : ((x > 1) ? 1.f : x);
        case BCNN_HIP_ACT_PRELU: return x > 0 ? x : slope * x;
        default: return x;
    }
}
__device__ __forceinline__ float act_bwd_cheap(float y, int act, float slope) {
    switch (act) {
        case BCNN_HIP_ACT_TANH: return 1 - y * y;
        case BCNN_HIP_ACT_RELU: return (float)(y > 0);
        case BCNN_HIP_ACT_LRELU: return y > 0 ? 1.0f : 0.1f;
        case BCNN_HIP_ACT_RAMP: return (float)(y > 0) + 0.1f;
        case BCNN_HIP_ACT_ABS: return y >= 0 ? 1.0f : -1.0f;
        case BCNN_HIP_ACT_CLAMP: return (float)(y > 0.0f && y < 1.0f);
        case BCNN_HIP_ACT_LOGISTIC: return (1 - y) * y;
        case BCNN_HIP_ACT_PRELU: return y > 0 ? 1.0f : slope;
        default: return 1.0f;
    }
}
__host__ __device__ __forceinline__ bool act_bwd_is_cheap(int act) { return act != BCNN_HIP_ACT_SOFTPLUS; }

// derivative factor from the POST-activation value, bcnn_activation_layer.c:165-226
__device__ __forceinline__ float act_bwd_factor(float y, int act, float slope) {
    switch (act) {
        case BCNN_HIP_ACT_TANH: return 1 - y * y;
        case BCNN_HIP_ACT_RELU: return (float)(y > 0);
        case BCNN_HIP_ACT_LRELU: return y > 0 ? 1.0f : 0.1f;
        case BCNN_HIP_ACT_RAMP: return (float)(y > 0) + 0.1f;
        case BCNN_HIP_ACT_SOFTPLUS: return 1.0f / (1.0f + (float)exp((double)(-y)));
        case BCNN_HIP_ACT_ABS: return y >= 0 ? 1.0f : -1.0f;
        case BCNN_HIP_ACT_CLAMP: return (float)(y > 0.0f && y < 1.0f);
        case BCNN_HIP_ACT_LOGISTIC: return (1 - y) * y;
        case BCNN_HIP_ACT_PRELU: return y > 0 ? 1.0f : slope;
        default: return 1.0f;
    }
}
#endif  // __HIPCC__

}  // namespace bcnn_hip
