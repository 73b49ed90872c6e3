// conv_common.h -- shapes and MFMA helpers shared by the convolution kernels.
#pragma once
#include "common.h"

namespace bcnn_hip {

// One convolution problem, whole batch. Derived fields filled by make_conv_shape().
struct ConvShape {
    int N, C, H, W, F, ksz, stride, pad, groups;
    int OH, OW, OHOW, HW, Cg, Mg, K;  // Cg = C/groups, Mg = F/groups, K = Cg*ksz*ksz
    long long total_q;                // N*OH*OW output pixel columns (batch folded into GEMM-N)
    long long total_p;                // N*H*W input pixel columns (for dX)
    int pointwise;                    // ksz == 1: source read as a raw [Cg][OH*OW] matrix (quirk 1)
};

inline ConvShape make_conv_shape(int n, int c, int h, int w, int f, int k, int stride, int pad,
                                 int groups) {
    ConvShape s;
    s.N = n; s.C = c; s.H = h; s.W = w; s.F = f; s.ksz = k; s.stride = stride; s.pad = pad;
    s.groups = groups;
    s.OH = (h + 2 * pad - k) / stride + 1;  // reference bcnn_conv_layer.c:126-134
    s.OW = (w + 2 * pad - k) / stride + 1;
    s.OHOW = s.OH * s.OW; s.HW = h * w;
    s.Cg = c / groups; s.Mg = f / groups; s.K = s.Cg * k * k;
    s.total_q = (long long)n * s.OHOW;
    s.total_p = (long long)n * s.HW;
    s.pointwise = (k == 1);
    return s;
}

// Batch-norm statistics produced by a convolution kernel's epilogue (conv_igemm_dma.hip):
// partials[(channel * splits + i) * 2 + {sum, sum of squares}], the layout bn_stats_finalize consumes.
struct ConvStats {
    float* partials;
    int splits;
    size_t capacity;  // floats behind `partials`; every forward path checks its own slot count against it
};

// Backward sums of the stand-alone batch-norm node IN FRONT of a 1x1 convolution, emitted by that convolution's data-gradient
// kernel from the tile it is about to store (bcnn_batchnorm_layer.c:263-281 needs S1 = sum dz, S2 = sum dz * (y - mean)
// over (n, h, w) per channel; dz is this kernel's output, y the batch-norm node's input):
// partials[(channel * splits + column tile) * 2 + {S1, S2}], the layout bn_bwd_finalize consumes.
struct DxBnSums {
    const float* y;      // the batch-norm node's input (same shape as the convolution's input)
    const float* mean;   // its saved batch mean [C]
    float* partials;
    size_t capacity;     // floats behind `partials`
    int splits;          // out: partials per channel written, 0 = this path does not emit them
};

// A following eltwise node folded into a convolution node's batch-norm apply pass (bcnn_eltwise_layer.c:82-113)
struct BnResidual {
    const float* res;  // the eltwise node's second operand
    size_t count;      // elements of it that are added (the reference adds min_c * H * W of them: image 0)
    int act;           // the eltwise node's activation
};

// ---- weight packs made ahead of their use, for many layers in one launch (conv.hip: bcnn_hip_conv_prepack) -------
// The convolution kernels read the filter bank re-arranged: G g G^T in [16][Jpad][Mpad] (fused Winograd), A^T in
// [group][tap][Jpad][Mpad] (LDS-DMA GEMM). Stand-alone calls pack right before the kernel (one ~5 us launch per call);
// a caller that knows all layers of a pass hands them over at its start and the packs of a kind become ONE launch.
enum { PREPACK_WINO = 0, PREPACK_IGEMM = 1, PREPACK_KINDS = 2 };
struct WinoPackJob {
    const float* w;
    float* u;
    int F, C, dx_mode, Jpad, Mpad;
    int blocks;  // of 256 threads
    int npos;    // 16: G g G^T of F(2x2,3x3) (conv_winograd_fused.hip); 36: of F(4x4,3x3) (conv_winograd43.hip)
    int layout;  // npos 36 only: wino43_pack_one's layout (1: the stage order of conv_winograd43b.hip)
};
constexpr int kPackMaxTaps = 49;
struct IgemmPackJob {
    const float* w;
    float* at;
    int Mg, Cg, kk2, ksz;
    int M, J, Jpad, Mpad;
    int mode, groups;
    int gx, gy, gz;  // block grid of this job: 64 m x 4 j per block, gz = groups * taps
    unsigned char tapoff[kPackMaxTaps];  // kr*ksz + kc of packed tap index
    // forward form only, optional (BnFold below): row j (input channel c = g * Cg + j) of the pack is multiplied by
    // fold_scales[c] / sqrtf(fold_var[c] + 1e-6f)
    const float* fold_var;
    const float* fold_scales;
};

// The stand-alone batch-norm node IN FRONT of a 1x1 convolution, folded into that convolution (TRAIN mode, no activation in
// between: z = a y + b per channel with a = scale / sqrt(var + 1e-6), b = bias - mean a, bcnn_batchnorm_layer.c:226-241):
// W z = (W diag(a)) y + W b, so the GEMM reads the batch-norm's INPUT y with column-scaled weights and the per-filter
// constant W b only moves the mean of the convolution's own batch-norm behind it (conv.hip: bcnn_hip_conv_set_input_bnfold).
struct BnFold {
    const float* mean;   // saved batch mean [C]; nullptr: no fold
    const float* var;
    const float* scales;
    const float* bias;
};
#ifdef __HIPCC__
__device__ __forceinline__ float bnfold_a(const float* __restrict__ var, const float* __restrict__ scales, int c) {
    return scales[c] / sqrtf(var[c] + 0.000001f);  // one formula for the pack, the constant and the weight gradient
}
#endif
// true while the calling thread's weight gradients go to the side stream without a join per call (conv.hip)
bool conv_side_stream_deferred();
// the pack of (w, kind, mode) made by the current prepack batch and not yet used, or nullptr (conv.hip)
float* prepack_take(const float* w, int kind, int mode, size_t floats);
// what the kernel a layer will run on needs packed; false: nothing (another kernel takes the layer)
bool wino_fused_pack_plan(const ConvShape& s, int dx_mode, WinoPackJob* job, size_t* floats);  // conv_winograd_fused.hip
bool wino43_pack_plan(const ConvShape& s, int dx_mode, WinoPackJob* job, size_t* floats);      // conv_winograd43.hip
bool dma_pack_plan(const ConvShape& s, int dx_mode, IgemmPackJob* job, size_t* floats);        // conv_igemm_dma.hip
bool conv_winograd_unfused_takes(const ConvShape& s);                                          // conv_winograd.hip
void wino_fused_pack_launch(const WinoPackJob* jobs_dev, int n, int max_blocks);
void dma_pack_launch(const IgemmPackJob* jobs_dev, int n, int max_blocks);

#ifdef __HIPCC__
typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact fp32 fma chain (guide section 3).
// lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5) for accumulator register r in [0,16).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
#endif

}  // namespace bcnn_hip
