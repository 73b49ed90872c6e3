// bn_math.h -- the per-element arithmetic of batch normalisation, shared by batchnorm.hip and the depthwise backward
// kernel that applies a following batch-norm node's backward on the fly (depthwise_lds.hip). Operation order and rounding
// follow bcnn_batchnorm_layer.c:196-242 (forward) and :292-296 (backward) exactly.
#pragma once
#include "common.h"

namespace bcnn_hip {
#ifdef __HIPCC__
__device__ __forceinline__ float bn_one(float x, float mean, float rs, float sc, float b, int predict,
                                        int act, float* xn_out) {
    float v;
    if (predict) {
        v = __fadd_rn(__fmul_rn(x, sc), b);  // scale_and_add_bias, bcnn_batchnorm_layer.c:183-194
    } else {
        v = __fdiv_rn(__fsub_rn(x, mean), rs);
        *xn_out = v;
        if (sc == 0.0f) v = 0.f;             // bcnn_scal: a == 0 -> memset
        else if (sc != 1.0f) v = __fmul_rn(v, sc);
        if (b != 0.0f && b != 1.0f) v = __fadd_rn(v, b);  // bcnn_add_scalar quirk
    }
    return act_fwd_cheap(v, act, 0.f);
}

__device__ __forceinline__ float bn_bwd_one(float g, float yv, float xv, float mean, float rs, float sc,
                                            float dm_m, float dv, float fM, int act) {
    if (act != BCNN_HIP_ACT_NONE) g *= act_bwd_cheap(yv, act, 0.f);
    if (sc == 0.0f) g = 0.f;
    else if (sc != 1.0f) g = __fmul_rn(g, sc);
    // grad*1.0f/sqrtf(var+1e-5) + dvar*2*(x-mean)/M + dmean/M     (bcnn_batchnorm_layer.c:292-296)
    const float t1 = __fdiv_rn(g, rs);
    const float t2 = __fdiv_rn(__fmul_rn(__fmul_rn(dv, 2.0f), __fsub_rn(xv, mean)), fM);
    return __fadd_rn(__fadd_rn(t1, t2), dm_m);
}
#endif
}  // namespace bcnn_hip
