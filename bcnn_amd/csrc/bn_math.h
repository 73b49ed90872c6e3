// bn_math.h -- the per-element arithmetic of batch normalisation, shared by batchnorm.hip and the depthwise backward
// kernel that applies a following batch-norm node's backward on the fly (depthwise_lds.hip). Operation order and rounding
// follow bcnn_batchnorm_layer.c:196-242 (forward) and :292-296 (backward) exactly.
#pragma once
#include "common.h"

namespace bcnn_hip {
#ifdef __HIPCC__
// A divisor that is constant per channel (sqrt(var + eps), the element count M) together with its correctly rounded
// reciprocal. bn_div(a, d) = a / d, correctly rounded like the reference's division, in three instructions instead of the
// ~12 of an IEEE division sequence: q0 = a * r, then ONE correction step with the exact remainder, q1 = q0 + (a - d * q0) * r
// (Markstein: with r = RN(1 / d) and no over / underflow q1 = RN(a / d); tools/micro/div_exact.hip finds no mismatch against
// __fdiv_rn in 2.4e11 operand pairs, divisors with all-ones / all-zero / alternating significands included). A quotient
// that is not finite is returned uncorrected (Inf / d = Inf, NaN stays NaN: the remainder would be NaN). Known difference:
// -0 / d gives +0.
struct BnDiv {
    float d, r;
};
__device__ __forceinline__ BnDiv bn_divisor(float d) {
    BnDiv v;
    v.d = d;
    v.r = __fdiv_rn(1.0f, d);
    return v;
}
__device__ __forceinline__ float bn_div(float a, const BnDiv& dv) {
    const float q0 = __fmul_rn(a, dv.r);
    const float q1 = __fmaf_rn(__fmaf_rn(-dv.d, q0, a), dv.r, q0);
    return fabsf(q0) < __builtin_inff() ? q1 : q0;
}

__device__ __forceinline__ float bn_one(float x, float mean, const BnDiv& rs, float sc, float b, int predict,
                                        int act, float* xn_out) {
    float v;
    if (predict) {
        v = __fadd_rn(__fmul_rn(x, sc), b);  // scale_and_add_bias, bcnn_batchnorm_layer.c:183-194
    } else {
        v = bn_div(__fsub_rn(x, mean), rs);
        *xn_out = v;
        if (sc == 0.0f) v = 0.f;             // bcnn_scal: a == 0 -> memset
        else if (sc != 1.0f) v = __fmul_rn(v, sc);
        if (b != 0.0f && b != 1.0f) v = __fadd_rn(v, b);  // bcnn_add_scalar quirk
    }
    return act_fwd_cheap(v, act, 0.f);
}

__device__ __forceinline__ float bn_bwd_one(float g, float yv, float xv, float mean, const BnDiv& rs, float sc,
                                            float dm_m, float dv, const BnDiv& fM, int act) {
    if (act != BCNN_HIP_ACT_NONE) g *= act_bwd_cheap(yv, act, 0.f);
    if (sc == 0.0f) g = 0.f;
    else if (sc != 1.0f) g = __fmul_rn(g, sc);
    // grad*1.0f/sqrtf(var+1e-5) + dvar*2*(x-mean)/M + dmean/M     (bcnn_batchnorm_layer.c:292-296)
    const float t1 = bn_div(g, rs);
    const float t2 = bn_div(__fmul_rn(__fmul_rn(dv, 2.0f), __fsub_rn(xv, mean)), fM);
    return __fadd_rn(__fadd_rn(t1, t2), dm_m);
}
#endif
}  // namespace bcnn_hip
