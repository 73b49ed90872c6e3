// wino43_math.h -- the data transforms of Winograd F(4x4, 3x3) (Lavin & Gray's matrices), shared by the two kernel forms
// (conv_winograd43.hip, conv_winograd43b.hip) and the weight-gradient kernel.
#pragma once
#include "common.h"

namespace bcnn_hip {

// one 6-vector through B^T: (4 d0 - 5 d2 + d4, -4 d1 - 4 d2 + d3 + d4, 4 d1 - 4 d2 - d3 + d4, -2 d1 - d2 + 2 d3 + d4,
//                            2 d1 - d2 - 2 d3 + d4, 4 d1 - 5 d3 + d5) -- 13 instructions
__device__ __forceinline__ void w43_bt(const float (&d)[6], float (&t)[6]) {
    const float a = __builtin_fmaf(-4.f, d[2], d[4]), b = __builtin_fmaf(-4.f, d[1], d[3]);
    const float c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
}
// one 6-vector through A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void w43_at(const float (&m)[6], float (&y)[4]) {
    const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    y[0] = m[0] + s1 + s2;
    y[1] = __builtin_fmaf(2.f, d2, d1);
    y[2] = __builtin_fmaf(4.f, s2, s1);
    y[3] = __builtin_fmaf(8.f, d2, d1) + m[5];
}

// weight gradient (the transposed algorithm: dW = G^T [ sum_t (A dy_t A^T) . (B^T d_t B) ] G):
// one 4-vector through A = (A^T)^T: (d0, d0 + d1 + d2 + d3, d0 - d1 + d2 - d3, d0 + 2 d1 + 4 d2 + 8 d3, d0 - 2 d1 + 4 d2 - 8 d3, d3)
__device__ __forceinline__ void w43_a(const float (&d)[4], float (&t)[6]) {
    const float s = d[0] + d[2], u = d[1] + d[3];
    const float v = __builtin_fmaf(4.f, d[2], d[0]), w = 2.f * __builtin_fmaf(4.f, d[3], d[1]);
    t[0] = d[0];
    t[1] = s + u;
    t[2] = s - u;
    t[3] = v + w;
    t[4] = v - w;
    t[5] = d[3];
}
// one 6-vector through G^T = [1/4 -1/6 -1/6 1/24 1/24 0; 0 -1/6 1/6 1/12 -1/12 0; 0 -1/6 -1/6 1/6 1/6 1]
__device__ __forceinline__ void w43_gt(const float (&m)[6], float (&y)[3]) {
    const float s1 = m[1] + m[2], d1 = m[2] - m[1], s2 = m[3] + m[4], d2 = m[3] - m[4];
    y[0] = __builtin_fmaf(0.25f, m[0], __builtin_fmaf(-1.f / 6.f, s1, (1.f / 24.f) * s2));
    y[1] = __builtin_fmaf(1.f / 6.f, d1, (1.f / 12.f) * d2);
    y[2] = __builtin_fmaf(1.f / 6.f, s2 - s1, m[5]);
}

// sum over the 16 lanes of a DPP row, in every lane of the row (the first four steps of wave_sum_dpp)
__device__ __forceinline__ float row16_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

}  // namespace bcnn_hip
