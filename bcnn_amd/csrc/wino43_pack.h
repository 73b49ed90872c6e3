// wino43_pack.h -- the weight transform of the F(4x4, 3x3) kernels (conv_winograd43.hip), shared with the multi-layer pack
// launch of conv_winograd_fused.hip (bcnn_hip_conv_prepack).
#pragma once
#include "conv_common.h"

namespace bcnn_hip {

// U[xi][j][m] = (G g G^T)[xi] with zero padding.
//   forward: m = f, j = c; dX: m = c, j = f and the filter rotated by 180 degrees
//   layout 0: [36][Jpad][Mpad] (conv_winograd43.hip)
//   layout 1: [m / 64][j / 4][xi / 4][j % 4][m % 64][xi % 4] -- the LDS stages of conv_winograd43b.hip, one linear 36 KB run
//             per (channel block, sub-chunk of 4 reduction channels); Mpad a multiple of 64, Jpad of 8
__device__ __forceinline__ void wino43_pack_one(const float* __restrict__ w, float* __restrict__ u, int F, int C, int dx_mode,
                                                int Jpad, int Mpad, int idx, int layout = 0) {
    if (idx >= Jpad * Mpad) return;
    const int j = idx / Mpad, m = idx - j * Mpad;
    const int M = dx_mode ? C : F, J = dx_mode ? F : C;
    float g[3][3];
    const bool live = m < M && j < J;
    {
        const int f = dx_mode ? j : m, c = dx_mode ? m : j;
        const float* p = w + (live ? ((size_t)f * C + c) * 9 : 0);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[r][b] = live ? (dx_mode ? p[(2 - r) * 3 + (2 - b)] : p[r * 3 + b]) : 0.f;
    }
    // G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
    auto gvec = [](float a0, float a1, float a2, float (&o)[6]) {
        o[0] = a0 * (1.f / 4.f);
        o[1] = (a0 + a1 + a2) * (-1.f / 6.f);
        o[2] = (a0 - a1 + a2) * (-1.f / 6.f);
        o[3] = a0 * (1.f / 24.f) + a1 * (1.f / 12.f) + a2 * (1.f / 6.f);
        o[4] = a0 * (1.f / 24.f) - a1 * (1.f / 12.f) + a2 * (1.f / 6.f);
        o[5] = a2;
    };
    float t[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float o[6];
        gvec(g[0][b], g[1][b], g[2][b], o);
#pragma unroll
        for (int r = 0; r < 6; ++r) t[r][b] = o[r];
    }
    const size_t plane = (size_t)Jpad * Mpad;
    float* dst = u + idx;
    float* dst1 = u + ((((size_t)(m >> 6) * (Jpad >> 2) + (j >> 2)) * 9 * 4 + (j & 3)) * 64 + (m & 63)) * 4;  // + (xi / 4) * 1024 + xi % 4
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float o[6];
        gvec(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int xi = 6 * r + c;
            if (layout == 0) dst[(size_t)xi * plane] = o[c];
            else dst1[(xi >> 2) * 1024 + (xi & 3)] = o[c];
        }
    }
}


}  // namespace bcnn_hip
