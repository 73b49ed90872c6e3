// wino43b_mma.h -- what the F(4x4, 3x3) kernels on v_mfma_f32_16x16x4_f32 share (conv_winograd43b.hip: forward / dX;
// conv_winograd43_dw.hip: weight gradient): the LDS stage geometry and the 36-MFMA step of one sub-chunk.
//   A stage [xi/4 9][k 4][m 64][xi%4 4]   36,864 B: lane l of a wave reads A[m = 16 cb + l % 16][k = l / 16] of FOUR positions
//   B stage [xi/4 9][k 4][n 32][xi%4 4]   18,432 B: B[k = l / 16][n = 16 tbw + l % 16]                      with one ds_read_b128
#pragma once
#include "common.h"
#include "lds_dma.h"

namespace bcnn_hip {

constexpr int WB_BF = 64;                      // output channels per unit
constexpr int WB_BT = 32;                      // tiles per unit
constexpr int WB_KP = 8;                       // reduction channels per period
constexpr int WB_NW = 8;                       // waves
constexpr int WB_USTAGE = 9 * 4 * 64 * 4;      // floats: one sub-chunk of U
constexpr int WB_VC = 9 * 4 * 32 * 4;          // floats: one sub-chunk of V
constexpr int WB_VSTAGE = 2 * WB_VC;           // floats: one period of V

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// 16-byte store as inline assembly with its own wait states (lds_dma.h explains why there is no intrinsic wrapper); an
// out-of-range voff drops the store, soff is not range-checked
__device__ __forceinline__ void wb_store_x4(f32x4 v, rsrc_i4 rs, unsigned voff, unsigned soff) {
    soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);  // wave-uniform by construction; keeps it in a scalar register
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// 36 MFMAs: one reduction step of 4 channels for every position; FIRST: onto zero (a unit's first sub-chunk). The fragments
// come in three groups of 3 + 3 ds_read_b128 (twelve positions), each requested while the group before is multiplied: 48
// registers. Left to itself hipcc hoists all 18 reads to the top (72 registers), which with the 144 accumulators and the
// 25 patch registers in flight no longer fits.
// `between(g)` runs in front of fragment group g = 1, 2 and behind the last one (g = 3): memory requests placed there stall --
// if the texture path is busy -- while the SIMD's other wave multiplies.
template <bool FIRST, class Between>
__device__ __forceinline__ void wb_mma(f32x4 (&acc)[36], const float* up, const float* vp, Between between) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 a4[2][3], b4[2][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        a4[0][x] = *reinterpret_cast<const f32x4*>(up + x * 1024);
        b4[0][x] = *reinterpret_cast<const f32x4*>(vp + x * 512);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < 3) {
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                a4[(g + 1) & 1][x] = *reinterpret_cast<const f32x4*>(up + (3 * (g + 1) + x) * 1024);
                b4[(g + 1) & 1][x] = *reinterpret_cast<const f32x4*>(vp + (3 * (g + 1) + x) * 512);
            }
        }
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xi = 12 * g + 4 * x + j;
                acc[xi] = mfma16(a4[g & 1][x][j], b4[g & 1][x][j], FIRST ? zero : acc[xi]);
            }
        __builtin_amdgcn_sched_barrier(0);
        between(g + 1);
    }
}
template <bool FIRST>
__device__ __forceinline__ void wb_mma(f32x4 (&acc)[36], const float* up, const float* vp) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 a4[2][3], b4[2][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        a4[0][x] = *reinterpret_cast<const f32x4*>(up + x * 1024);
        b4[0][x] = *reinterpret_cast<const f32x4*>(vp + x * 512);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < 3) {
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                a4[(g + 1) & 1][x] = *reinterpret_cast<const f32x4*>(up + (3 * (g + 1) + x) * 1024);
                b4[(g + 1) & 1][x] = *reinterpret_cast<const f32x4*>(vp + (3 * (g + 1) + x) * 512);
            }
        }
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xi = 12 * g + 4 * x + j;
                acc[xi] = mfma16(a4[g & 1][x][j], b4[g & 1][x][j], FIRST ? zero : acc[xi]);
            }
    }
}

}  // namespace bcnn_hip
