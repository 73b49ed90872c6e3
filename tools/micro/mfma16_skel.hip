// mfma16_skel.hip -- what the skeleton of conv_winograd43b.hip can sustain: 8 waves per CU (two per SIMD), each 36 independent
// v_mfma_f32_16x16x4_f32 accumulators, groups of 36 MFMAs with or without a workgroup barrier / LDS fragment reads in between.
// build: hipcc --offload-arch=gfx950 -O3 mfma16_skel.hip -o mfma16_skel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NW, int MODE>   // MODE 0: registers only; 1: + barrier per group; 2: + 18 ds_read_b128 per group; 3: both
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void k16(float* out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(1024))) float lds[36864];   // 147,456 B like the kernel: one workgroup per CU
    f32x4 acc[36];
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    if (MODE >= 2) for (int i = threadIdx.x; i < 36864; i += 64 * NW) lds[i] = a + i;
    __syncthreads();
    f32x4 a4[9], b4[9];
    for (int x = 0; x < 9; ++x) { a4[x] = f32x4{a + lane, a, b, a + x}; b4[x] = f32x4{b, a, b + x, b}; }
    const float* up = lds + lane * 4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE >= 2) {
#pragma unroll
            for (int x = 0; x < 9; ++x) {
                a4[x] = *reinterpret_cast<const f32x4*>(up + x * 1024 + (it & 1) * 9216);
                b4[x] = *reinterpret_cast<const f32x4*>(up + 18432 + x * 512 + (it & 1) * 9216);
            }
        }
#pragma unroll
        for (int x = 0; x < 9; ++x)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * x + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[x][j], b4[x][j], acc[4 * x + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 36; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
}
// the first form's shape: 12 waves, 3 accumulators of 32x32x2, 12 MFMAs per group
template <int MODE>
__global__ __launch_bounds__(768, 3) void k32(float* out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(1024))) float lds[36864];
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63;
    float av[12], bv[12];
    for (int x = 0; x < 12; ++x) { av[x] = a + lane + x; bv[x] = b + x; }
    if (MODE >= 2) for (int i = threadIdx.x; i < 36864; i += 768) lds[i] = a + i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3 * ks + j], bv[3 * ks + j], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 768 + threadIdx.x] = s;
}
template <class F>
static void run(const char* name, F launch, double flop_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, iters);   // warm clocks too
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop_per_block_iter * blocks * iters / ms / 1e9);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 4000;   // x 36 MFMAs: ~ the kernel's 1800 per launch, twice over
    const double f16 = 36.0 * 2048 * 8;   // per block and iteration, 8 waves
    for (int rep = 0; rep < 2; ++rep) {
        run("16x16x4 8 waves, registers", [&](int b, int it) { k16<8, 0><<<b, 512>>>(out, it, 1.f, 2.f); }, f16, 256, iters);
        run("16x16x4 8 waves, + barrier per 36", [&](int b, int it) { k16<8, 1><<<b, 512>>>(out, it, 1.f, 2.f); }, f16, 256, iters);
        run("16x16x4 8 waves, + 18 ds_read_b128 per 36", [&](int b, int it) { k16<8, 2><<<b, 512>>>(out, it, 1.f, 2.f); }, f16, 256, iters);
        run("16x16x4 8 waves, + both", [&](int b, int it) { k16<8, 3><<<b, 512>>>(out, it, 1.f, 2.f); }, f16, 256, iters);
        run("16x16x4 4 waves, registers", [&](int b, int it) { k16<4, 0><<<b, 256>>>(out, it, 1.f, 2.f); }, f16 / 2, 256, iters);
        run("32x32x2 12 waves x 12, registers", [&](int b, int it) { k32<0><<<b, 768>>>(out, it, 1.f, 2.f); }, 12.0 * 4096 * 12, 256, iters);
        run("32x32x2 12 waves x 12, + barrier", [&](int b, int it) { k32<1><<<b, 768>>>(out, it, 1.f, 2.f); }, 12.0 * 4096 * 12, 256, iters);
    }
    return 0;
}
