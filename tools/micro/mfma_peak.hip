// mfma_peak.hip -- what v_mfma_f32_32x32x2_f32 / 16x16x4_f32 actually sustain on this part (no memory traffic).
// build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// plain vector FMA for comparison (packed v_pk_fma_f32 if the compiler finds it)
__global__ __launch_bounds__(256) void kfma(float* out, int iters, float a, float b) {
    float2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = make_float2(0.f, 0.f);
    float2 av = make_float2(a + threadIdx.x, a), bv = make_float2(b, b * 0.5f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[i].x = __builtin_fmaf(av.x, bv.x, acc[i].x); acc[i].y = __builtin_fmaf(av.y, bv.y, acc[i].y); }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F>
static void run(const char* name, F launch, double flop_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flop_per_block_iter * blocks * iters / ms / 1e9);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 8192 * sizeof(float));
    const int iters = 2000;
    for (int occ = 1; occ <= 4; ++occ) {
        const int blocks = 256 * occ;
        char nm[64];
        snprintf(nm, 64, "32x32x2 acc4 occ%d", occ);
        run(nm, [&](int b, int it) { k32<4><<<b, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 4 * 4096, blocks, iters);
        snprintf(nm, 64, "32x32x2 acc1 occ%d", occ);
        run(nm, [&](int b, int it) { k32<1><<<b, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 1 * 4096, blocks, iters);
        snprintf(nm, 64, "16x16x4 acc4 occ%d", occ);
        run(nm, [&](int b, int it) { k16<4><<<b, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 4 * 2048, blocks, iters);
        snprintf(nm, 64, "vector fma occ%d", occ);
        run(nm, [&](int b, int it) { kfma<<<b, 256>>>(out, it, 1.f, 2.f); }, 256.0 * 4 * 32 * 2, blocks, iters);
    }
    return 0;
}
