// mfma_valu.hip -- do fp32 MFMAs and ordinary VALU instructions overlap on a SIMD? Three kernels with the same
// launch shape: MFMA only, VALU only (v_fma_f32 on independent registers), and both interleaved 1 MFMA : R VALU.
// If the pipes were independent the mixed kernel would take max(t_mfma, t_valu); if they share issue it takes
// the sum. build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int R>  // MODE 0: MFMA only, 1: VALU only, 2: interleaved
__global__ __launch_bounds__(256) void kern(float* out, int iters, float a, float b) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    const float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE != 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(u * R + k) & 7]) : "v"(av), "v"(bv));
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F>
static float run(F launch, int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(blocks, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(blocks, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int R>
static void row(float* out, int occ) {
    const int blocks = 256 * occ, iters = 4000;
    const float tm = run([&](int b, int it) { kern<0, R><<<b, 256>>>(out, it, 1.f, 2.f); }, blocks, iters);
    const float tv = run([&](int b, int it) { kern<1, R><<<b, 256>>>(out, it, 1.f, 2.f); }, blocks, iters);
    const float tb = run([&](int b, int it) { kern<2, R><<<b, 256>>>(out, it, 1.f, 2.f); }, blocks, iters);
    printf("occ %d  %2d VALU per MFMA:  mfma %.3f ms  valu %.3f ms  both %.3f ms   sum %.3f  max %.3f\n", occ, R, tm, tv, tb,
           tm + tv, tm > tv ? tm : tv);
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 8192 * sizeof(float));
    for (int occ = 1; occ <= 4; occ *= 2) { row<4>(out, occ); row<8>(out, occ); row<16>(out, occ); }
    return 0;
}
