// v_mfma_f32_32x32x16_bf16 on gfx950: operand layout check against a host product, and issue rate.
// build: hipcc --offload-arch=gfx950 -O3 mfma_bf16.hip -o mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// assumed layout: A[i][k]: lane l holds i = l % 32, k = 8 * (l / 32) + e; B[k][j]: lane l holds j = l % 32, k = 8 * (l / 32) + e;
// D[i][j]: j = l % 32, i = (r & 3) + 8 * (r >> 2) + 4 * (l / 32)
__global__ void layout(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[(l % 32) * 16 + 8 * (l / 32) + e];
        b[e] = (__bf16)B[(8 * (l / 32) + e) * 32 + (l % 32)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l / 32)) * 32 + (l % 32)] = c[r];
}
template <int NACC>
__global__ void rate(float* out, int iters, long long* cyc) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) c[n][r] = 0.f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[n], 0, 0, 0);
    const long long t1 = clock64();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += c[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], *dA, *dB, *dD;
    for (auto& v : hA) v = (float)((rand() % 17) - 8) * 0.25f;   // exactly representable in bf16
    for (auto& v : hB) v = (float)((rand() % 13) - 6) * 0.5f;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int k = 0; k < 16; ++k) s += (double)hA[i * 16 + k] * hB[k * 32 + j];
        worst = fmax(worst, fabs(s - hD[i * 32 + j]));
    }
    printf("layout check: max |D - A*B| = %g (0 = the assumed operand layout is right)\n", worst);
    float* out; long long *cyc, hc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int waves : {1, 2, 4}) {
        rate<4><<<256, 64 * waves * 4>>>(out, iters, cyc);  // waves per SIMD = waves (4 SIMDs per CU, one block per CU)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); rate<4><<<256, 64 * waves * 4>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        const double fl = 2.0 * 32 * 32 * 16 * 4.0 * iters * 256 * 4 * waves;
        printf("%d wave(s)/SIMD: %.3f ms, %.0f TFLOP/s, %.1f cycles per MFMA per SIMD (s_memtime)\n", waves, ms, fl / ms / 1e9,
               (double)hc / (4.0 * iters * waves));
    }
    return 0;
}
