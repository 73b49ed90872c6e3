// Issue cost of the vector-ALU instructions the Winograd produce step is made of (gfx950): N independent chains per wave,
// s_memtime around a long unrolled loop. build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float* out, int iters, long long* cyc) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.5f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    const float s = out[0];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + s;                                   // v_add_f32
            if (OP == 1) p[i] = p[i] + f2{s, s};                            // v_pk_add_f32
            if (OP == 2) { bf2 h = __builtin_convertvector(p[i], bf2); p[i][0] += __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned, h) << 16); }  // cvt_pk + shift + add
            if (OP == 3) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x138, 0xf, 0xf, false)) ;  // v_mov_dpp wave_shr
            if (OP == 4) a[i] = (threadIdx.x & 1) ? a[i] : s;               // v_cndmask
            if (OP == 5) a[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a[i]) & 0xffff0000u);  // v_and
            if (OP == 6) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x111, 0xf, 0xf, false)) ;  // v_mov_dpp row_shr
        }
    }
    const long long t1 = clock64();
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + p[i][0] + p[i][1];
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, float* out, long long* cyc) {
    for (int waves : {1, 2}) {
        const int iters = 4000;
        k<OP><<<256, 64 * 4 * waves>>>(out, iters, cyc); hipDeviceSynchronize();
        k<OP><<<256, 64 * 4 * waves>>>(out, iters, cyc); hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s %d wave(s)/SIMD: %5.1f cycles per instruction group per wave, %5.1f per SIMD slot\n", name, waves, (double)c / (8.0 * iters), (double)c / (8.0 * iters * waves));
    }
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 4 * (1 + 256 * 1024)); hipMalloc(&cyc, 8); hipMemset(out, 0, 4);
    run<0>("v_add_f32", out, cyc); run<1>("v_pk_add_f32", out, cyc); run<2>("v_cvt_pk_bf16_f32 + lshl + add", out, cyc);
    run<3>("v_mov_b32_dpp wave_shr:1", out, cyc); run<6>("v_mov_b32_dpp row_shr:1", out, cyc); run<4>("v_cndmask_b32", out, cyc); run<5>("v_and_b32", out, cyc);
    return 0;
}
