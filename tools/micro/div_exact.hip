// Is  q1 = fma(fma(-b, q0, a), r, q0),  q0 = a * r,  r = RN(1 / b)  the correctly rounded a / b?  (Markstein's correction step:
// three instructions instead of the ~12 of an IEEE division, for a divisor that is constant per channel.)  Counts mismatches
// against __fdiv_rn over random operands, divisors with all-ones / all-zero / alternating significands, and tiny / huge
// quotients.  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_exact div_exact.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__global__ void k(unsigned long long* bad, unsigned long long* tot, int mode, uint32_t seed, float* ex) {
    uint32_t s = seed * 2654435761u + (blockIdx.x * blockDim.x + threadIdx.x) * 40503u + 1u;
    unsigned long long nb = 0, nt = 0;
    for (int it = 0; it < 4096; ++it) {
        uint32_t mb = rng(s) & 0x7fffffu, ma = rng(s) & 0x7fffffu;
        int eb = 127 + (int)(rng(s) % 41) - 20, ea = 127 + (int)(rng(s) % 81) - 40;
        if (mode == 1) mb = 0x7fffffu;                       // divisor significand all ones
        if (mode == 2) mb = 0;                               // power of two
        if (mode == 3) mb = (it & 1) ? 0x555555u : 0x2aaaaau;
        if (mode == 4) { mb = 0x7fffffu - (rng(s) & 0xffu); }  // near all ones
        if (mode == 5) { mb = rng(s) & 0xffu; }                // near power of two
        if (mode == 6) { ea = 127 + (int)(rng(s) % 201) - 100; }  // wide quotient range
        const float b = __uint_as_float(((uint32_t)eb << 23) | mb) * ((rng(s) & 1) ? 1.f : -1.f);
        const float a = __uint_as_float(((uint32_t)ea << 23) | ma) * ((rng(s) & 1) ? 1.f : -1.f);
        const float r = __fdiv_rn(1.0f, b);
        const float q0 = __fmul_rn(a, r);
        const float e = __fmaf_rn(-b, q0, a);
        const float q1 = __fmaf_rn(e, r, q0);
        const float q = __fdiv_rn(a, b);
        ++nt;
        if (__float_as_uint(q) != __float_as_uint(q1)) { if (!nb) { ex[0] = a; ex[1] = b; ex[2] = q; ex[3] = q1; } ++nb; }
    }
    atomicAdd(bad, nb); atomicAdd(tot, nt);
}
int main() {
    unsigned long long *bad, *tot; float* ex;
    hipMalloc(&bad, 8); hipMalloc(&tot, 8); hipMalloc(&ex, 16);
    for (int mode = 0; mode <= 6; ++mode) {
        hipMemset(bad, 0, 8); hipMemset(tot, 0, 8);
        for (uint32_t seed = 1; seed <= 8; ++seed) k<<<4096, 256>>>(bad, tot, mode, seed, ex);
        unsigned long long hb, ht; float hex[4];
        hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, tot, 8, hipMemcpyDeviceToHost); hipMemcpy(hex, ex, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches in %llu", mode, hb, ht);
        if (hb) printf("  e.g. a=%a b=%a div=%a corr=%a", hex[0], hex[1], hex[2], hex[3]);
        printf("\n");
    }
    return 0;
}
