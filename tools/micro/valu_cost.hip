// Issue cost on gfx950 of the individual instructions the Winograd produce step is made of, measured on exactly the
// instruction named (inline assembly, 16 independent destinations per loop trip, s_memtime around 2000 trips; one and two
// waves per SIMD). valu_rate.hip lets the compiler choose the instructions and so also counts the moves / s_nops it adds
// around DPP operations and selects. build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void k(float* out, int iters, long long* cyc) {
    __shared__ float lds[4096];
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.5f + i; b[i] = a[i] * 3.f; }
    lds[threadIdx.x] = a[0];
    __syncthreads();
    const unsigned addr = ((threadIdx.x + 1) & 63) * 4;
    const unsigned laddr = (threadIdx.x & 63) * 4;
    unsigned long long mask = 0x5555555555555555ull;
    asm volatile("" : "+s"(mask));
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define PKADD(i) if (i < 8) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(*(double*)&a[2 * (i)]) : "v"(*(double*)&b[2 * (i)]));
#define PKADDNEG(i) if (i < 8) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "+v"(*(double*)&a[2 * (i)]) : "v"(*(double*)&b[2 * (i)]));
#define CND(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask));
#define CND0(i) asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(a[i]) : "s"(mask));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
#define DPPROW(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
#define SUBDPP(i) asm volatile("v_sub_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
#define AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a[i]) : "v"(addr), "v"(b[i]));
#define SWZ(i) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM,\"01pip\")" : "=v"(a[i]) : "v"(b[i]));
#define DSW2(i) asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:8" : : "v"(laddr), "v"(a[i]), "v"(b[i]) : "memory");
#define DSR(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(laddr));
        if (OP == 0) { REP16(ADD) }
        if (OP == 1) { REP16(PKADD) }
        if (OP == 2) { REP16(CND) }
        if (OP == 3) { REP16(DPP) }
        if (OP == 4) { REP16(BFI) }
        if (OP == 5) { REP16(AND) }
        if (OP == 6) { REP16(BPERM) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 7) { REP16(DSW2) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 8) { REP16(SUBDPP) }
        if (OP == 9) { REP16(CND0) }
        if (OP == 10) { REP16(DPPROW) }
        if (OP == 11) { REP16(PKADDNEG) }
        if (OP == 12) { REP16(SWZ) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 13) { REP16(DSR) asm volatile("s_waitcnt lgkmcnt(0)"); }
    }
    const long long t1 = clock64();
    float r = 0; for (int i = 0; i < 16; ++i) r += a[i] + b[i];
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = r + lds[(threadIdx.x * 7) & 4095];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, int per_trip, float* out, long long* cyc) {
    for (int waves : {1, 2}) {
        const int iters = 2000;
        k<OP><<<256, 64 * 4 * waves>>>(out, iters, cyc); hipDeviceSynchronize();
        k<OP><<<256, 64 * 4 * waves>>>(out, iters, cyc); hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %d wave(s)/SIMD: %6.2f ticks per instruction and wave, %6.2f per SIMD\n", name, waves,
               (double)c / ((double)per_trip * iters), (double)c / ((double)per_trip * iters * waves));
    }
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 4 * (1 + 256 * 1024)); hipMalloc(&cyc, 8); hipMemset(out, 0, 4);
    run<0>("v_add_f32", 16, out, cyc);
    run<1>("v_pk_add_f32", 8, out, cyc);
    run<11>("v_pk_add_f32 op_sel / neg_lo", 8, out, cyc);
    run<2>("v_cndmask_b32_e64 v, v, sgpr-pair", 16, out, cyc);
    run<9>("v_cndmask_b32_e64 v, 0, sgpr-pair", 16, out, cyc);
    run<3>("v_mov_b32_dpp wave_shr:1", 16, out, cyc);
    run<10>("v_mov_b32_dpp row_shr:1", 16, out, cyc);
    run<8>("v_sub_f32_dpp wave_shr:1", 16, out, cyc);
    run<4>("v_bfi_b32", 16, out, cyc);
    run<5>("v_and_b32", 16, out, cyc);
    run<6>("ds_bpermute_b32", 16, out, cyc);
    run<12>("ds_swizzle_b32", 16, out, cyc);
    run<7>("ds_write2st64_b32", 16, out, cyc);
    run<13>("ds_read_b32", 16, out, cyc);
    return 0;
}
