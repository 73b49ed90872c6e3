// lds_dma.h -- global -> LDS direct loads (`buffer_load_dword ... lds`) for the DMA-staged GEMM kernels.
#pragma once
#include "common.h"

namespace bcnn_hip {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef int rsrc_i4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOOB = 0x80000000u;  // buffers are limited to < 2 GiB so this voffset is always out of range

// raw buffer descriptor (stride 0, no swizzle): range check on voffset + inst_offset only, soffset is unchecked
__device__ __forceinline__ rsrc_i4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    rsrc_i4 r;
    r[0] = (int)(unsigned)u; r[1] = (int)(unsigned)((u >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}

// 4 / 8 bytes per lane into registers, out-of-range voffset -> 0.0. Declared on the LLVM intrinsic because hipcc 7.2
// narrows element reads of `__builtin_amdgcn_raw_buffer_load_b64/_b128` to one dword (tools/micro/bufload_probe.hip).
typedef float buf_f32x2 __attribute__((ext_vector_type(2)));
__device__ float buffer_load_f32(rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ buf_f32x2 buffer_load_f32x2(rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
// stores with an out-of-range voffset are dropped: masking by address instead of by branch
__device__ void buffer_store_f32(float v, rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void buffer_store_f32x2(buf_f32x2 v, rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
typedef float buf_f32x4 __attribute__((ext_vector_type(4)));
__device__ buf_f32x4 buffer_load_f32x4(rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
// NOTE: there is deliberately no 16-byte buffer STORE here. hipcc 7.2 schedules a vector-ALU write of the data registers
// directly behind a `buffer_store_dwordx4` on gfx950 without the wait states the hardware needs, and the store then sends
// the NEW register contents for its last lanes (seen as element 0 of lanes 12-15 of every 16 replaced by the next
// instruction's voffset). A 16-byte store has to be inline assembly that carries its own `s_nop 1`.

__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(size_t)(lds_void_ptr)p; }

// One LDS-DMA slab: 64 lanes x 4 B from rsrc[voff + soff] to LDS[lds_base + 4*lane]; an out-of-range voff
// deposits 0.0. Written as inline assembly on purpose: the compiler's waitcnt pass cannot tell the two LDS
// buffers apart and would put `s_waitcnt vmcnt(0)` in front of the first ds_read of the tile being
// multiplied, i.e. serialise the prefetch with the MFMAs. The loops drain vmcnt themselves right before
// their barrier (dma_wait).
__device__ __forceinline__ void dma_row(rsrc_i4 rs, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :
                 : "s"(lds_base), "v"(voff), "s"(rs), "s"(soff)
                 : "memory", "m0");
}
// 16 bytes per lane: 64 lanes x 16 B = 1 KiB of LDS at lds_base + 16*lane (gfx950 `buffer_load_dwordx4 ... lds`).
__device__ __forceinline__ void dma_row_x4(rsrc_i4 rs, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_base), "v"(voff), "s"(rs), "s"(soff)
                 : "memory", "m0");
}
// the same with the non-temporal hint: a stream that is read exactly once should not displace what other kernels keep in
// L2 / Infinity Cache
__device__ __forceinline__ void dma_row_x4_nt(rsrc_i4 rs, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                 :
                 : "s"(lds_base), "v"(voff), "s"(rs), "s"(soff)
                 : "memory", "m0");
}
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. every global load the wave
// has in flight -- fatal for a pipeline that keeps operand requests flying across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's loads are outstanding (vmcnt retires in order: the N youngest may fly)
template <int N>
__device__ __forceinline__ void dma_wait_n() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// n / d via multiply-high with magic = ceil(2^32 / d) (exact while n * d < 2^32); d == 1 is encoded as magic 0.
__device__ __forceinline__ unsigned magic_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

}  // namespace bcnn_hip
