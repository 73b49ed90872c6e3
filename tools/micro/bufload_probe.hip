// What do raw buffer loads of 8 / 16 bytes return at 4-byte-aligned (not naturally aligned) offsets, and at offsets
// whose tail crosses num_records? build: hipcc --offload-arch=gfx950 -O2 bufload_probe.hip -o bufload_probe
// Loads go through the LLVM intrinsics directly: hipcc 7.2 narrows element reads of __builtin_amdgcn_raw_buffer_load_b64 /
// _b128 to ONE dword load (check with -S), so the builtins cannot answer the question.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ f32x2 load2(i32x4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ f32x4 load4(i32x4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__global__ void probe(const float* p, int bytes, float* out, int word3) {
    i32x4 rs;
    rs[0] = (int)(unsigned long long)p; rs[1] = (int)(((unsigned long long)p >> 32) & 0xffff); rs[2] = bytes; rs[3] = word3;
    const int lane = threadIdx.x;
    const f32x4 q = load4(rs, lane * 4, 0, 0);   // offsets 0,4,8,...: mostly unaligned
    const f32x2 h = load2(rs, lane * 4, 0, 0);
    for (int k = 0; k < 4; ++k) out[lane * 8 + k] = q[k];
    for (int k = 0; k < 2; ++k) out[lane * 8 + 4 + k] = h[k];
}
int main() {
    const int n = 70;
    float h[n], *d, *o, ho[64 * 8];
    for (int i = 0; i < n; ++i) h[i] = 100.f + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int word3 : {0x00020000}) {
    printf("descriptor word 3 = 0x%08x\n", word3);
    probe<<<1, 64>>>(d, 64 * 4, o, word3);   // num_records = 256 bytes: elements 0..63 in range
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 5, 60, 61, 62, 63})
        printf("lane %2d (byte offset %3d): b128 = %5.0f %5.0f %5.0f %5.0f   b64 = %5.0f %5.0f\n", l, l * 4, ho[l * 8], ho[l * 8 + 1],
               ho[l * 8 + 2], ho[l * 8 + 3], ho[l * 8 + 4], ho[l * 8 + 5]);
  }
    return 0;
}
