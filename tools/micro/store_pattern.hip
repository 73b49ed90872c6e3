// How fast can the configs[1] result tensor (N x 64 planes x 224 x 224 fp32 = 1.64 GB) be WRITTEN, as a function of how a
// wave's 64 x 32 (filters x pixels) MFMA tile is laid out over its store instructions? No arithmetic, no loads: the walk
// over tiles is the forward kernel's (a workgroup of 4 waves per 8-row strip of one image, 32-pixel tiles), only the
// stores differ:
//   A  32 x buffer_store_dword : lane = pixel, two filter planes per instruction (2 x 128 B runs)   [what the kernel does]
//   B   8 x 16-byte stores     : lane = filter (the swapped-operand MFMA layout), 4 consecutive pixels per lane
//                                (64 pieces of 16 B over 32 planes per instruction)
//   C   8 x 16-byte stores     : lane = 4 pixels of one plane, 8 planes per instruction (8 x 128 B runs; needs a
//                                transpose the MFMA layout does not give for free)
//   D  plain fill of the same bytes, 16 B per lane, contiguous (the ceiling)
// build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern ; run: ./store_pattern [nt=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 128, F = 64, H = 224, W = 224, R = 8, HW = H * W;

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k(float* __restrict__ y) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const int n = blockIdx.x / (H / R), oh0 = (blockIdx.x % (H / R)) * R;
    float* img = y + (size_t)n * F * HW + (size_t)oh0 * W;
    for (int t = wid; t < R * (W / 32); t += 4) {
        const int row = t / (W / 32), ct = t % (W / 32);
        float* tile = img + row * W + ct * 32;
        if (PAT == 0) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float* p = tile + (size_t)f * HW + l31;
                    if (NT) __builtin_nontemporal_store((float)r, p); else *p = (float)r;
                }
        } else if (PAT == 1) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f4* p = reinterpret_cast<f4*>(tile + (size_t)(tm * 32 + l31) * HW + 8 * q + 4 * hi);
                    const f4 v = {(float)q, 1.f, 2.f, 3.f};
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
        } else if (PAT == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f4* p = reinterpret_cast<f4*>(tile + (size_t)(q * 8 + (lane >> 3)) * HW + 4 * (lane & 7));
                const f4 v = {(float)q, 1.f, 2.f, 3.f};
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void kfill(f4* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4 v = {1.f, 2.f, 3.f, 4.f};
        if (NT) __builtin_nontemporal_store(v, y + i); else y[i] = v;
    }
}
int main(int argc, char** argv) {
    const bool nt = argc < 2 || atoi(argv[1]) != 0;
    const size_t n = (size_t)N * F * HW;
    float* y;
    CK(hipMalloc(&y, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = N * (H / R);
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (pat == 0) { if (nt) k<0, true><<<blocks, 256>>>(y); else k<0, false><<<blocks, 256>>>(y); }
            if (pat == 1) { if (nt) k<1, true><<<blocks, 256>>>(y); else k<1, false><<<blocks, 256>>>(y); }
            if (pat == 2) { if (nt) k<2, true><<<blocks, 256>>>(y); else k<2, false><<<blocks, 256>>>(y); }
            if (pat == 3) { if (nt) kfill<true><<<256 * 8, 256>>>((f4*)y, n / 4); else kfill<false><<<256 * 8, 256>>>((f4*)y, n / 4); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("pattern %c %s: %.3f ms  %.2f TB/s\n", "ABCD"[pat], nt ? "nt" : "plain", best, n * 4 / best / 1e9);
    }
    return 0;
}
