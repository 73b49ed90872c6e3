// Infinity Cache (256 MB, memory side) probe: what does a consumer kernel see of a tensor the producer kernel has just written?
// For buffer sizes 16 MB ... 512 MB:  (a) a float4 read sweep repeated on the same buffer (resident if it fits),
// (b) write sweep (plain stores) then read sweep, (c) write sweep with non-temporal stores then read sweep,
// (d) read sweep of the buffer right after a read sweep of ANOTHER buffer of 512 MB (cold).  Reports the READ sweep's GB/s.
// build: hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_sweep(const float4* __restrict__ p, size_t n4, float* sink) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) *sink = s;
}
template <bool NT>
__global__ __launch_bounds__(256) void write_sweep(float4* __restrict__ p, size_t n4, float val) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = {val, val + 1.f, val + 2.f, val + 3.f};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p) + i); else reinterpret_cast<f4*>(p)[i] = v;
    }
}

int main() {
    const size_t big = 512ull << 20;
    float4 *a, *other; float* sink;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&other, big)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, big)); CK(hipMemset(other, 0, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    auto timed_read = [&](const float4* p, size_t n4) -> float {
        hipEventRecord(e0); read_sweep<<<grid, 256>>>(p, n4, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
    };
    // warm clocks
    for (int i = 0; i < 50; ++i) read_sweep<<<grid, 256>>>(other, big / 16, sink);
    CK(hipDeviceSynchronize());
    printf("%8s %14s %14s %14s %14s   (GB/s of the read sweep, median of 7)\n", "MB", "re-read", "after write", "after nt write", "cold");
    for (size_t mb : {16, 32, 64, 103, 128, 192, 256, 384, 512}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        std::vector<float> r[4];
        for (int rep = 0; rep < 7; ++rep) {
            read_sweep<<<grid, 256>>>(a, n4, sink);                      // (a) second read of the same buffer
            r[0].push_back(timed_read(a, n4));
            write_sweep<false><<<grid, 256>>>(a, n4, (float)rep);        // (b)
            r[1].push_back(timed_read(a, n4));
            write_sweep<true><<<grid, 256>>>(a, n4, (float)rep);         // (c)
            r[2].push_back(timed_read(a, n4));
            read_sweep<<<grid, 256>>>(other, big / 16, sink);            // (d) evict with 512 MB of something else
            r[3].push_back(timed_read(a, n4));
        }
        printf("%8zu", mb);
        for (int k = 0; k < 4; ++k) { std::sort(r[k].begin(), r[k].end()); printf(" %14.0f", bytes / (r[k][3] * 1e-3) / 1e9); }
        printf("\n");
    }
    return 0;
}
