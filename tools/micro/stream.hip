// What do plain streaming kernels reach on this box? copy (1 read + 1 write), scale-add (2 reads + 1 write), read-only sum,
// write-only fill -- float4 per lane, grid-stride, with plain / non-temporal accesses and several grid sizes.
// build: hipcc --offload-arch=gfx950 -O3 stream.hip -o stream ; run: ./stream [MiB per tensor, default 392]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void krand(unsigned* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = 0x3f000000u | ((unsigned)(i * 2654435761u) >> 9);  // floats in [0.5, 1)
}
template <int MODE, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t n4, float* sink) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
        f4 va[UNROLL], vb[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n4) {
                if (MODE != 3) va[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
                if (MODE == 1) vb[u] = NT ? __builtin_nontemporal_load(b + j) : b[j];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n4) {
                f4 r;
                if (MODE == 0) r = va[u];
                else if (MODE == 1) r = va[u] * 1.5f + vb[u];
                else if (MODE == 2) { acc += va[u]; continue; }
                else r = (f4){1.f, 2.f, 3.f, 4.f};
                if (NT) __builtin_nontemporal_store(r, c + j); else c[j] = r;
            }
        }
    }
    if (MODE == 2 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) *sink = 1.f;
}
// the shape of plane_map_kernel (chan_reduce.h): one short-lived workgroup per 4096 elements, per thread 4 x (load 16 B,
// store 16 B) -- BATCH: all four loads before the first store (what a `restrict`-free in-place body cannot be given)
template <bool BATCH>
__global__ __launch_bounds__(256) void kplane(const f4* a, f4* c, size_t n4) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (BATCH) {
        f4 v[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) if (base + it * 256 < n4) v[it] = a[base + it * 256];
#pragma unroll
        for (int it = 0; it < 4; ++it) if (base + it * 256 < n4) c[base + it * 256] = v[it] * 1.5f;
    } else {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (base + it * 256 < n4) {
                f4 v = a[base + it * 256];
                asm volatile("" ::: "memory");
                c[base + it * 256] = v * 1.5f;
                asm volatile("" ::: "memory");
            }
    }
}
template <bool BATCH>
static void run_plane(const f4* a, f4* c, size_t n4, double bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n4 + 1023) / 1024);
    for (int w = 0; w < 2; ++w) kplane<BATCH><<<grid, 256>>>(a, c, n4);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) kplane<BATCH><<<grid, 256>>>(a, c, n4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("plane-style copy, one WG per 4096 elements, batch=%d        : %7.3f ms  %6.2f TB/s\n", (int)BATCH, ms, bytes / ms / 1e9);
}
// per-channel sums over an NCHW tensor the way chan_reduce_partial walks it: workgroup (c, sp) reads the planes of
// channel c of `per` images -- BLOCKED: images sp*per .. sp*per+per-1 (1024 far-apart streams), else images sp, sp+splits,
// ... (all workgroups sweep one window of memory together)
template <bool BLOCKED>
__global__ __launch_bounds__(256) void kchan(const f4* a, int C, int HW4, int N, int splits, float* sink) {
    const int c = blockIdx.x % C, sp = blockIdx.x / C;
    const int per = N / splits;
    f4 acc = {0, 0, 0, 0};
    for (int k = 0; k < per; ++k) {
        const int n = BLOCKED ? sp * per + k : sp + k * splits;
        const f4* p = a + ((size_t)n * C + c) * HW4;
        for (int i = threadIdx.x; i < HW4; i += 1024) {
            f4 v0 = p[i], v1 = {0, 0, 0, 0}, v2 = v1, v3 = v1;
            if (i + 256 < HW4) v1 = p[i + 256];
            if (i + 512 < HW4) v2 = p[i + 512];
            if (i + 768 < HW4) v3 = p[i + 768];
            acc += (v0 + v1) + (v2 + v3);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) *sink = 1.f;
}
// the loop of chan_reduce_partial as shipped until round 2: flat index over the channel's N*HW elements, per-thread carries
template <int VARIANT>
__global__ __launch_bounds__(256) void kflat(const float* x, int C, int HW, int M, int splits, float* partials) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, sp = blockIdx.y;
    const int per = (((M + splits - 1) / splits) + 3) & ~3;
    const int lo = sp * per;
    int hi = lo + per;
    if (hi > M) hi = M;
    float acc[2] = {0.f, 0.f};
    int idx = lo + threadIdx.x * 4;
    if (idx < hi) {
        const int n0 = idx / HW;
        int i = idx - n0 * HW;
        long long base = ((long long)n0 * C + c) * HW;
        const long long img = (long long)C * HW;
        if (VARIANT == 0) {
#pragma unroll 4
            for (; idx < hi; idx += 256 * 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + base + i);
                acc[0] += (v.x + v.y) + (v.z + v.w);
                acc[1] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                i += 256 * 4;
                while (i >= HW) { i -= HW; base += img; }
            }
        } else {  // same walk, the squares left out (is it the arithmetic?)
#pragma unroll 4
            for (; idx < hi; idx += 256 * 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + base + i);
                acc[0] += (v.x + v.y) + (v.z + v.w);
                i += 256 * 4;
                while (i >= HW) { i -= HW; base += img; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int v = 0; v < 2; ++v) {
        float t = acc[v];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0) red[wid][v] = t;
    }
    __syncthreads();
    if (threadIdx.x < 2) partials[((long long)c * splits + sp) * 2 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <int VARIANT>
static void run_flat(const f4* a, int N, int C, int HW, int splits, float* part) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid(C, splits);
    for (int w = 0; w < 2; ++w) kflat<VARIANT><<<grid, 256>>>((const float*)a, C, HW, N * HW, splits, part);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) kflat<VARIANT><<<grid, 256>>>((const float*)a, C, HW, N * HW, splits, part);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("flat-index channel sums (shipped loop) N=%d C=%d HW=%d splits=%d variant=%d : %7.3f ms  %6.2f TB/s\n", N, C, HW, splits,
           VARIANT, ms, (double)N * C * HW * 4 / ms / 1e9);
}
template <bool BLOCKED>
static void run_chan(const f4* a, int N, int C, int HW, int splits, float* sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) kchan<BLOCKED><<<C * splits, 256>>>(a, C, HW / 4, N, splits, sink);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) kchan<BLOCKED><<<C * splits, 256>>>(a, C, HW / 4, N, splits, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("channel sums N=%d C=%d HW=%d splits=%d blocked=%d : %7.3f ms  %6.2f TB/s\n", N, C, HW, splits, (int)BLOCKED, ms,
           (double)N * C * HW * 4 / ms / 1e9);
}
template <int MODE, bool NT, int UNROLL>
static void run(const char* name, const f4* a, const f4* b, f4* c, size_t n4, float* sink, int wg_per_cu, double bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    for (int w = 0; w < 2; ++w) k<MODE, NT, UNROLL><<<grid, 256>>>(a, b, c, n4, sink);
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int w = 0; w < it; ++w) k<MODE, NT, UNROLL><<<grid, 256>>>(a, b, c, n4, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    printf("%-28s nt=%d unroll=%d wg/cu=%2d : %7.3f ms  %6.2f TB/s\n", name, (int)NT, UNROLL, wg_per_cu, ms, bytes / ms / 1e9);
}
int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? atoi(argv[1]) : 392;
    const size_t n4 = mib * (1 << 20) / 16;
    f4 *a, *b, *c; float* sink;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&c, n4 * 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
    if (argc > 2) {  // any second argument: pseudo-random contents instead of zeros
        krand<<<4096, 256>>>((unsigned*)a, n4 * 4); krand<<<4096, 256>>>((unsigned*)b, n4 * 4);
        CK(hipDeviceSynchronize());
        printf("contents: pseudo-random\n");
    }
    const double B = (double)n4 * 16;
    printf("tensor = %zu MiB\n", mib);
    {   // a read-only pass right after a kernel that WROTE a big tensor: the dirty lines that kernel left in L2 / Infinity
        // Cache are written back while this one reads, so its apparent rate drops although HBM is just as busy
        hipEvent_t ev[21]; for (auto& x : ev) CK(hipEventCreate(&x));
        const int grid = 256 * 8;
        for (int w = 0; w < 2; ++w) { k<3, false, 4><<<grid, 256>>>(a, b, c, n4, sink); k<2, false, 4><<<grid, 256>>>(a, b, c, n4, sink); }
        CK(hipEventRecord(ev[0]));
        for (int w = 0; w < 10; ++w) {
            k<3, false, 4><<<grid, 256>>>(a, b, c, n4, sink);  // fill c
            CK(hipEventRecord(ev[2 * w + 1]));
            k<2, false, 4><<<grid, 256>>>(a, b, c, n4, sink);  // sum a
            CK(hipEventRecord(ev[2 * w + 2]));
        }
        CK(hipEventSynchronize(ev[20]));
        float tf = 0, ts = 0, ms;
        for (int w = 0; w < 10; ++w) {
            CK(hipEventElapsedTime(&ms, ev[2 * w], ev[2 * w + 1])); tf += ms;
            CK(hipEventElapsedTime(&ms, ev[2 * w + 1], ev[2 * w + 2])); ts += ms;
        }
        printf("alternating fill(c) / sum(a): fill %7.3f ms %6.2f TB/s | sum %7.3f ms %6.2f TB/s (apparent)\n", tf / 10, B / (tf / 10) / 1e9,
               ts / 10, B / (ts / 10) / 1e9);
    }
    if (mib >= 392) {
        float* part; CK(hipMalloc(&part, 1 << 20));
        run_flat<0>(a, 128, 64, 12544, 16, part); run_flat<1>(a, 128, 64, 12544, 16, part);
        run_flat<0>(a, 128, 64, 3136, 16, part);
        run_chan<true>(a, 128, 64, 12544, 16, sink); run_chan<false>(a, 128, 64, 12544, 16, sink);
        run_chan<true>(a, 128, 64, 12544, 32, sink); run_chan<false>(a, 128, 64, 12544, 32, sink);
        run_chan<true>(a, 128, 64, 3136, 16, sink); run_chan<false>(a, 128, 64, 3136, 16, sink);
        run_chan<true>(a, 256, 64, 3136, 16, sink); run_chan<false>(a, 256, 64, 3136, 16, sink);
    }
    run_plane<false>(a, c, n4, 2 * B);
    run_plane<true>(a, c, n4, 2 * B);
    for (int wg : {8, 32}) {
        run<0, false, 1>("copy (1R+1W)", a, b, c, n4, sink, wg, 2 * B);
        run<0, false, 4>("copy (1R+1W)", a, b, c, n4, sink, wg, 2 * B);
        run<0, true, 4>("copy (1R+1W)", a, b, c, n4, sink, wg, 2 * B);
        run<1, false, 4>("scale-add (2R+1W)", a, b, c, n4, sink, wg, 3 * B);
        run<1, true, 4>("scale-add (2R+1W)", a, b, c, n4, sink, wg, 3 * B);
        run<2, false, 4>("sum (1R)", a, b, c, n4, sink, wg, B);
        run<2, true, 4>("sum (1R)", a, b, c, n4, sink, wg, B);
        run<3, false, 4>("fill (1W)", a, b, c, n4, sink, wg, B);
        run<3, true, 4>("fill (1W)", a, b, c, n4, sink, wg, B);
    }
    return 0;
}
