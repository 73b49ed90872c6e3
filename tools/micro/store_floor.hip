// Where is the WRITE-stream floor of the configs[1] result tensor (128 x 64 x 224 x 224 fp32 = 1.644 GB) on this part, and
// which store layouts reach it?  VERDICT r4 item 1: hipMemset reached 6.7 TB/s where store_pattern.hip's "plain fill" managed
// 4.75 -- so the micro-benchmark, not the chip, set the floor the forward kernel was compared with.  This file measures:
//   M   hipMemsetAsync / hipMemsetD32Async of the tensor (what __amd_rocclr_fillBufferAligned does)
//   G   grid-stride fills: blocks x unroll x {plain, nt}, 16 B per lane
//   S   span fills: each workgroup owns ONE contiguous span (4 KB ... 448 KB) and walks it 4 KB (256 lanes x 16 B) at a time
//   K   the forward kernel's geometry (a workgroup = 8 output rows of one image = a 7 KB run in each of the 64 filter planes)
//       drained with 16-byte stores in several orders:
//        K0  today's kernel: 32-pixel tiles, dword stores, two planes per instruction (pattern A of store_pattern.hip)
//        K1  wave w owns planes w, w+4, ...: each plane's 7 KB strip as 7 x 1 KB instructions, plane after plane
//        K2  wave w owns planes 16w .. 16w+15 the same way
//        K3  row by row: all 64 planes of output row r (896 B runs, 56 lanes x 16 B), then row r + 1   [LDS drain per row]
//        K4  two rows at a time (1792 B runs)
//        K5  half rows of 112 pixels (448 B runs)                                                    [LDS drain per half row]
//       each with the strips in blockIdx order (n-major) and persistent (grid = 256 * k workgroups looping over strips)
// build: hipcc --offload-arch=gfx950 -O3 store_floor.hip -o store_floor ; run: ./store_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 128, F = 64, H = 224, W = 224, R = 8, HW = H * W;
constexpr size_t TOTAL = (size_t)N * F * HW;

template <bool NT> __device__ __forceinline__ void st16(f4* p, f4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void kgrid(f4* __restrict__ y, size_t n4) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (i + u * 256 < n4) st16<NT>(y + i + u * 256, v);
    }
}
// each workgroup owns span4 float4s (contiguous)
template <bool NT>
__global__ __launch_bounds__(256) void kspan(f4* __restrict__ y, size_t n4, unsigned span4) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t wg = blockIdx.x; wg * span4 < n4; wg += gridDim.x) {
        f4* p = y + wg * span4;
        for (unsigned i = threadIdx.x; i < span4; i += 256) st16<NT>(p + i, v);
    }
}

// the forward kernel's geometry
template <int PAT, bool NT>
__device__ __forceinline__ void strip_store(float* __restrict__ y, int sidx) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = sidx / (H / R), oh0 = (sidx % (H / R)) * R;
    float* img = y + (size_t)n * F * HW + (size_t)oh0 * W;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    if (PAT == 0) {
        const int l31 = lane & 31, hi = lane >> 5;
        for (int t = wid; t < R * (W / 32); t += 4) {
            const int row = t / (W / 32), ct = t % (W / 32);
            float* tile = img + row * W + ct * 32;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float* p = tile + (size_t)f * HW + l31;
                    if (NT) __builtin_nontemporal_store((float)r, p); else *p = (float)r;
                }
        }
    } else if (PAT == 1 || PAT == 2) {
        for (int i = 0; i < 16; ++i) {
            const int f = (PAT == 1) ? wid + 4 * i : wid * 16 + i;
            f4* p = reinterpret_cast<f4*>(img + (size_t)f * HW);
#pragma unroll
            for (int j = 0; j < R * W / 256; ++j) st16<NT>(p + j * 64 + lane, v);
        }
    } else if (PAT == 3 || PAT == 4 || PAT == 5) {
        // run = RUN floats per plane and step; 256 threads cover 64 planes x RUN floats as flattened 16-byte pieces
        constexpr int RUN = (PAT == 3) ? W : (PAT == 4 ? 2 * W : W / 2);
        constexpr int P4 = RUN / 4;               // 16-byte pieces per plane and step
        constexpr int STEPS = R * W / RUN;
        for (int sstep = 0; sstep < STEPS; ++sstep) {
            float* base = img + sstep * RUN;
            for (int i = tid; i < F * P4; i += 256) {
                const int f = i / P4, j = i - f * P4;
                st16<NT>(reinterpret_cast<f4*>(base + (size_t)f * HW) + j, v);
            }
        }
    }
}
template <int PAT, bool NT>
__global__ __launch_bounds__(256) void kstrip(float* __restrict__ y, int nstrips) {
    for (int sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) strip_store<PAT, NT>(y, sidx);
}

// cache-policy bits of the buffer stores (gfx940+: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1), on the K0 / K1 walks
typedef int rsrc_i4 __attribute__((ext_vector_type(4)));
__device__ void buffer_store_f32(float v, rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void buffer_store_f32x4(f4 v, rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
template <int PAT, int AUX>
__global__ __launch_bounds__(256) void kaux(float* __restrict__ y, int nstrips) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned long long u = (unsigned long long)y;
    rsrc_i4 rs;
    rs[0] = (int)(unsigned)u; rs[1] = (int)(unsigned)((u >> 32) & 0xffffu); rs[2] = 0x7ffffffc; rs[3] = 0x00020000;
    for (int sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) {
        const int n = sidx / (H / R), oh0 = (sidx % (H / R)) * R;
        const unsigned img = ((unsigned)n * F * HW + (unsigned)oh0 * W) * 4u;
        if (PAT == 0) {
            const int l31 = lane & 31, hi = lane >> 5;
            for (int t = wid; t < R * (W / 32); t += 4) {
                const int row = t / (W / 32), ct = t % (W / 32);
                const unsigned tile = img + (unsigned)(row * W + ct * 32 + l31 + 4 * hi * HW) * 4u;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        buffer_store_f32((float)r, rs, (int)tile, (tm * 32 + (r & 3) + 8 * (r >> 2)) * HW * 4, AUX);
            }
        } else {
            const f4 v = {1.f, 2.f, 3.f, 4.f};
            for (int i = 0; i < 16; ++i) {
                const int f = wid + 4 * i;
                const unsigned p = img + (unsigned)f * HW * 4u + lane * 16u;
#pragma unroll
                for (int j = 0; j < R * W / 256; ++j) buffer_store_f32x4(v, rs, (int)p, j * 1024, AUX);
            }
        }
    }
}

// K6: the chip sweeps the tensor row by row: unit = RU output rows of one image (all 64 planes), WG = 7 waves (one per
// 32-pixel column tile), units dealt out interleaved (unit = it * grid + wg) so the active window is grid * RU rows
template <int RU, int AUX>
__global__ __launch_bounds__(448) void krows(float* __restrict__ y, int nunits) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned long long u = (unsigned long long)y;
    rsrc_i4 rs;
    rs[0] = (int)(unsigned)u; rs[1] = (int)(unsigned)((u >> 32) & 0xffffu); rs[2] = 0x7ffffffc; rs[3] = 0x00020000;
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int UPI = H / RU;  // units per image
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const int n = unit / UPI, oh0 = (unit % UPI) * RU;
        for (int row = 0; row < RU; ++row) {
            const unsigned tile = ((unsigned)n * F * HW + (unsigned)((oh0 + row) * W + wid * 32 + l31 + 4 * hi * HW)) * 4u;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buffer_store_f32((float)r, rs, (int)tile, (tm * 32 + (r & 3) + 8 * (r >> 2)) * HW * 4, AUX);
        }
    }
}

// K1 walk with a padded plane stride (bytes): is the slowness of "slow" placements an aliasing of the 200704-byte stride?
template <int AUX>
__global__ __launch_bounds__(256) void kstride(float* __restrict__ y, int nstrips, unsigned plane_bytes) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned long long u = (unsigned long long)y;
    rsrc_i4 rs;
    rs[0] = (int)(unsigned)u; rs[1] = (int)(unsigned)((u >> 32) & 0xffffu); rs[2] = 0x7ffffffc; rs[3] = 0x00020000;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (int sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) {
        const int n = sidx / (H / R), oh0 = (sidx % (H / R)) * R;
        const unsigned img = (unsigned)n * F * plane_bytes + (unsigned)oh0 * W * 4u;
        for (int i = 0; i < 16; ++i) {
            const int f = wid + 4 * i;
            const unsigned p = img + (unsigned)f * plane_bytes + lane * 16u;
#pragma unroll
            for (int j = 0; j < R * W / 256; ++j) buffer_store_f32x4(v, rs, (int)p, j * 1024, AUX);
        }
    }
}

// K1 walk over a tensor whose even / odd strips live in two different places (theory: "fast" placements are buffers whose
// pieces lie in distant physical regions, so that concurrent streams meet more DRAM ranks / banks)
__global__ __launch_bounds__(256) void ksplit(float* __restrict__ ya, float* __restrict__ yb, int nstrips) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (int sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) {
        float* y = (sidx & 1) ? yb : ya;
        const int loc = sidx >> 1;
        const int n = loc / (H / R), oh0 = (loc % (H / R)) * R;
        float* img = y + (size_t)n * F * HW + (size_t)oh0 * W;
        for (int i = 0; i < 16; ++i) {
            const int f = wid + 4 * i;
            f4* p = reinterpret_cast<f4*>(img + (size_t)f * HW);
#pragma unroll
            for (int j = 0; j < R * W / 256; ++j) p[j * 64 + lane] = v;
        }
    }
}

// streaming copy / scale-add / sum: do a read stream and a write stream in different zones help each other?
__global__ __launch_bounds__(256) void kcopy(const f4* __restrict__ a, f4* __restrict__ c, size_t n4) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) if (base + it * 256 < n4) v[it] = a[base + it * 256];
#pragma unroll
    for (int it = 0; it < 4; ++it) if (base + it * 256 < n4) c[base + it * 256] = v[it] * 1.5f;
}
__global__ __launch_bounds__(256) void kaxpy(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t n4) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f4 v[4], w[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) if (base + it * 256 < n4) { v[it] = a[base + it * 256]; w[it] = b[base + it * 256]; }
#pragma unroll
    for (int it = 0; it < 4; ++it) if (base + it * 256 < n4) c[base + it * 256] = v[it] * 1.5f + w[it];
}

static hipEvent_t e0, e1;
template <class Fn> static float best_ms(Fn fn, int reps = 6) {
    float best = 1e9f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(e0));
        fn();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}
template <class Fn> static float avg_ms(Fn fn, int reps = 10) {  // back to back, like a step would issue them
    fn(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < reps; ++rep) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
static void report(const char* name, float best, float avg) {
    printf("%-64s best %.3f ms %5.2f TB/s | back-to-back %.3f ms %5.2f TB/s\n", name, best, TOTAL * 4 / best / 1e9, avg,
           TOTAL * 4 / avg / 1e9);
    fflush(stdout);
}
#define RUN2(name, launch) do { auto fn = [&]() { launch; }; report(name, best_ms(fn), avg_ms(fn)); } while (0)

int main(int argc, char** argv) {
    float* y;
    CK(hipMalloc(&y, TOTAL * 4));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n4 = TOTAL / 4;
    char name[128];
    const int nstrips = N * (H / R);
    if (argc > 1 && argv[1][0] == 'c') {   // copy / scale-add with the streams in one zone or in two
        char* big;
        CK(hipMalloc(&big, (size_t)100 << 30));
        const unsigned grid = (unsigned)((n4 + 1023) / 1024);
        for (size_t dm : {(size_t)2048, (size_t)8192, (size_t)32768, (size_t)66560, (size_t)73728, (size_t)90000}) {
            const f4* a = (const f4*)big;
            const f4* b2 = (const f4*)(big + ((size_t)3 << 30));
            f4* c = (f4*)(big + (dm << 20));
            auto fc = [&]() { kcopy<<<grid, 256>>>(a, c, n4); };
            auto fa = [&]() { kaxpy<<<grid, 256>>>(a, b2, c, n4); };
            auto fa2 = [&]() { kaxpy<<<grid, 256>>>(a, (const f4*)(big + ((dm + 3072) << 20)), c, n4); };
            const float tc = avg_ms(fc, 5), ta = avg_ms(fa, 5), ta2 = avg_ms(fa2, 5);
            printf("dst %5zu MiB from src: copy %.3f ms %.2f TB/s | scale-add (both sources near) %.3f ms %.2f TB/s | (second source near dst) %.3f ms %.2f TB/s\n",
                   dm, tc, 2 * TOTAL * 4 / tc / 1e9, ta, 3 * TOTAL * 4 / ta / 1e9, ta2, 3 * TOTAL * 4 / ta2 / 1e9);
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'd') {   // two half tensors a distance D apart inside one big allocation
        char* big;
        CK(hipMalloc(&big, (size_t)100 << 30));
        for (size_t base : {(size_t)0, (size_t)7 << 30, (size_t)20 << 30})
            for (size_t dm : {(size_t)1024, (size_t)2048, (size_t)4096, (size_t)8192, (size_t)16384, (size_t)24576, (size_t)32768, (size_t)49152, (size_t)65536}) {
                float* ya = (float*)(big + base);
                float* yb = (float*)(big + base + (dm << 20));
                auto f = [&]() { ksplit<<<1280, 256>>>(ya, yb, nstrips); };
                auto f1 = [&]() { kaux<1, 0><<<1280, 256>>>(ya, nstrips); };
                printf("base %2zu GiB, halves %5zu MiB apart: split K1 %.3f ms   (contiguous K1 at base: %.3f)\n", base >> 30, dm, avg_ms(f, 5), avg_ms(f1, 5));
                fflush(stdout);
            }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {   // map of one big allocation: the K1 walk over windows of 40 images (514 MB)
        const size_t gib = argc > 2 ? (size_t)atol(argv[2]) : 64;
        const int sep = argc > 3 ? atoi(argv[3]) : 0;  // 1: separate allocations of 1 GiB instead of one big one
        char* big = nullptr;
        if (!sep) { CK(hipMalloc(&big, gib << 30)); printf("one allocation of %zu GiB at %p\n", gib, (void*)big); }
        const int win_strips = 40 * (H / R);
        for (size_t k = 0; k + 1 <= gib * 2; ++k) {
            char* b = big + (k << 29);
            if (sep) { CK(hipMalloc(&b, (size_t)520 << 20)); }
            float* yy = (float*)b;
            auto f1 = [&]() { kaux<1, 0><<<1120, 256>>>(yy, win_strips); };
            auto f3 = [&]() { CK(hipMemsetAsync(yy, 0, (size_t)40 * F * HW * 4, 0)); };
            const float t1 = avg_ms(f1, 4), t3 = avg_ms(f3, 4);
            printf("%s%5.1f GiB %p: K1 %.4f  memset %.4f  ratio %.2f %s\n", sep ? "alloc " : "offset ", k * 0.5, (void*)b, t1, t3, t1 / t3,
                   t1 / t3 > 1.12 ? "SLOW" : "fast");
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 's') {   // plane-stride sweep on a few allocations
        for (int k = 0; k < 6; ++k) {
            char* b;
            CK(hipMalloc(&b, (size_t)2400 << 20));
            float* yy = (float*)b;
            auto f1 = [&]() { kaux<1, 0><<<1280, 256>>>(yy, nstrips); };
            printf("alloc %d %p: K1 plain %.3f |", k, (void*)b, avg_ms(f1, 5));
            for (unsigned pad : {0u, 256u, 512u, 1024u, 2048u, 4096u, 8192u, 16384u, 32768u}) {
                auto f2 = [&]() { kstride<0><<<1280, 256>>>(yy, nstrips, HW * 4 + pad); };
                printf(" +%u: %.3f", pad, avg_ms(f2, 5));
            }
            printf("\n");
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'v') {   // allocation APIs: which of them hand out memory the strided walks like?
        for (int api = 0; api < 4; ++api) {
            for (int k = 0; k < 10; ++k) {
                char* b = nullptr;
                const size_t bytes = TOTAL * 4;
                hipError_t e = hipSuccess;
                if (api == 0) e = hipMalloc(&b, bytes);
                else if (api == 1) e = hipExtMallocWithFlags((void**)&b, bytes, hipDeviceMallocUncached);
                else if (api == 2) e = hipMallocAsync((void**)&b, bytes, 0);
                else {
                    hipMemAllocationProp prop = {};
                    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
                    size_t gran = 0;
                    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
                    if (e == hipSuccess) {
                        const size_t sz = (bytes + gran - 1) / gran * gran;
                        hipMemGenericAllocationHandle_t h;
                        e = hipMemCreate(&h, sz, &prop, 0);
                        if (e == hipSuccess) e = hipMemAddressReserve((void**)&b, sz, 0, nullptr, 0);
                        if (e == hipSuccess) e = hipMemMap(b, sz, 0, h, 0);
                        hipMemAccessDesc acc = {};
                        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                        if (e == hipSuccess) e = hipMemSetAccess(b, sz, &acc, 1);
                        if (k == 0) printf("VMM granularity %zu\n", gran);
                    }
                }
                if (e != hipSuccess) { printf("api %d: %s\n", api, hipGetErrorString(e)); break; }
                float* yy = (float*)b;
                auto f0 = [&]() { kaux<0, 2><<<1280, 256>>>(yy, nstrips); };
                auto f1 = [&]() { kaux<1, 0><<<1280, 256>>>(yy, nstrips); };
                auto f3 = [&]() { CK(hipMemsetAsync(yy, 0, TOTAL * 4, 0)); };
                printf("api %d alloc %2d %p: K0 nt %.3f  K1 plain %.3f  memset %.3f ms\n", api, k, (void*)b, avg_ms(f0, 5), avg_ms(f1, 5), avg_ms(f3, 5));
                fflush(stdout);
            }
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'm') {   // many separate allocations, the same walks: does the placement matter?
        const int count = argc > 2 ? atoi(argv[2]) : 40;
        const size_t extra = argc > 3 ? (size_t)atol(argv[3]) : 0;
        for (int k = 0; k < count; ++k) {
            char* b;
            CK(hipMalloc(&b, TOTAL * 4 + extra + (size_t)k * 4096 * 3));
            float* yy = (float*)b;
            auto f0 = [&]() { kaux<0, 2><<<1280, 256>>>(yy, nstrips); };
            auto f1 = [&]() { kaux<1, 0><<<1280, 256>>>(yy, nstrips); };
            auto f2 = [&]() { kaux<0, 0><<<1280, 256>>>(yy, nstrips); };
            auto f3 = [&]() { CK(hipMemsetAsync(yy, 0, TOTAL * 4, 0)); };
            printf("alloc %2d %p: K0 nt %.3f  K0 plain %.3f  K1 plain %.3f  memset %.3f ms\n", k, (void*)b, avg_ms(f0, 5), avg_ms(f2, 5),
                   avg_ms(f1, 5), avg_ms(f3, 5));
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'o') {   // base-address sweep: the same walk at base + k * step
        char* big;
        const size_t gib = argc > 4 ? (size_t)atol(argv[4]) : 4;
        CK(hipMalloc(&big, gib << 30));
        if (argc <= 4) big += (size_t)1 << 30;
        printf("big+1GiB = %p\n", (void*)big);
        const size_t step = argc > 2 ? (size_t)atol(argv[2]) : 4096;
        const int count = argc > 3 ? atoi(argv[3]) : 64;
        for (int k = 0; k < count; ++k) {
            float* yy = (float*)(big + k * step);
            auto f0 = [&]() { kaux<0, 2><<<1280, 256>>>(yy, nstrips); };
            auto f1 = [&]() { kaux<1, 0><<<1280, 256>>>(yy, nstrips); };
            auto f2 = [&]() { kaux<0, 0><<<1280, 256>>>(yy, nstrips); };
            printf("offset %8zu: K0 nt %.3f  K0 plain %.3f  K1 plain %.3f ms\n", k * step, avg_ms(f0, 5), avg_ms(f2, 5), avg_ms(f1, 5));
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'a') {   // does the placement of the buffer matter? several allocations, same walks
        float* bufs[4];
        char* big;
        CK(hipMalloc(&big, (size_t)6 << 30));
        bufs[0] = y;
        bufs[1] = (float*)(big + ((size_t)1 << 30));
        bufs[2] = (float*)(big + ((size_t)3 << 30) + 4096 * 17);
        CK(hipMalloc(&bufs[3], TOTAL * 4));
        for (int b = 0; b < 4; ++b) {
            float* yy = bufs[b];
            snprintf(name, sizeof name, "buffer %d (%p): memset", b, (void*)yy);
            RUN2(name, CK(hipMemsetAsync(yy, 0, TOTAL * 4, 0)));
            snprintf(name, sizeof name, "buffer %d: K0 plain grid 1280", b);
            RUN2(name, (kaux<0, 0><<<1280, 256>>>(yy, nstrips)));
            snprintf(name, sizeof name, "buffer %d: K0 nt grid 1280", b);
            RUN2(name, (kaux<0, 2><<<1280, 256>>>(yy, nstrips)));
            snprintf(name, sizeof name, "buffer %d: K1 plain grid 1280", b);
            RUN2(name, (kaux<1, 0><<<1280, 256>>>(yy, nstrips)));
            snprintf(name, sizeof name, "buffer %d: one 16-byte store per thread", b);
            RUN2(name, (kgrid<false, 1><<<(unsigned)(n4 / 256), 256>>>((f4*)yy, n4)));
            snprintf(name, sizeof name, "buffer %d: grid-stride 1024 blocks", b);
            RUN2(name, (kgrid<false, 1><<<1024, 256>>>((f4*)yy, n4)));
        }
        return 0;
    }
    if (argc > 1) {   // occupancy: the same walks with dynamic LDS limiting the workgroups per CU (160 KB / lds)
        for (int ldsKB : {28}) {
            const int per_cu = ldsKB ? 160 / ldsKB : 8;
            for (int grid : {nstrips, 256 * per_cu}) {
                snprintf(name, sizeof name, "K0 strip walk, %d WG/CU (lds %d KB), grid %5d, nt", per_cu, ldsKB, grid);
                RUN2(name, (kstrip<0, true><<<grid, 256, ldsKB * 1024>>>(y, nstrips)));
                snprintf(name, sizeof name, "K0 strip walk, %d WG/CU (lds %d KB), grid %5d, plain", per_cu, ldsKB, grid);
                RUN2(name, (kstrip<0, false><<<grid, 256, ldsKB * 1024>>>(y, nstrips)));
                snprintf(name, sizeof name, "K1 strip walk, %d WG/CU (lds %d KB), grid %5d, nt", per_cu, ldsKB, grid);
                RUN2(name, (kstrip<1, true><<<grid, 256, ldsKB * 1024>>>(y, nstrips)));
            }
        }
#define KAUX(P, A) snprintf(name, sizeof name, "K%d buffer stores, aux %2d, grid 1280", P, A); \
        RUN2(name, (kaux<P, A><<<1280, 256>>>(y, nstrips)));
        KAUX(0, 0) KAUX(0, 1) KAUX(0, 2) KAUX(0, 3) KAUX(0, 16) KAUX(0, 17) KAUX(0, 18) KAUX(0, 19)
        KAUX(1, 0) KAUX(1, 1) KAUX(1, 2) KAUX(1, 3) KAUX(1, 16) KAUX(1, 17) KAUX(1, 18) KAUX(1, 19)
#define KROWS(RU, A, G) snprintf(name, sizeof name, "K6 row sweep, %d rows per unit, aux %2d, grid %4d x 448", RU, A, G); \
        RUN2(name, (krows<RU, A><<<G, 448>>>(y, N * H / RU)));
        for (int G : {256, 512, 768, 1024}) {
            KROWS(1, 0, G) KROWS(1, 2, G) KROWS(2, 0, G) KROWS(2, 2, G) KROWS(4, 0, G) KROWS(4, 2, G)
        }
        RUN2("M  hipMemsetAsync(0)", CK(hipMemsetAsync(y, 0, TOTAL * 4, 0)));
        return 0;
    }
    RUN2("M  hipMemsetAsync(0)", CK(hipMemsetAsync(y, 0, TOTAL * 4, 0)));
    RUN2("M  hipMemsetD32Async(1.5f)", CK(hipMemsetD32Async((hipDeviceptr_t)y, 0x3fc00000, TOTAL, 0)));
    for (int blocks : {256, 512, 1024, 2048, 4096, 8192, 16384, 65536}) {
        snprintf(name, sizeof name, "G  grid-stride plain, %6d blocks, unroll 1", blocks);
        RUN2(name, (kgrid<false, 1><<<blocks, 256>>>((f4*)y, n4)));
        snprintf(name, sizeof name, "G  grid-stride nt   , %6d blocks, unroll 1", blocks);
        RUN2(name, (kgrid<true, 1><<<blocks, 256>>>((f4*)y, n4)));
        snprintf(name, sizeof name, "G  grid-stride plain, %6d blocks, unroll 4", blocks);
        RUN2(name, (kgrid<false, 4><<<blocks, 256>>>((f4*)y, n4)));
        snprintf(name, sizeof name, "G  grid-stride nt   , %6d blocks, unroll 4", blocks);
        RUN2(name, (kgrid<true, 4><<<blocks, 256>>>((f4*)y, n4)));
    }
    {   // one store per thread
        const unsigned blocks = (unsigned)(n4 / 256);
        RUN2("G  one 16-byte store per thread, plain", (kgrid<false, 1><<<blocks, 256>>>((f4*)y, n4)));
        RUN2("G  one 16-byte store per thread, nt", (kgrid<true, 1><<<blocks, 256>>>((f4*)y, n4)));
    }
    for (unsigned spanKB : {4u, 16u, 64u, 448u, 1792u}) {
        const unsigned span4 = spanKB * 64;
        const unsigned full = (unsigned)((n4 + span4 - 1) / span4);
        for (unsigned grid : {full, 2048u, 1024u, 512u}) {
            if (grid > full) continue;
            snprintf(name, sizeof name, "S  span %4u KB per WG, grid %6u, plain", spanKB, grid);
            RUN2(name, (kspan<false><<<grid, 256>>>((f4*)y, n4, span4)));
            snprintf(name, sizeof name, "S  span %4u KB per WG, grid %6u, nt", spanKB, grid);
            RUN2(name, (kspan<true><<<grid, 256>>>((f4*)y, n4, span4)));
        }
    }
#define KSTRIP(P) \
    for (int grid : {nstrips, 2048, 1024, 512}) { \
        snprintf(name, sizeof name, "K%d strip walk, grid %5d, plain", P, grid); \
        RUN2(name, (kstrip<P, false><<<grid, 256>>>(y, nstrips))); \
        snprintf(name, sizeof name, "K%d strip walk, grid %5d, nt", P, grid); \
        RUN2(name, (kstrip<P, true><<<grid, 256>>>(y, nstrips))); \
    }
    KSTRIP(0) KSTRIP(1) KSTRIP(2) KSTRIP(3) KSTRIP(4) KSTRIP(5)
    return 0;
}
