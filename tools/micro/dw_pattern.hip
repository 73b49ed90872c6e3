// Access-pattern probe for the small-plane depthwise backward (depthwise_march.hip, 14 x 14 planes): three 103 MB read
// streams + one write stream, (a) as the marching kernel issues them -- a lane owns 2 columns of a plane, 7 lanes per row, 9
// planes per wave, 14 rows one after the other, PF rows requested ahead -- against (b) a plain grid-stride float4 sweep of
// the same bytes and (c) the marching pattern with 9 ROWS of one plane across the lane groups (contiguous 504 bytes per
// instruction). No arithmetic beyond one add per element. build: hipcc --offload-arch=gfx950 -O3 dw_pattern.hip -o dw_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PF, int LDSF = 1>
__global__ __launch_bounds__(256) void march_planes(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                    float* __restrict__ d, unsigned planes) {
    __shared__ float occupancy_pad[LDSF];  // LDSF floats of LDS per workgroup: limits the workgroups a CU holds
    if (planes == 0xffffffffu) occupancy_pad[threadIdx.x % LDSF] = 1.f;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int grp = lane / 7, cg = lane - grp * 7;
    const unsigned p = wave * 9u + grp;
    const bool on = grp < 9 && p < planes;
    unsigned off = (p * 196u + cg * 2u) * 4u;
    float2 ra[PF], rb[PF], rc[PF];
    auto ld = [&](const float* base, unsigned o, bool ok) { float2 v = make_float2(0.f, 0.f); if (ok) v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + o); return v; };
#pragma unroll
    for (int u = 0; u < PF; ++u) { ra[u] = ld(a, off + u * 56u, on); rb[u] = ld(b, off + u * 56u, on); rc[u] = ld(c, off + u * 56u, on); }
    for (int k0 = 0; k0 < 14; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k = k0 + u;
            const float2 x = ra[u], y = rb[u], z = rc[u];
            const bool more = on && k + PF < 14;
            ra[u] = ld(a, off + PF * 56u, more); rb[u] = ld(b, off + PF * 56u, more); rc[u] = ld(c, off + PF * 56u, more);
            if (on && k < 14) *reinterpret_cast<float2*>(reinterpret_cast<char*>(d) + off) = make_float2(x.x + y.x + z.x, x.y + y.y + z.y);
            off += 56u;
        }
    }
}

// (a) again with the product's instructions (round 5): raw buffer loads / stores, masked by an out-of-range offset instead of a
// branch, straight-line
typedef int rsrc_i4 __attribute__((ext_vector_type(4)));
typedef float buf_f32x2 __attribute__((ext_vector_type(2)));
__device__ buf_f32x2 buffer_load_f32x2(rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ void buffer_store_f32x2(buf_f32x2 v, rsrc_i4 rs, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ __forceinline__ rsrc_i4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    rsrc_i4 r;
    r[0] = (int)(unsigned)u; r[1] = (int)(unsigned)((u >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}
template <int PF, bool WRITE, bool LOADS_FIRST = false, int LDSF = 1>
__global__ __launch_bounds__(256) void march_planes_buf(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                        float* __restrict__ d, unsigned planes, unsigned bytes) {
    __shared__ float occupancy_pad[LDSF];
    if (planes == 0xffffffffu) occupancy_pad[threadIdx.x % LDSF] = 1.f;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int grp = lane / 7, cg = lane - grp * 7;
    const unsigned p = wave * 9u + grp;
    const bool on = grp < 9 && p < planes;
    const rsrc_i4 ra_ = make_rsrc(a, bytes), rb_ = make_rsrc(b, bytes), rc_ = make_rsrc(c, bytes), rd_ = make_rsrc(d, bytes);
    unsigned off = (p * 196u + cg * 2u) * 4u;
    buf_f32x2 ra[PF], rb[PF], rc[PF];
    auto ld = [&](rsrc_i4 rs, unsigned o, bool ok) { return buffer_load_f32x2(rs, (int)(ok ? o : 0x80000000u), 0, 0); };
#pragma unroll
    for (int u = 0; u < PF; ++u) { ra[u] = ld(ra_, off + u * 56u, on); rb[u] = ld(rb_, off + u * 56u, on); rc[u] = ld(rc_, off + u * 56u, on); }
    float acc = 0.f;
    for (int k0 = 0; k0 < 14; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k = k0 + u;
            const buf_f32x2 x = ra[u], y = rb[u], z = rc[u];
            const buf_f32x2 o = {x[0] + y[0] + z[0], x[1] + y[1] + z[1]};
            const bool more = on && k + PF < 14;
            if (LOADS_FIRST) { ra[u] = ld(ra_, off + PF * 56u, more); rb[u] = ld(rb_, off + PF * 56u, more); rc[u] = ld(rc_, off + PF * 56u, more); }
            if (WRITE) buffer_store_f32x2(o, rd_, (int)((on && k < 14) ? off : 0x80000000u), 0, 0);
            else acc += o[0] + o[1];
            if (!LOADS_FIRST) { ra[u] = ld(ra_, off + PF * 56u, more); rb[u] = ld(rb_, off + PF * 56u, more); rc[u] = ld(rc_, off + PF * 56u, more); }
            off += 56u;
        }
    }
    if (!WRITE && acc == 123.456f) d[0] = acc;
}

// (a) with flat-global instructions in straight-line form: every lane loads (lanes that sit out read offset 0 and drop the
// value by a select), the store stays under its branch
template <int PF, int LDSF = 1>
__global__ __launch_bounds__(256) void march_planes_clamp(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                          float* __restrict__ d, unsigned planes) {
    __shared__ float occupancy_pad[LDSF];
    if (planes == 0xffffffffu) occupancy_pad[threadIdx.x % LDSF] = 1.f;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int grp = lane / 7, cg = lane - grp * 7;
    const unsigned p = wave * 9u + grp;
    const bool on = grp < 9 && p < planes;
    unsigned off = (p * 196u + cg * 2u) * 4u;
    float2 ra[PF], rb[PF], rc[PF];
    auto ld = [&](const float* base, unsigned o, bool ok) {
        const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + (ok ? o : 0u));
        return make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) { ra[u] = ld(a, off + u * 56u, on); rb[u] = ld(b, off + u * 56u, on); rc[u] = ld(c, off + u * 56u, on); }
    for (int k0 = 0; k0 < 14; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k = k0 + u;
            const float2 x = ra[u], y = rb[u], z = rc[u];
            const bool more = on && k + PF < 14;
            ra[u] = ld(a, off + PF * 56u, more); rb[u] = ld(b, off + PF * 56u, more); rc[u] = ld(c, off + PF * 56u, more);
            if (on && k < 14) *reinterpret_cast<float2*>(reinterpret_cast<char*>(d) + off) = make_float2(x.x + y.x + z.x, x.y + y.y + z.y);
            off += 56u;
        }
    }
}

// 9 consecutive ROWS of the tensor (viewed as rows of 14 floats) across the lane groups: 504 contiguous bytes per instruction;
// a wave walks `steps` such 9-row slabs that lie 9 rows apart (so the whole tensor is covered by consecutive waves)
template <int PF>
__global__ __launch_bounds__(256) void march_rows(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                  float* __restrict__ d, unsigned rows, int steps) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned r0 = wave * (unsigned)(steps * 9);
    const bool lane_on = lane < 63;
    unsigned off = (r0 * 14u) * 4u + lane * 8u;
    float2 ra[PF], rb[PF], rc[PF];
    auto ld = [&](const float* base, unsigned o, bool ok) { float2 v = make_float2(0.f, 0.f); if (ok) v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + o); return v; };
    auto ok = [&](int k) { return lane_on && k < steps && r0 + (unsigned)k * 9u + (unsigned)(lane / 7) < rows; };
#pragma unroll
    for (int u = 0; u < PF; ++u) { ra[u] = ld(a, off + u * 504u, ok(u)); rb[u] = ld(b, off + u * 504u, ok(u)); rc[u] = ld(c, off + u * 504u, ok(u)); }
    for (int k0 = 0; k0 < steps; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k = k0 + u;
            const float2 x = ra[u], y = rb[u], z = rc[u];
            ra[u] = ld(a, off + PF * 504u, ok(k + PF)); rb[u] = ld(b, off + PF * 504u, ok(k + PF)); rc[u] = ld(c, off + PF * 504u, ok(k + PF));
            if (ok(k)) *reinterpret_cast<float2*>(reinterpret_cast<char*>(d) + off) = make_float2(x.x + y.x + z.x, x.y + y.y + z.y);
            off += 504u;
        }
    }
}

__global__ __launch_bounds__(256) void sweep4(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                              float4* __restrict__ d, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i], z = c[i];
        d[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
    }
}

int main() {
    const unsigned planes = 512u * 256u;
    const size_t n = (size_t)planes * 196;
    float *a, *b, *c, *d;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f us  %5.2f TB/s\n", name, ms * 100, 4.0 * n * 4 / (ms / 10 * 1e-3) / 1e12);
    };
    // the same launches COLD: a 512 MB sweep of something else before each one (events around the measured launch only). The
    // loop above re-reads the same 411 MB next to a 256 MB Infinity Cache; inside a training step nothing is re-read.
    float4* evict; CK(hipMalloc(&evict, 512ull << 20)); CK(hipMemset(evict, 0, 512ull << 20));
    auto time_cold = [&](const char* name, auto launch) {
        float tot = 0.f;
        for (int i = 0; i < 8; ++i) {
            sweep4<<<2048, 256>>>(evict, evict + (128u << 20) / 16, evict + (256u << 20) / 16, evict + (384u << 20) / 16, (128u << 20) / 16);
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (i >= 2) tot += ms;
        }
        printf("%-44s %7.1f us  (cold)\n", name, tot / 6 * 1000);
    };
    const unsigned waves = (planes + 8) / 9, blocks = (waves + 3) / 4;
    time("march 9 planes x 56 B per instruction, PF 1", [&] { march_planes<1><<<blocks, 256>>>(a, b, c, d, planes); });
    time("march 9 planes x 56 B per instruction, PF 2", [&] { march_planes<2><<<blocks, 256>>>(a, b, c, d, planes); });
    time("march 9 planes x 56 B per instruction, PF 4", [&] { march_planes<4><<<blocks, 256>>>(a, b, c, d, planes); });
    time("march 9 planes x 56 B per instruction, PF 7", [&] { march_planes<7><<<blocks, 256>>>(a, b, c, d, planes); });
    time("  PF 4, 6 workgroups (waves / SIMD) per CU", [&] { march_planes<4, 6600><<<blocks, 256>>>(a, b, c, d, planes); });
    time("  PF 4, 4 per CU", [&] { march_planes<4, 10000><<<blocks, 256>>>(a, b, c, d, planes); });
    time("  PF 4, 3 per CU", [&] { march_planes<4, 13500><<<blocks, 256>>>(a, b, c, d, planes); });
    time("  PF 4, 2 per CU", [&] { march_planes<4, 20000><<<blocks, 256>>>(a, b, c, d, planes); });
    time("  PF 7, 3 per CU", [&] { march_planes<7, 13500><<<blocks, 256>>>(a, b, c, d, planes); });
    time("  PF 2, 3 per CU", [&] { march_planes<2, 13500><<<blocks, 256>>>(a, b, c, d, planes); });
    time("  buffer instructions, straight-line, PF 4", [&] { march_planes_buf<4, true><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time("  buffer instructions, PF 2", [&] { march_planes_buf<2, true><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time("  buffer instructions, PF 4, reads only (3/4 of the bytes)", [&] { march_planes_buf<4, false><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("march 9 planes, global loads, PF 4", [&] { march_planes<4><<<blocks, 256>>>(a, b, c, d, planes); });
    time_cold("  PF 4, 3 per CU", [&] { march_planes<4, 13500><<<blocks, 256>>>(a, b, c, d, planes); });
    time_cold("  global instructions, straight-line loads, PF 4", [&] { march_planes_clamp<4><<<blocks, 256>>>(a, b, c, d, planes); });
    time_cold("  global instructions, straight-line loads, PF 7", [&] { march_planes_clamp<7><<<blocks, 256>>>(a, b, c, d, planes); });
    time_cold("  buffer instructions, straight-line, PF 4", [&] { march_planes_buf<4, true><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer instructions, loads before the store, PF 4", [&] { march_planes_buf<4, true, true><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer instructions, loads before the store, PF 7", [&] { march_planes_buf<7, true, true><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer instructions, store first, PF 7", [&] { march_planes_buf<7, true, false><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer, PF 4, 6 per CU", [&] { march_planes_buf<4, true, false, 6600><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer, PF 4, 4 per CU", [&] { march_planes_buf<4, true, false, 10000><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer, PF 4, 3 per CU", [&] { march_planes_buf<4, true, false, 13500><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer, PF 4, 2 per CU", [&] { march_planes_buf<4, true, false, 20000><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer, PF 7, 3 per CU", [&] { march_planes_buf<7, true, false, 13500><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  buffer, PF 7, 2 per CU", [&] { march_planes_buf<7, true, false, 20000><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("  global straight-line, PF 4, 3 per CU", [&] { march_planes_clamp<4, 13500><<<blocks, 256>>>(a, b, c, d, planes); });
    time_cold("  global conditional, PF 4, 2 per CU", [&] { march_planes<4, 20000><<<blocks, 256>>>(a, b, c, d, planes); });
    time_cold("  global conditional, PF 4, 6 per CU", [&] { march_planes<4, 6600><<<blocks, 256>>>(a, b, c, d, planes); });
    time_cold("  buffer instructions, PF 4, reads only", [&] { march_planes_buf<4, false><<<blocks, 256>>>(a, b, c, d, planes, (unsigned)(n * 4)); });
    time_cold("float4 sweep, one float4 per thread", [&] { sweep4<<<(unsigned)((n / 4 + 255) / 256), 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)d, n / 4); });
    const unsigned rows = planes * 14u;
    for (int steps : {14, 28}) {
        const unsigned w2 = (rows + steps * 9 - 1) / (steps * 9), b2 = (w2 + 3) / 4;
        char nm[64];
        snprintf(nm, sizeof nm, "march 9 rows (504 B) per instruction, %d steps, PF 2", steps);
        time(nm, [&] { march_rows<2><<<b2, 256>>>(a, b, c, d, rows, steps); });
        snprintf(nm, sizeof nm, "march 9 rows (504 B) per instruction, %d steps, PF 4", steps);
        time(nm, [&] { march_rows<4><<<b2, 256>>>(a, b, c, d, rows, steps); });
    }
    time("grid-stride float4 sweep, 2048 blocks", [&] { sweep4<<<2048, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)d, n / 4); });
    time("float4 sweep, one float4 per thread", [&] { sweep4<<<(unsigned)((n / 4 + 255) / 256), 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)d, n / 4); });
    return 0;
}
