// mfma_dma.hip -- the 64x64-tile igemm loop in isolation: per K-tile a wave issues 8 LDS-DMA loads (L2-resident
// source), 16 ds_read_b32 and 8 dependent MFMAs, then drains vmcnt and hits a barrier. Reports TFLOP/s and the
// shader clock (s_memtime cycles / wall_clock64 ticks at 100 MHz) to separate issue limits from clock limits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../bcnn_amd/csrc/lds_dma.h"
using namespace bcnn_hip;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: MFMA + LDS reads only; 1: + DMA loads (no wait until tile end); 2: DMA only (no MFMA)
__global__ __launch_bounds__(256) void kern(const float* src, unsigned src_bytes, float* out, int tiles,
                                            unsigned long long* clk) {
    __shared__ float As[2][16][64];
    __shared__ float Bs[2][16][64];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    for (int i = tid; i < 2 * 16 * 64; i += 256) { (&As[0][0][0])[i] = 1.0f + (i & 15) * 0.01f; (&Bs[0][0][0])[i] = 0.5f; }
    __syncthreads();
    const rsrc_i4 rs = make_rsrc(src, src_bytes);
    const unsigned la = lds_offset(&As[0][0][0]), lb = lds_offset(&Bs[0][0][0]);
    unsigned voff = (unsigned)((blockIdx.x * 977 + lane) * 4) % (src_bytes - 4096);
    voff &= ~3u;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned long long t0 = wall_clock64(), c0 = clock64();
    for (int t = 0; t < tiles; ++t) {
        const int cur = t & 1;
        if (MODE >= 1) {
            const unsigned soff = (unsigned)((t * 64 * 1024) % (int)(src_bytes / 2)) & ~3u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dma_row(rs, lb + (unsigned)((((cur ^ 1) * 16 + 4 * wid + r) * 64) * 4), voff, soff + r * 3072);
                dma_row(rs, la + (unsigned)((((cur ^ 1) * 16 + 4 * wid + r) * 64) * 4), voff, soff + r * 3072 + 1536);
            }
        }
        if (MODE != 2) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const float a = As[cur][2 * ks + lhi][wm * 32 + l31], b = Bs[cur][2 * ks + lhi][wn * 32 + l31];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        dma_wait();
        __syncthreads();
    }
    const unsigned long long c1 = clock64(), t1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}
template <class F>
static void run(const char* name, F launch, int blocks, int tiles, unsigned long long* clk_h, unsigned long long* clk_d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(blocks, 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(blocks, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(clk_h, clk_d, 16, hipMemcpyDeviceToHost);
    const double flop = 4.0 * 8 * 4096.0 * blocks * (double)tiles;
    printf("%-30s blocks %5d tiles %5d  %8.3f ms  %7.1f TFLOP/s  shader clock %.2f GHz\n", name, blocks, tiles, ms,
           flop / ms / 1e9, (double)clk_h[0] / ((double)clk_h[1] / 100e6) / 1e9);
}
int main() {
    float *out, *src;
    unsigned long long *clk_d, clk_h[2];
    const unsigned src_bytes = 64u << 20;
    (void)hipMalloc(&out, 256 * 8192 * sizeof(float));
    (void)hipMalloc(&src, src_bytes);
    {   // operand data matters for power: zeros keep the clock at 2.4 GHz, random fp32 does not
        const char* z = getenv("MFMA_DMA_ZERO");
        if (z) (void)hipMemset(src, 0, src_bytes);
        else {
            float* h = (float*)malloc(src_bytes);
            unsigned st = 12345u;
            for (size_t i = 0; i < src_bytes / 4; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
            (void)hipMemcpy(src, h, src_bytes, hipMemcpyHostToDevice);
            free(h);
        }
    }
    (void)hipMalloc(&clk_d, 16);
    for (int pass = 0; pass < 2; ++pass)
        for (int occ = 2; occ <= 8; occ *= 2) {
            char nm[64];
            snprintf(nm, 64, "mfma+lds occ%d", occ);
            run(nm, [&](int b, int t) { kern<0><<<b, 256>>>(src, src_bytes, out, t, clk_d); }, 256 * occ, 2000, clk_h, clk_d);
            snprintf(nm, 64, "mfma+lds+dma occ%d", occ);
            run(nm, [&](int b, int t) { kern<1><<<b, 256>>>(src, src_bytes, out, t, clk_d); }, 256 * occ, 2000, clk_h, clk_d);
            snprintf(nm, 64, "dma only occ%d", occ);
            run(nm, [&](int b, int t) { kern<2><<<b, 256>>>(src, src_bytes, out, t, clk_d); }, 256 * occ, 2000, clk_h, clk_d);
        }
    return 0;
}
