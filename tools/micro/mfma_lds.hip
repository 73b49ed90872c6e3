// mfma_lds.hip -- sustained rate of the igemm inner loop shape: per k-step 2x ds_read2_b32 feeding 4 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: MFMA only, 1: LDS-fed MFMAs (the real loop), 2: LDS-fed, prefetched one k-step ahead
__global__ __launch_bounds__(256) void kern(float* out, int tiles) {
    __shared__ float As[16][128];
    __shared__ float Bs[16][128];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    for (int i = tid; i < 16 * 128; i += 256) { unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; (&As[0][0])[i] = RND ? ((h & 0xffffff) / 8388608.0f - 1.0f) : 1.0f + i * 1e-6f; h *= 0x27d4eb2fu; h ^= h >> 15; (&Bs[0][0])[i] = RND ? ((h & 0xffffff) / 8388608.0f - 1.0f) : 0.5f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a0 = tid, a1 = tid + 1, b0 = 2.f, b1 = 3.f;
    for (int t = 0; t < tiles; ++t) {
        if (MODE == 0) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const float x0 = As[2 * ks + lhi][(wm * 2 + 0) * 32 + l31], x1 = As[2 * ks + lhi][(wm * 2 + 1) * 32 + l31];
                const float y0 = Bs[2 * ks + lhi][(wn * 2 + 0) * 32 + l31], y1 = Bs[2 * ks + lhi][(wn * 2 + 1) * 32 + l31];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[1][1], 0, 0, 0);
            }
            if (MODE == 2) __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}
template <class F>
static void run(const char* name, F launch, int blocks, int tiles, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(blocks, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch(blocks, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 4.0 * 32 * 4096.0 * blocks * (double)tiles * reps;
    printf("%-34s blocks %5d tiles %5d reps %3d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, tiles, reps, ms, flop / ms / 1e9);
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 16384 * sizeof(float));
    for (int pass = 0; pass < 2; ++pass) {
        for (int occ = 1; occ <= 3; ++occ) {
            char nm[64];
            snprintf(nm, 64, "mfma only occ%d", occ);
            run(nm, [&](int b, int t) { kern<0><<<b, 256>>>(out, t); }, 256 * occ, 400, 1);
            snprintf(nm, 64, "lds-fed occ%d", occ);
            run(nm, [&](int b, int t) { kern<1><<<b, 256>>>(out, t); }, 256 * occ, 400, 1);
            snprintf(nm, 64, "lds-fed + barrier/tile occ%d", occ);
            run(nm, [&](int b, int t) { kern<2><<<b, 256>>>(out, t); }, 256 * occ, 400, 1);
        }
        // the conv shape: 1568 blocks x 36 tiles, and a sustained 100-launch train
        run("lds-fed 1568 blk x 36 tiles", [&](int b, int t) { kern<1><<<b, 256>>>(out, t); }, 1568, 36, 1);
        run("lds-fed 1536 blk x 36 tiles", [&](int b, int t) { kern<1><<<b, 256>>>(out, t); }, 1536, 36, 1);
        run("lds-fed 1568 blk x 36 tiles x100", [&](int b, int t) { kern<1><<<b, 256>>>(out, t); }, 1568, 36, 100);
        run("mfma only 768 blk x 400 x20", [&](int b, int t) { kern<0><<<b, 256>>>(out, t); }, 768, 400, 20);
    }
    return 0;
}
