// chan_reduce.h -- deterministic per-channel reductions over dense NCHW tensors (HBM-bound).
//
// Used by batch-norm statistics / backward sums, bias and scale gradients. Two levels:
//   1. chan_reduce_partial: grid (C, splits); each 256-thread workgroup streams its slice of the
//      channel's N*HW elements with 16-byte loads (when HW % 4 == 0), reduces with wave64 shuffles
//      (DPP row moves) and a tiny LDS cross-wave step, and writes NV partial sums;
//   2. the caller's finalize kernel combines the `splits` partials in a fixed order (in double),
// so results do not depend on scheduling -- unlike the reference CUDA kernels' unsynchronised
// `+=` (src/kernels/bcnn_mat.cu:377-379).
#pragma once
#include "common.h"

namespace bcnn_hip {

#ifndef CHAN_WG_PER_CU
#define CHAN_WG_PER_CU 4
#endif
#ifndef CHAN_MIN_ELEMS
#define CHAN_MIN_ELEMS 4096
#endif
inline int chan_splits(int channels, long long per_channel) {
    // aim for >= ~4 workgroups per CU overall (16 measured no faster: tools/exp/bn_trace.sh), but keep >= 4096
    // elements per workgroup
    long long want = ((long long)CHAN_WG_PER_CU * kCUs + channels - 1) / channels;
    long long maxs = (per_channel + CHAN_MIN_ELEMS - 1) / CHAN_MIN_ELEMS;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

#ifdef __HIPCC__
// F: struct with  __device__ void operator()(long long off, int c, float (&acc)[NV]) const  (one element)
// and             __device__ void vec4(long long off, int c, float (&acc)[NV]) const        (4 consecutive)
// U: elements in flight per thread (unroll). The backward-sums functor runs with 2: its kernel then needs <= 48 registers and
// two of its waves per SIMD fit NEXT TO the 2 x 192 registers of a fused Winograd weight-gradient workgroup -- inside a
// backward pass these reductions run while the weight gradients occupy the chip on the side stream (conv.hip).
template <int NV, class F, int U = 4>
__global__ __launch_bounds__(256) void chan_reduce_partial(const F f, int C, int HW, int M /* N*HW */,
                                                           int splits, float* __restrict__ partials) {
    __shared__ float red[4][NV];
    const int c = blockIdx.x, sp = blockIdx.y;
    // slice [lo, hi) of the channel's virtual index space, multiples of 4
    const int per = (((M + splits - 1) / splits) + 3) & ~3;
    const int lo = sp * per;
    int hi = lo + per;
    if (hi > M) hi = M;
    float acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    // (image, offset in plane) of the running index is carried along instead of divided out per step (the per-thread
    // order of the additions is unchanged: bit-identical results; measured neutral, the loop is not VALU-bound)
    if ((HW & 3) == 0) {
        int idx = lo + threadIdx.x * 4;
        if (idx < hi) {
            const int n0 = idx / HW;
            int i = idx - n0 * HW;
            long long base = ((long long)n0 * C + c) * HW;
            const long long img = (long long)C * HW;
#pragma unroll U
            for (; idx < hi; idx += 256 * 4) {
                f.vec4(base + i, c, acc);
                i += 256 * 4;
                while (i >= HW) { i -= HW; base += img; }
            }
        }
    } else {
        int idx = lo + threadIdx.x;
        if (idx < hi) {
            const int n0 = idx / HW;
            int i = idx - n0 * HW;
            long long base = ((long long)n0 * C + c) * HW;
            const long long img = (long long)C * HW;
#pragma unroll U
            for (; idx < hi; idx += 256) {
                f(base + i, c, acc);
                i += 256;
                while (i >= HW) { i -= HW; base += img; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float t = wave_sum(acc[v]);
        if (lane == 0) red[wid][v] = t;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int v = threadIdx.x;
        partials[((long long)c * splits + sp) * NV + v] = (red[0][v] + red[1][v]) + (red[2][v] + red[3][v]);
    }
}

template <int NV, class F>
inline void launch_chan_reduce(const F& f, int C, int HW, long long M, int splits, float* partials) {
    dim3 grid((unsigned)C, (unsigned)splits);
    chan_reduce_partial<NV, F, F::kInFlight><<<grid, 256, 0, current_stream()>>>(f, C, HW, (int)M, splits, partials);
    KERNEL_CHECK();
}
#endif

#ifdef __HIPCC__
// Per-channel elementwise maps over NCHW: one workgroup per (plane, 4096-element chunk) when planes
// are large (channel index is then workgroup-uniform: no per-element division), else a flat
// grid-stride loop with 32-bit index math. Body: __device__ void operator()(unsigned off, int c, int cnt)
// processes cnt (1..4) consecutive elements starting at `off` (cnt == 4 => 16-byte aligned when
// HW % 4 == 0 and the tensor base is 16-byte aligned).
#ifndef PLANE_MAP_ITERS
#define PLANE_MAP_ITERS 4  // float4 groups per thread (a workgroup's chunk is 1024 x that many elements)
#endif
template <class Body>
__global__ __launch_bounds__(256) void plane_map_kernel(const Body body, int C, int HW, int chunks) {
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int c = plane % C;
    const unsigned base = (unsigned)plane * (unsigned)HW;
#pragma unroll
    for (int it = 0; it < PLANE_MAP_ITERS; ++it) {
        const int i = chunk * (1024 * PLANE_MAP_ITERS) + it * 1024 + threadIdx.x * 4;
        if (i < HW) body(base + i, c, (HW - i) >= 4 ? 4 : (HW - i));
    }
}
template <class Body>
__global__ __launch_bounds__(256) void flat_map_div_kernel(const Body body, int C, int HW, unsigned total) {
    const unsigned stride = gridDim.x * blockDim.x * 4u;
    for (unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * 4u; i < total; i += stride) {
        const unsigned plane = i / (unsigned)HW, in = i - plane * (unsigned)HW;
        const int c = (int)(plane % (unsigned)C);
        const int cnt = (total - i) >= 4u ? 4 : (int)(total - i);
        if (in + (unsigned)cnt <= (unsigned)HW) { body(i, c, cnt); continue; }
        // the 4 elements straddle a plane boundary: one at a time
        for (int k = 0; k < cnt; ++k) {
            const unsigned pl = (i + k) / (unsigned)HW;
            body(i + k, (int)(pl % (unsigned)C), 1);
        }
    }
}
// The running element's (offset in plane, channel) are carried along instead of divided out per step: the grid stride is
// step_planes whole planes + step_in elements (host), so one step is two adds, two compares and two conditional
// subtracts -- the two 32-bit divisions per float4 this loop used to do cost more vector-ALU time than the map bodies
// (rocprofv3 SQ counters, profiles/r04_sq_step_resnet18.txt: the batch-norm maps on 28 x 28 planes were 40-60 % VALU-busy).
template <class Body>
__global__ __launch_bounds__(256) void flat_map_kernel(const Body body, int C, int HW, unsigned total, unsigned step_in,
                                                       unsigned step_c) {
    const unsigned stride = gridDim.x * blockDim.x * 4u;
    unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (i >= total) return;
    const unsigned plane0 = i / (unsigned)HW;
    unsigned in = i - plane0 * (unsigned)HW, c = plane0 % (unsigned)C;
    for (; i < total; i += stride) {
        const int cnt = (total - i) >= 4u ? 4 : (int)(total - i);
        if (in + (unsigned)cnt <= (unsigned)HW) {
            body(i, (int)c, cnt);
        } else {  // the 4 elements straddle one or more plane boundaries: one at a time
            unsigned ik = in, ck = c;
            for (int k = 0; k < cnt; ++k) {
                body(i + k, (int)ck, 1);
                if (++ik == (unsigned)HW) { ik = 0; ck = ck + 1 == (unsigned)C ? 0 : ck + 1; }
            }
        }
        in += step_in;
        c += step_c;
        if (in >= (unsigned)HW) { in -= (unsigned)HW; ++c; }
        if (c >= (unsigned)C) c -= (unsigned)C;
    }
}
template <class Body>
inline void launch_chan_map(const Body& body, int N, int C, int HW) {
    const long long total = (long long)N * C * HW;
    if (total == 0) return;
    if (HW >= 1024) {
        const int chunks = (HW + 1024 * PLANE_MAP_ITERS - 1) / (1024 * PLANE_MAP_ITERS);
        plane_map_kernel<Body><<<(unsigned)(N * C * chunks), 256, 0, current_stream()>>>(body, C, HW, chunks);
    } else {
        const int blocks = stream_grid((size_t)(total / 4 + 1), 256);
        if (BCNN_EXP_ENV("BCNN_HIP_FLAT_MAP_DIV")) {  // A/B switch (experiment build): the loop with two divisions per step
            flat_map_div_kernel<Body><<<blocks, 256, 0, current_stream()>>>(body, C, HW, (unsigned)total);
            KERNEL_CHECK();
            return;
        }
        const unsigned stride = (unsigned)blocks * 256u * 4u;  // = step_planes * HW + step_in
        const unsigned step_planes = stride / (unsigned)HW;
        flat_map_kernel<Body><<<blocks, 256, 0, current_stream()>>>(body, C, HW, (unsigned)total,
                                                                    stride - step_planes * (unsigned)HW, step_planes % (unsigned)C);
    }
    KERNEL_CHECK();
}
#endif

// Small per-stream scratch for reduction partials (grow-only, freed at process exit).
float* reduce_scratch(size_t floats);  // blas1.hip

}  // namespace bcnn_hip
