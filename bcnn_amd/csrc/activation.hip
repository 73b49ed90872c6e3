// activation.hip -- the activation map, stand-alone entry points (HBM-bound streaming kernels).
// Reference semantics: bcnn_forward_activation_cpu / bcnn_backward_activation_cpu,
// src/layers/bcnn_activation_layer.c:90-146, 165-226. Inside conv / depthwise / batch-norm nodes the
// same act_fwd / act_bwd_factor functions run fused in the producing kernel's epilogue instead.
#include "chan_reduce.h"

namespace bcnn_hip {

struct ActFwdBody {
    float* x;
    const float* slopes;
    int act;
    bool al;
    __device__ void operator()(unsigned off, int c, int cnt) const {
        const float sl = (act == BCNN_HIP_ACT_PRELU) ? slopes[c] : 0.f;
        if (cnt == 4 && al && (off & 3u) == 0) {
            float4 v = *reinterpret_cast<float4*>(x + off);
            v.x = act_fwd(v.x, act, sl); v.y = act_fwd(v.y, act, sl);
            v.z = act_fwd(v.z, act, sl); v.w = act_fwd(v.w, act, sl);
            *reinterpret_cast<float4*>(x + off) = v;
        } else {
            for (int k = 0; k < cnt; ++k) x[off + k] = act_fwd(x[off + k], act, sl);
        }
    }
};

struct ActBwdBody {
    const float* x;
    float* dx;
    const float* slopes;
    int act;
    bool al;
    __device__ void operator()(unsigned off, int c, int cnt) const {
        const float sl = (act == BCNN_HIP_ACT_PRELU) ? slopes[c] : 0.f;
        if (cnt == 4 && al && (off & 3u) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(x + off);
            float4 g = *reinterpret_cast<float4*>(dx + off);
            g.x *= act_bwd_factor(v.x, act, sl); g.y *= act_bwd_factor(v.y, act, sl);
            g.z *= act_bwd_factor(v.z, act, sl); g.w *= act_bwd_factor(v.w, act, sl);
            *reinterpret_cast<float4*>(dx + off) = g;
        } else {
            for (int k = 0; k < cnt; ++k) dx[off + k] *= act_bwd_factor(x[off + k], act, sl);
        }
    }
};

// PReLU slope gradient: dslope[c] += sum dx*x*(x<0), taken BEFORE dx is rescaled (:213-216)
struct PreluGradF {
    static constexpr int kInFlight = 4;  // chan_reduce_partial's unroll
    const float* x;
    const float* dx;
    __device__ void operator()(long long off, int, float (&acc)[1]) const {
        const float v = x[off];
        acc[0] += dx[off] * v * (float)(v < 0);
    }
    __device__ void vec4(long long off, int c, float (&acc)[1]) const {
        for (int k = 0; k < 4; ++k) (*this)(off + k, c, acc);
    }
};

__global__ void accumulate_kernel(const float* __restrict__ partials, int C, int splits, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < splits; ++i) s += (double)partials[(long long)c * splits + i];
    out[c] += (float)s;
}

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_activation_forward(float* x, size_t size, int act, const float* slopes, int spatial,
                                 int channels) {
    if (!size || act == BCNN_HIP_ACT_NONE) return;
    const bool al = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (act != BCNN_HIP_ACT_PRELU) {  // no channel dependence: treat as one long plane
        launch_chan_map(ActFwdBody{x, nullptr, act, al}, 1, 1, (int)size);
        return;
    }
    const int n = (int)(size / ((size_t)spatial * channels));
    launch_chan_map(ActFwdBody{x, slopes, act, al}, n, channels, spatial);
}

void bcnn_hip_activation_backward(const float* x, float* dx, size_t size, int act, const float* slopes,
                                  float* dslopes, int spatial, int channels) {
    if (!size || act == BCNN_HIP_ACT_NONE) return;
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    if (act != BCNN_HIP_ACT_PRELU) {
        launch_chan_map(ActBwdBody{x, dx, nullptr, act, al}, 1, 1, (int)size);
        return;
    }
    const int n = (int)(size / ((size_t)spatial * channels));
    if (dslopes) {
        const long long M = (long long)n * spatial;
        const int splits = chan_splits(channels, M);
        float* part = reduce_scratch((size_t)channels * splits);
        launch_chan_reduce<1>(PreluGradF{x, dx}, channels, spatial, M, splits, part);
        accumulate_kernel<<<ceil_div(channels, 256), 256, 0, current_stream()>>>(part, channels, splits, dslopes);
        KERNEL_CHECK();
    }
    launch_chan_map(ActBwdBody{x, dx, slopes, act, al}, n, channels, spatial);
}

}  // extern "C"
