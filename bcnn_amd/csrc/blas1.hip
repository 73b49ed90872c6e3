// blas1.hip -- streaming BLAS-1 and per-channel helpers (HBM-bound; 16-byte accesses, grid-stride).
// Reference semantics: src/kernels/bcnn_mat.c:52-115 (axpy), :319-364 (scal), :366-412 (add_scalar),
// :761-811 (add_bias / scales / grad_scales / grad_bias); src/bcnn_learner.c:67-83 (SGD step).
#include "chan_reduce.h"

namespace bcnn_hip {

// ---- per-thread reduction scratch ---------------------------------------------------------------
struct Scratch {
    float* p = nullptr;
    size_t cap = 0;
    int dev = -1;
};
static thread_local Scratch g_scratch[64];  // per host thread and device

float* reduce_scratch(size_t floats) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { fprintf(stderr, "[bcnn_hip] device ordinal %d out of range\n", dev); exit(1); }
    Scratch& sc = g_scratch[dev];
    if (sc.p == nullptr || sc.cap < floats) {
        if (sc.p) HIP_CHECK(hipFree(sc.p));  // hipFree syncs the device
        size_t cap = floats < (1u << 16) ? (1u << 16) : floats * 2;
        HIP_CHECK(hipMalloc((void**)&sc.p, cap * sizeof(float)));
        sc.cap = cap;
        sc.dev = dev;
    }
    return sc.p;
}

// ---- elementwise -------------------------------------------------------------------------------
template <class Op>
__global__ __launch_bounds__(256) void map2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   size_t n, Op op) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (al) {
        const size_t n4 = n / 4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* y4 = reinterpret_cast<float4*>(y);
        for (size_t j = i; j < n4; j += stride) {
            const float4 a = x4[j];
            float4 b = y4[j];
            b.x = op(a.x, b.x); b.y = op(a.y, b.y); b.z = op(a.z, b.z); b.w = op(a.w, b.w);
            y4[j] = b;
        }
        for (size_t j = n4 * 4 + i; j < n; j += stride) y[j] = op(x[j], y[j]);
    } else {
        for (size_t j = i; j < n; j += stride) y[j] = op(x[j], y[j]);
    }
}

struct AxpyOp { float a; __device__ float operator()(float x, float y) const { return __fadd_rn(__fmul_rn(x, a), y); } };
struct CopyOp { __device__ float operator()(float x, float) const { return x; } };
struct ScalOp { float a; __device__ float operator()(float, float y) const { return y * a; } };

// ---- per-channel maps --------------------------------------------------------------------------
// mode 0: y += bias[c] (skipped for bias == 0.0f / 1.0f exactly, bcnn_add_scalar quirk)
// mode 1: y *= scale[c] (scale == 0 -> 0, scale == 1 -> untouched, bcnn_scal)
struct ChanMapBody {
    float* y;
    const float* p;
    int mode;
    bool al;  // 16-byte accesses allowed
    __device__ float one(float v, float q) const {
        if (mode == 0) return (q != 0.0f && q != 1.0f) ? v + q : v;
        return (q == 0.0f) ? 0.f : ((q != 1.0f) ? v * q : v);
    }
    __device__ void operator()(unsigned off, int c, int cnt) const {
        const float q = p[c];
        if (cnt == 4 && al && (off & 3u) == 0) {
            float4 v = *reinterpret_cast<float4*>(y + off);
            v.x = one(v.x, q); v.y = one(v.y, q); v.z = one(v.z, q); v.w = one(v.w, q);
            *reinterpret_cast<float4*>(y + off) = v;
        } else {
            for (int k = 0; k < cnt; ++k) y[off + k] = one(y[off + k], q);
        }
    }
};

// ---- per-channel reductions ---------------------------------------------------------------------
struct SumF {
    static constexpr int kInFlight = 4;  // chan_reduce_partial's unroll
    const float* g;
    __device__ void operator()(long long off, int, float (&acc)[1]) const { acc[0] += g[off]; }
    __device__ void vec4(long long off, int, float (&acc)[1]) const {
        const float4 v = *reinterpret_cast<const float4*>(g + off);
        acc[0] += (v.x + v.y) + (v.z + v.w);
    }
};
// g = dy * act'(y) written back over dy, and summed: the activation backward and the bias gradient of a layer in
// one sweep (the depthwise node runs them back to back, bcnn_depthwise_conv_layer.c:311-317)
struct ActBwdSumF {
    static constexpr int kInFlight = 4;  // chan_reduce_partial's unroll
    const float* y;
    float* dy;
    int act;
    __device__ void operator()(long long off, int, float (&acc)[1]) const {
        const float g = dy[off] * act_bwd_cheap(y[off], act, 0.f);
        dy[off] = g;
        acc[0] += g;
    }
    __device__ void vec4(long long off, int, float (&acc)[1]) const {
        float4 g = *reinterpret_cast<const float4*>(dy + off);
        const float4 v = *reinterpret_cast<const float4*>(y + off);
        g.x *= act_bwd_cheap(v.x, act, 0.f); g.y *= act_bwd_cheap(v.y, act, 0.f);
        g.z *= act_bwd_cheap(v.z, act, 0.f); g.w *= act_bwd_cheap(v.w, act, 0.f);
        *reinterpret_cast<float4*>(dy + off) = g;
        acc[0] += (g.x + g.y) + (g.z + g.w);
    }
};
struct DotF {
    static constexpr int kInFlight = 4;  // chan_reduce_partial's unroll
    const float* g;
    const float* x;
    __device__ void operator()(long long off, int, float (&acc)[1]) const { acc[0] += g[off] * x[off]; }
    __device__ void vec4(long long off, int, float (&acc)[1]) const {
        const float4 a = *reinterpret_cast<const float4*>(g + off);
        const float4 b = *reinterpret_cast<const float4*>(x + off);
        acc[0] += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
};

// out[c] += sum over splits (fixed order, double)
__global__ void chan_accumulate_kernel(const float* __restrict__ partials, int C, int splits,
                                       float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < splits; ++i) s += (double)partials[(long long)c * splits + i];
    out[c] += (float)s;
}

// ---- SGD -----------------------------------------------------------------------------------------
// one pass per buffer instead of the reference's axpy/axpy/scal sequence; identical arithmetic:
//   g' = g + (decay*B)*w ; w' = w + (-lr/B)*g' ; g'' = g'*momentum        (bcnn_learner.c:76-80)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, float* __restrict__ g, size_t n,
                                                  float wd_b, float neg_lr_b, float momentum) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i], wi = w[i];
        if (wd_b != 0.f) gi = __fadd_rn(__fmul_rn(wi, wd_b), gi);
        wi = __fadd_rn(__fmul_rn(gi, neg_lr_b), wi);
        w[i] = wi;
        g[i] = (momentum == 0.0f) ? 0.f : ((momentum == 1.0f) ? gi : gi * momentum);
    }
}

// ---- Adam -----------------------------------------------------------------------------------------
// the weights half of bcnn_adam_update_cpu (bcnn_learner.c:119-129) in one pass instead of nine BLAS-1 sweeps:
//   g += (decay*B)*w ; m = (1-b1)*g + b1*m ; v = (1-b2)*g*g + b2*v ; w += (-lr/B*mu) * m / (sqrt(v) + 1e-7) ; g = 0
// every product and sum rounded separately like the reference's AVX loops (no contraction).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float wd_b,
                                                   float one_m_b1, float b1, float one_m_b2, float b2, float step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i], wi = w[i];
        gi = __fadd_rn(__fmul_rn(wi, wd_b), gi);
        const float mi = __fadd_rn(__fmul_rn(one_m_b1, gi), __fmul_rn(b1, m[i]));
        const float vi = __fadd_rn(__fmul_rn(one_m_b2, __fmul_rn(gi, gi)), __fmul_rn(b2, v[i]));
        m[i] = mi;
        v[i] = vi;
        const float q = __fdiv_rn(mi, __fadd_rn(__fsqrt_rn(vi), 0.0000001f));
        w[i] = __fadd_rn(__fmul_rn(q, step), wi);
        g[i] = 0.f;
    }
}

// one workgroup per table entry (<= BCNN_HIP_SGD_CHUNK elements of one buffer)
__global__ __launch_bounds__(256) void sgd_chunks_kernel(const bcnn_hip_sgd_chunk* __restrict__ chunks, float wd_b,
                                                         float neg_lr_b, float momentum) {
    const bcnn_hip_sgd_chunk ch = chunks[blockIdx.x];
    const float wd = ch.use_decay ? wd_b : 0.f;
    for (unsigned i = threadIdx.x; i < ch.count; i += 256) {
        float gi = ch.g_d[i], wi = ch.w_d[i];
        if (wd != 0.f) gi = __fadd_rn(__fmul_rn(wi, wd), gi);
        wi = __fadd_rn(__fmul_rn(gi, neg_lr_b), wi);
        ch.w_d[i] = wi;
        ch.g_d[i] = (momentum == 0.0f) ? 0.f : ((momentum == 1.0f) ? gi : gi * momentum);
    }
}

// one workgroup per table entry (<= BCNN_HIP_FILL_CHUNK floats of one buffer): zero fill
__global__ __launch_bounds__(256) void zero_chunks_kernel(const bcnn_hip_fill_chunk* __restrict__ chunks) {
    const bcnn_hip_fill_chunk ch = chunks[blockIdx.x];
    float* p = ch.p_d;
    const unsigned n = ch.count;
    // scalar head up to 16-byte alignment, 16-byte stores on the body, scalar tail
    unsigned head = (unsigned)(((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4);
    if (head > n) head = n;
    if (threadIdx.x < head) p[threadIdx.x] = 0.f;
    float4* b4 = reinterpret_cast<float4*>(p + head);
    const unsigned nb = n - head, n4 = nb / 4;
    for (unsigned j = threadIdx.x; j < n4; j += 256) b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned j = n4 * 4 + threadIdx.x; j < nb; j += 256) p[head + j] = 0.f;
}

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_axpy(size_t n, float a, const float* x, float* y) {
    if (!n) return;
    map2_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, current_stream()>>>(x, y, n, AxpyOp{a});
    KERNEL_CHECK();
}

void bcnn_hip_scal(size_t n, float a, float* x) {
    if (!n) return;
    if (a == 0.0f) { HIP_CHECK(hipMemsetAsync(x, 0, n * sizeof(float), current_stream())); return; }
    if (a == 1.0f) return;
    map2_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, current_stream()>>>((const float*)x, x, n, ScalOp{a});
    KERNEL_CHECK();
}

void bcnn_hip_copy_f32(size_t n, const float* x, float* y) {
    if (!n || x == y) return;
    HIP_CHECK(hipMemcpyAsync(y, x, n * sizeof(float), hipMemcpyDeviceToDevice, current_stream()));
}

void bcnn_hip_add_bias(float* y, const float* bias, int n, int c, int hw) {
    launch_chan_map(ChanMapBody{y, bias, 0, (reinterpret_cast<uintptr_t>(y) & 15) == 0}, n, c, hw);
}

void bcnn_hip_scales(float* y, const float* scales, int n, int c, int hw) {
    launch_chan_map(ChanMapBody{y, scales, 1, (reinterpret_cast<uintptr_t>(y) & 15) == 0}, n, c, hw);
}

void bcnn_hip_grad_bias(float* dbias, const float* g, int n, int c, int hw) {
    const long long M = (long long)n * hw;
    if (!M || !c) return;
    const int splits = chan_splits(c, M);
    float* part = reduce_scratch((size_t)c * splits);
    launch_chan_reduce<1>(SumF{g}, c, hw, M, splits, part);
    chan_accumulate_kernel<<<ceil_div(c, 256), 256, 0, current_stream()>>>(part, c, splits, dbias);
    KERNEL_CHECK();
}

}  // extern "C"

namespace bcnn_hip {
// dy *= act'(y) in place and dbias += per-channel sum of the result, one pass (cheap activations, 16-byte aligned
// tensors); otherwise the two separate passes.
void activation_backward_grad_bias(const float* y, float* dy, float* dbias, int n, int c, int hw, int act) {
    const long long M = (long long)n * hw;
    if (!M || !c) return;
    const bool al = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    if (act == BCNN_HIP_ACT_NONE || !act_bwd_is_cheap(act) || act == BCNN_HIP_ACT_PRELU || !al) {
        bcnn_hip_activation_backward(y, dy, (size_t)(M * c), act, nullptr, nullptr, hw, c);
        bcnn_hip_grad_bias(dbias, dy, n, c, hw);
        return;
    }
    const int splits = chan_splits(c, M);
    float* part = reduce_scratch((size_t)c * splits);
    launch_chan_reduce<1>(ActBwdSumF{y, dy, act}, c, hw, M, splits, part);
    chan_accumulate_kernel<<<ceil_div(c, 256), 256, 0, current_stream()>>>(part, c, splits, dbias);
    KERNEL_CHECK();
}
}  // namespace bcnn_hip

extern "C" {

void bcnn_hip_grad_scales(const float* x_norm, const float* g, int n, int c, int hw, float* dscales) {
    const long long M = (long long)n * hw;
    if (!M || !c) return;
    const int splits = chan_splits(c, M);
    float* part = reduce_scratch((size_t)c * splits);
    launch_chan_reduce<1>(DotF{g, x_norm}, c, hw, M, splits, part);
    chan_accumulate_kernel<<<ceil_div(c, 256), 256, 0, current_stream()>>>(part, c, splits, dscales);
    KERNEL_CHECK();
}

void bcnn_hip_sgd_update(float* w, float* b, float* dw, float* db, size_t w_size, size_t b_size,
                         int batch_size, float lr, float momentum, float decay) {
    const float neg_lr_b = -lr / batch_size;
    if (b && db && b_size) {
        sgd_kernel<<<stream_grid(b_size, 256), 256, 0, current_stream()>>>(b, db, b_size, 0.f, neg_lr_b, momentum);
        KERNEL_CHECK();
    }
    if (w && dw && w_size) {
        sgd_kernel<<<stream_grid(w_size, 256), 256, 0, current_stream()>>>(w, dw, w_size, decay * batch_size,
                                                                         neg_lr_b, momentum);
        KERNEL_CHECK();
    }
}

void bcnn_hip_adam_update(float* w, float* b, float* dw, float* db, float* adam_m, float* adam_v, size_t w_size,
                          size_t b_size, int batch_size, int iter, float beta1, float beta2, float lr, float momentum,
                          float decay) {
    // bias correction exactly as the reference computes it on the host (bcnn_learner.c:111-112); `iter` is
    // learner->seen there, i.e. SAMPLES seen, not iterations -- kept
    const float mu = sqrtf(1.0f - powf(beta2, (float)iter + 1)) / (1.0f - powf(beta1, (float)iter + 1));
    if (b && db && b_size) {  // biases take the plain momentum step (:113-117)
        sgd_kernel<<<stream_grid(b_size, 256), 256, 0, current_stream()>>>(b, db, b_size, 0.f, -lr / batch_size,
                                                                         momentum);
        KERNEL_CHECK();
    }
    if (w && dw && w_size) {
        if (!adam_m || !adam_v) {
            fprintf(stderr, "[bcnn_hip] bcnn_hip_adam_update: moment buffers missing\n");
            exit(1);
        }
        adam_kernel<<<stream_grid(w_size, 256), 256, 0, current_stream()>>>(
            w, dw, adam_m, adam_v, w_size, decay * batch_size, 1.0f - beta1, beta1, 1.0f - beta2, beta2,
            -lr / batch_size * mu);
        KERNEL_CHECK();
    }
}

void bcnn_hip_zero_chunks(const bcnn_hip_fill_chunk* chunks_d, int num_chunks) {
    if (num_chunks <= 0) return;
    zero_chunks_kernel<<<num_chunks, 256, 0, current_stream()>>>(chunks_d);
    KERNEL_CHECK();
}

void bcnn_hip_sgd_update_chunks(const bcnn_hip_sgd_chunk* chunks_d, int num_chunks, int batch_size, float lr,
                                float momentum, float decay) {
    if (num_chunks <= 0) return;
    sgd_chunks_kernel<<<num_chunks, 256, 0, current_stream()>>>(chunks_d, decay * batch_size, -lr / batch_size,
                                                              momentum);
    KERNEL_CHECK();
}

}  // extern "C"
