// conv.hip -- C-ABI entry points of the convolution node (include/bcnn_hip.h), composing the
// implicit-GEMM kernels (conv_fwd.hip, conv_bwd.hip) with the batch-norm / activation kernels the way
// bcnn_forward_conv_layer_cpu / bcnn_backward_conv_layer_cpu do (reference bcnn_conv_layer.c:367-587).
#include "conv_common.h"
#include <cstring>
#include <vector>

namespace bcnn_hip {
void conv_forward_dispatch(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                           const ConvShape& s, int act, int raw, ConvStats* stats);
void batchnorm_forward_impl(const float* x, float* y, float* run_mean, float* run_var, const float* scales,
                            const float* bias, float* saved_mean, float* saved_var, float* x_norm, float* workspace,
                            int n, int c, int hw, int mode, int act, const ConvStats* pre,
                            const BnResidual* res, bool stats_only, const float* mean_shift = nullptr);  // batchnorm.hip
void batchnorm_backward_residual(const float* dout, const float* out, int act_res, const float* res, float* dres,
                                 size_t res_count, float* dx, const float* scales, float* dscales, float* dbias,
                                 const float* fwd_bias, const float* saved_mean, const float* saved_var, float* dmean,
                                 float* dvar, const float* workspace, int n, int c, int hw);  // batchnorm.hip
void batchnorm_backward_impl(float* dy, float* dx, const float* y, int act, const float* scales, float* dscales,
                             float* dbias, const float* saved_mean, const float* saved_var, float* dmean,
                             float* dvar, const float* workspace, int n, int c, int hw, const float* fwd_bias);
void batchnorm_backward_presummed(float* dy, const float* y, int act, const float* scales, float* dscales, float* dbias,
                                  const float* saved_mean, const float* saved_var, float* dmean, float* dvar,
                                  const float* workspace, int n, int c, int hw, const float* fwd_bias, const float* sums,
                                  int splits);  // batchnorm.hip
float* reduce_scratch(size_t floats);                                                        // blas1.hip
size_t conv_dw_workspace_floats(const ConvShape& s);
bool conv_backward_weights(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s,
                           float* workspace, size_t workspace_floats, bool want_bias);
void conv_backward_data(const float* w, const float* dy, float* dx, const ConvShape& s, DxBnSums* bs = nullptr);
// conv_dw_dma.hip: per-tap GEMM with LDS-DMA staging (the general fast path)
size_t conv_dw_dma_workspace_floats(const ConvShape& s);
bool conv_backward_weights_dma(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                               size_t workspace_floats, const BnFold* fold = nullptr);
bool conv_forward_dma(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                      const ConvShape& s, int act, int raw, ConvStats* stats, const BnFold* fold = nullptr);  // conv_igemm_dma.hip
bool conv_forward_dma_supported(const ConvShape& s);
// conv_direct.hip: LDS-free kernels for small reduction lengths (K <= 32)
bool conv_forward_direct(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                         const ConvShape& s, int act, int raw);
size_t conv_dw_direct_workspace_floats(const ConvShape& s);
bool conv_backward_weights_direct(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s,
                                  float* workspace, size_t workspace_floats);
// conv_window.hip: window-in-LDS kernels for 3x3 / s1 layers with K <= 27 (configs[1])
bool conv_forward_window(const float* x, const float* w, const float* bias, const float* slopes, float* y, const ConvShape& s,
                         int act, int raw);
bool conv_forward_stem(const float* x, const float* w, const float* bias, const float* slopes, float* y, const ConvShape& s,
                       int act, int raw, ConvStats* stats);
size_t conv_dw_window_workspace_floats(const ConvShape& s);
size_t conv_dw_stem_workspace_floats(const ConvShape& s);
bool conv_backward_weights_stem(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s, float* workspace,
                                size_t workspace_floats);
bool conv_backward_weights_window(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s,
                                  float* workspace, size_t workspace_floats);
// conv_winograd.hip: F(2x2, 3x3) for the deep 3x3 / s1 layers (false: the layer stays on the direct kernels)
bool conv_forward_winograd(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                           const ConvShape& s, int act, int raw, ConvStats* stats);
bool conv_backward_data_winograd(const float* w, const float* dy, float* dx, const ConvShape& s);
// conv_winograd_fused.hip: the same algorithm in one kernel for the wide-and-shallow layers
bool conv_forward_winograd_fused(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                                 const ConvShape& s, int act, int raw, ConvStats* stats);
bool conv_backward_data_winograd_fused(const float* w, const float* dy, float* dx, const ConvShape& s);
// conv_winograd43.hip: F(4x4, 3x3) for planes of whole 4 x 4 tiles, raw output only
bool conv_forward_winograd43(const float* x, const float* w, float* y, const ConvShape& s, int raw, ConvStats* stats);
bool conv_backward_data_winograd43(const float* w, const float* dy, float* dx, const ConvShape& s);
// few input channels (the RGB stem): padded-plane GEMM over all (c, kr, kc) rows (conv_dw_dma.hip)
size_t conv_dw_small_c_workspace_floats(const ConvShape& s);
bool conv_backward_weights_small_c(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                   size_t workspace_floats);
size_t conv_dw_winograd_fused_workspace_floats(const ConvShape& s);
// conv_winograd43_dw.hip: F(4x4, 3x3) in its transposed form for planes of whole 4 x 4 tiles
size_t conv_dw_winograd43_workspace_floats(const ConvShape& s);
bool conv_backward_weights_winograd43(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                      size_t workspace_floats);
bool conv_backward_weights_winograd_fused(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                          size_t workspace_floats);
size_t conv_dw_winograd_workspace_floats(const ConvShape& s);
bool conv_backward_weights_winograd(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                    size_t workspace_floats);

static bool conv_backward_weights_dma_timed(const float* x, const float* dy, float* dw, const ConvShape& s,
                                            float* workspace, size_t workspace_floats, const BnFold* fold = nullptr) {
    if (conv_dw_dma_workspace_floats(s) == 0) return false;
    KTimer kt(K_CONV_DW, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
              4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
    return conv_backward_weights_dma(x, dy, dw, s, workspace, workspace_floats, fold);
}

// ---- a stand-alone batch-norm node in front of a 1x1 convolution, folded into it (BnFold, conv_common.h) -------------
// The host announces the fold right before the forward / backward call it applies to (same thread): the call takes it.
static thread_local BnFold g_fold_pending = {nullptr, nullptr, nullptr, nullptr};
static BnFold take_fold() {
    const BnFold f = g_fold_pending;
    g_fold_pending = BnFold{nullptr, nullptr, nullptr, nullptr};
    return f;
}
static bool bnfold_shape_ok(const ConvShape& s) {
    return s.ksz == 1 && s.stride == 1 && s.pad == 0 && s.groups == 1 && conv_forward_dma_supported(s) &&
           conv_dw_dma_workspace_floats(s) > 0;
}
// rowc[f] = sum_c W[f][c] b[c], b[c] = bias - mean a[c] (a = scale / sqrt(var + 1e-6); the reference's bcnn_add_scalar adds
// nothing for a bias of exactly 0 or 1, bcnn_mat.c:366-412): what the folded-away constant adds to every output of filter f.
// One wave per filter.
__global__ __launch_bounds__(256) void bnfold_rowconst_kernel(const float* __restrict__ w, const BnFold fold, int F, int C,
                                                              float* __restrict__ rowc) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= F) return;
    double acc = 0.0;
    for (int c = lane; c < C; c += 64) {
        float bv = fold.bias[c];
        if (bv == 1.0f) bv = 0.f;
        const float b = bv - fold.mean[c] * bnfold_a(fold.var, fold.scales, c);
        acc += (double)w[(size_t)f * C + c] * (double)b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) rowc[f] = (float)acc;
}
struct FoldScratch { float* p = nullptr; size_t cap = 0; };
static thread_local FoldScratch g_fold_scratch[64];
static float* fold_rowconst(const BnFold& fold, const float* w, int C, int F) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { fprintf(stderr, "[bcnn_hip] device ordinal %d out of range\n", dev); exit(1); }
    FoldScratch& sc = g_fold_scratch[dev];
    if (!sc.p || sc.cap < (size_t)F) {
        if (sc.p) { HIP_CHECK(hipStreamSynchronize(current_stream())); HIP_CHECK(hipFree(sc.p)); }
        sc.cap = F < 8192 ? 8192 : (size_t)F * 2;
        HIP_CHECK(hipMalloc((void**)&sc.p, sc.cap * sizeof(float)));
    }
    bnfold_rowconst_kernel<<<ceil_div(F, 4), 256, 0, current_stream()>>>(w, fold, F, C, sc.p);
    KERNEL_CHECK();
    return sc.p;
}


// per-thread side stream for the weight-gradient GEMM of bcnn_hip_conv_backward
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    int dev = -1;
    bool pending = false;  // deferred mode: work queued whose completion the caller's stream has not been ordered behind yet
};
// 0: weight gradients on the caller's stream (default); 1: on the side stream, joined before the call returns (the round-1
// experiment); 2: on the side stream, joined when the caller says so (bcnn_hip_conv_side_join) -- bcnn_backward's mode: the
// weight gradient of a layer then runs next to the batch-norm / pooling sweeps and the data gradients of the layers in front
static thread_local int g_side_mode = 0;
bool conv_side_stream_deferred() { return g_side_mode == 2; }
static SideStream* side_stream() {
    static thread_local SideStream ss;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (ss.stream == nullptr || ss.dev != dev) {
        // lowest priority: when a CU frees up, the caller's stream (the pass's critical chain: data gradients, sweeps) gets it first
        int prio_lo = 0, prio_hi = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_CHECK(hipStreamCreateWithPriority(&ss.stream, hipStreamNonBlocking, BCNN_EXP_ENV("BCNN_HIP_SIDE_PRIO_SAME") ? prio_hi : prio_lo));
        HIP_CHECK(hipEventCreateWithFlags(&ss.ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ss.done, hipEventDisableTiming));
        ss.dev = dev;
    }
    return &ss;
}

static void conv_fwd_any(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                         const ConvShape& s, int act, int raw, ConvStats* stats = nullptr) {
    if (stats) stats->splits = 0;
    static const int window_on = BCNN_EXP_ENV("BCNN_HIP_NO_WINDOW") ? 0 : 1;  // A/B switch: the LDS-free kernels instead
    if (window_on && conv_forward_window(x, w, bias, slopes, y, s, act, raw)) return;
    if (window_on && conv_forward_stem(x, w, bias, slopes, y, s, act, raw, stats)) return;
    if (conv_forward_direct(x, w, bias, slopes, y, s, act, raw)) return;
    if (raw && conv_forward_winograd43(x, w, y, s, raw, stats)) return;
    if (conv_forward_winograd_fused(x, w, bias, slopes, y, s, act, raw, stats)) return;
    if (conv_forward_winograd(x, w, bias, slopes, y, s, act, raw, stats)) return;
    conv_forward_dispatch(x, w, bias, slopes, y, s, act, raw, stats);
}

// ---- weight packs made ahead of their use (bcnn_hip_conv_prepack) -------------------------------------------------
struct PrepackEntry {
    const float* w;
    int kind, mode;
    float* buf;
    size_t cap, floats;
    unsigned long long epoch;
    bool fresh;
};
struct PrepackStore {
    std::vector<PrepackEntry> entries;
    unsigned long long epoch = 0;
    int dev = -1;
    void* table_dev[PREPACK_KINDS][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    size_t table_cap[PREPACK_KINDS][2] = {{0, 0}, {0, 0}};
    std::vector<char> table_host[PREPACK_KINDS][2];
};
static thread_local PrepackStore g_prepack;

float* prepack_take(const float* w, int kind, int mode, size_t floats) {
    PrepackStore& st = g_prepack;
    for (PrepackEntry& e : st.entries)
        if (e.w == w && e.kind == kind && e.mode == mode && e.fresh && e.epoch == st.epoch && e.floats == floats) {
            e.fresh = false;  // single use: a second call packs the (possibly rewritten) weights itself
            return e.buf;
        }
    return nullptr;
}

static void prepack_free_all(PrepackStore& st) {
    HIP_CHECK(hipDeviceSynchronize());
    for (PrepackEntry& e : st.entries)
        if (e.buf) HIP_CHECK(hipFree(e.buf));
    st.entries.clear();
    for (int k = 0; k < PREPACK_KINDS; ++k)
        for (int m = 0; m < 2; ++m) {
            if (st.table_dev[k][m]) HIP_CHECK(hipFree(st.table_dev[k][m]));
            st.table_dev[k][m] = nullptr;
            st.table_cap[k][m] = 0;
            st.table_host[k][m].clear();
        }
}

static PrepackEntry* prepack_entry(PrepackStore& st, const float* w, int kind, int mode, size_t floats) {
    PrepackEntry* e = nullptr;
    for (PrepackEntry& c : st.entries)
        if (c.w == w && c.kind == kind && c.mode == mode) { e = &c; break; }
    if (!e) {
        st.entries.push_back(PrepackEntry{w, kind, mode, nullptr, 0, 0, 0, false});
        e = &st.entries.back();
    }
    if (e->cap < floats) {
        if (e->buf) {
            HIP_CHECK(hipStreamSynchronize(current_stream()));
            HIP_CHECK(hipFree(e->buf));
        }
        HIP_CHECK(hipMalloc((void**)&e->buf, floats * sizeof(float)));
        e->cap = floats;
    }
    e->floats = floats;
    e->epoch = st.epoch;
    e->fresh = true;
    return e;
}

// the job table of one (kind, mode) on the device; uploaded only when it differs from the one already there
static const void* prepack_table(PrepackStore& st, int kind, int mode, const void* jobs, size_t bytes) {
    std::vector<char>& host = st.table_host[kind][mode];
    if (host.size() == bytes && memcmp(host.data(), jobs, bytes) == 0) return st.table_dev[kind][mode];
    HIP_CHECK(hipStreamSynchronize(current_stream()));  // a launch still reading the old table
    if (st.table_cap[kind][mode] < bytes) {
        if (st.table_dev[kind][mode]) HIP_CHECK(hipFree(st.table_dev[kind][mode]));
        HIP_CHECK(hipMalloc(&st.table_dev[kind][mode], bytes * 2));
        st.table_cap[kind][mode] = bytes * 2;
    }
    HIP_CHECK(hipMemcpy(st.table_dev[kind][mode], jobs, bytes, hipMemcpyHostToDevice));
    host.assign((const char*)jobs, (const char*)jobs + bytes);
    return st.table_dev[kind][mode];
}
}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_conv_prepack(const bcnn_hip_conv_desc* layers, int count, int data_gradient) {
    PrepackStore& st = g_prepack;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (st.dev != dev) {  // buffers of another device are left to that device's thread / context
        if (st.dev >= 0) prepack_free_all(st);
        st.dev = dev;
    }
    ++st.epoch;  // every copy made earlier and not used is stale now
    (void)take_fold();  // a pass begins: a fold announced earlier and never consumed must not meet this pass's first convolution
    const int mode = data_gradient ? 1 : 0;
    std::vector<WinoPackJob> wj;
    std::vector<IgemmPackJob> ij;
    int wmax = 0, imax = 0;
    for (int i = 0; i < count; ++i) {
        const bcnn_hip_conv_desc& d = layers[i];
        if (!d.w_d || d.groups <= 0 || d.n <= 0) continue;
        const ConvShape s = make_conv_shape(d.n, d.c, d.h, d.w, d.f, d.k, d.stride, d.pad, d.groups);
        if (s.total_q <= 0 || s.Mg == 0 || s.Cg == 0) continue;
        // the order bcnn_hip_conv_forward / _backward try their kernels in (conv_fwd_any, conv_backward_impl): the kernels
        // for few input channels (window, stem, direct, small-C dX) read the weights as they are
        WinoPackJob w1;
        IgemmPackJob i1;
        memset(&w1, 0, sizeof(w1));  // padding bytes too: the tables are compared bytewise against the uploaded ones
        memset(&i1, 0, sizeof(i1));
        size_t floats = 0;
        if (wino_fused_pack_plan(s, mode, &w1, &floats)) {
            PrepackEntry* e = prepack_entry(st, d.w_d, PREPACK_WINO, mode, floats);
            w1.w = d.w_d; w1.u = e->buf;
            if (w1.blocks > wmax) wmax = w1.blocks;
            wj.push_back(w1);
        } else if (conv_winograd_unfused_takes(s)) {
            continue;  // transforms its weights inside its own first kernel
        } else if (mode == 1 && !s.pointwise && s.groups == 1 && s.K <= 32 && s.Mg >= 32) {
            continue;  // conv_backward_data_small_c
        } else if (dma_pack_plan(s, mode, &i1, &floats)) {
            PrepackEntry* e = prepack_entry(st, d.w_d, PREPACK_IGEMM, mode, floats);
            i1.w = d.w_d; i1.at = e->buf;
            const int blocks = i1.gx * i1.gy * i1.gz;
            if (blocks > imax) imax = blocks;
            ij.push_back(i1);
        }
    }
    if (!wj.empty()) {
        const void* t = prepack_table(st, PREPACK_WINO, mode, wj.data(), wj.size() * sizeof(WinoPackJob));
        wino_fused_pack_launch((const WinoPackJob*)t, (int)wj.size(), wmax);
    }
    if (!ij.empty()) {
        const void* t = prepack_table(st, PREPACK_IGEMM, mode, ij.data(), ij.size() * sizeof(IgemmPackJob));
        dma_pack_launch((const IgemmPackJob*)t, (int)ij.size(), imax);
    }
}

void bcnn_hip_conv_prepack_discard(void) { ++g_prepack.epoch; }

void bcnn_hip_conv_prepack_reset(void) {
    PrepackStore& st = g_prepack;
    if (st.dev < 0) return;
    prepack_free_all(st);
    ++st.epoch;
}

size_t bcnn_hip_conv_workspace_size(int n, int c, int h, int w, int f, int k, int stride, int pad, int groups) {
    const ConvShape s = make_conv_shape(n, c, h, w, f, k, stride, pad, groups);
    size_t m = conv_dw_workspace_floats(s);
    size_t b = conv_dw_direct_workspace_floats(s);
    const size_t d = conv_dw_dma_workspace_floats(s), bw = conv_dw_window_workspace_floats(s);
    const size_t bs = conv_dw_stem_workspace_floats(s);
    if (bw > b) b = bw;
    if (bs > b) b = bs;
    size_t wg = conv_dw_winograd_workspace_floats(s);
    const size_t wgf = conv_dw_winograd_fused_workspace_floats(s);
    if (wgf > wg) wg = wgf;
    const size_t wg43 = conv_dw_winograd43_workspace_floats(s);
    if (wg43 > wg) wg = wg43;
    const size_t sc = conv_dw_small_c_workspace_floats(s);
    if (sc > wg) wg = sc;
    if (b > m) m = b;
    if (d > m) m = d;
    if (wg > m) m = wg;
    return m;
}

// res != NULL (batch_norm, TRAIN mode, cheap activations -- bcnn_hip_conv_residual_fusable): the following eltwise node
// is folded into the batch-norm apply pass, whose result goes to res_out; y is not written
static void conv_forward_impl(const float* x, const float* w, const float* bias, float* y, int n, int c, int h,
                              int wd, int f, int k, int stride, int pad, int groups, int act, const float* slopes,
                              int batch_norm, float* run_mean, float* run_var, const float* scales,
                              float* saved_mean, float* saved_var, float* x_norm, float* bn_workspace, int mode,
                              const BnResidual* res, float* res_out, bool stats_only = false) {
    const ConvShape s = make_conv_shape(n, c, h, wd, f, k, stride, pad, groups);
    const BnFold fold = take_fold();  // announced by bcnn_hip_conv_set_input_bnfold: x then is the batch-norm's INPUT
    if (fold.mean && (!batch_norm || mode != BCNN_HIP_MODE_TRAIN || !bnfold_shape_ok(s))) {
        fprintf(stderr, "[bcnn_hip] conv forward: a batch-norm fold was announced for a layer that cannot take it (ask "
                        "bcnn_hip_conv_bnfold_fusable)\n");
        exit(1);
    }
    if (!batch_norm) {
        if (act_is_cheap(act)) {
            conv_fwd_any(x, w, bias, slopes, y, s, act, /*raw=*/0);
        } else {  // tanh / softplus / logistic: bias in the epilogue, activation as a second in-place pass
            conv_fwd_any(x, w, bias, slopes, y, s, BCNN_HIP_ACT_NONE, /*raw=*/0);
            bcnn_hip_activation_forward(y, (size_t)n * f * s.OHOW, act, slopes, s.OHOW, f);
        }
        return;
    }
    // conv -> (pre-normalisation values, kept for backward) -> statistics -> normalise+scale+bias+act
    float* raw = (bn_workspace && mode != BCNN_HIP_MODE_PREDICT) ? bn_workspace : y;
    // TRAIN: the convolution epilogue also emits the per-channel sum / sum of squares of what it stores
    ConvStats st;
    st.partials = nullptr; st.splits = 0; st.capacity = 0;
    static const int fuse_stats = BCNN_EXP_ENV("BCNN_HIP_NO_FUSED_STATS") ? 0 : 1;  // A/B switch for profiling
    if (fuse_stats && mode == BCNN_HIP_MODE_TRAIN && s.total_q < 0x7fffffffLL) {
        // slots per channel: one per 64 output pixels on the GEMM paths; the fused Winograd kernel writes two per block
        // of 64 2x2 tiles, which is MORE than that when a tile covers fewer than two real pixels (H == 1 or W == 1)
        const long long tiles = (long long)n * ((s.OH + 1) / 2) * ((s.OW + 1) / 2);
        const long long slots_gemm = ceil_div(s.total_q, 64), slots_wino = 2 * ceil_div(tiles, 64);
        st.capacity = (size_t)f * (size_t)(slots_gemm > slots_wino ? slots_gemm : slots_wino) * 2;
        st.partials = reduce_scratch(st.capacity);
    }
    const float* mean_shift = nullptr;
    if (fold.mean) {
        // W z = (W diag(a)) y + W b: the GEMM reads y with column-scaled weights (packed here: a depends on this batch). The
        // constant W b is left out of the stored pre-normalisation values -- the batch-norm behind subtracts the batch mean,
        // so every later use (apply, backward, the consumers that normalise on the fly) sees raw - mean either way -- and
        // is added where it is visible: the running mean.
        trace_kernel("bnfold:fwd");
        mean_shift = fold_rowconst(fold, w, c, f);
        KTimer kt(K_CONV_FWD, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
                  4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
        ConvStats* stp = st.partials ? &st : nullptr;
        if (!conv_forward_dma(x, w, nullptr, nullptr, raw, s, BCNN_HIP_ACT_NONE, /*raw=*/1, stp, &fold)) {
            fprintf(stderr, "[bcnn_hip] conv forward: the LDS-DMA GEMM refused a folded layer\n");
            exit(1);
        }
    } else {
        conv_fwd_any(x, w, nullptr, nullptr, raw, s, BCNN_HIP_ACT_NONE, /*raw=*/1, st.partials ? &st : nullptr);
    }
    const int fused_act = (act == BCNN_HIP_ACT_PRELU) ? BCNN_HIP_ACT_NONE : act;
    // x_norm is not materialised on this path: the backward pass recomputes it from the raw convolution
    // output kept in bn_workspace (a full-tensor write and read less per layer and step).
    (void)x_norm;
    batchnorm_forward_impl(raw, res ? res_out : y, run_mean, run_var, scales, bias, saved_mean, saved_var, nullptr, raw, n, f,
                           s.OHOW, mode, fused_act, &st, res, stats_only, mean_shift);
    if (stats_only) return;
    if (act == BCNN_HIP_ACT_PRELU)
        bcnn_hip_activation_forward(y, (size_t)n * f * s.OHOW, act, slopes, s.OHOW, f);
}

int bcnn_hip_conv_bnfold_fusable(int n, int c, int h, int wd, int f) {
    if (n <= 0 || c <= 0 || h <= 0 || wd <= 0 || f <= 0) return 0;
    return bnfold_shape_ok(make_conv_shape(n, c, h, wd, f, 1, 1, 0, 1)) ? 1 : 0;
}

void bcnn_hip_conv_set_input_bnfold(const float* mean, const float* var, const float* scales, const float* bias) {
    g_fold_pending = BnFold{mean, var, scales, bias};
}

void bcnn_hip_conv_forward(const float* x, const float* w, const float* bias, float* y, int n, int c, int h,
                           int wd, int f, int k, int stride, int pad, int groups, int act, const float* slopes,
                           int batch_norm, float* run_mean, float* run_var, const float* scales,
                           float* saved_mean, float* saved_var, float* x_norm, float* bn_workspace, int mode) {
    conv_forward_impl(x, w, bias, y, n, c, h, wd, f, k, stride, pad, groups, act, slopes, batch_norm, run_mean, run_var,
                      scales, saved_mean, saved_var, x_norm, bn_workspace, mode, nullptr, nullptr);
}

// The convolution and the batch statistics (saved / running) of bcnn_hip_conv_forward, TRAIN mode, WITHOUT the apply sweep:
// the pre-normalisation values stay in bn_workspace and the consumer normalises them on the fly
void bcnn_hip_conv_forward_stats_only(const float* x, const float* w, const float* bias, int n, int c, int h, int wd, int f,
                                      int k, int stride, int pad, int groups, float* run_mean, float* run_var,
                                      const float* scales, float* saved_mean, float* saved_var, float* bn_workspace) {
    conv_forward_impl(x, w, bias, /*y=*/nullptr, n, c, h, wd, f, k, stride, pad, groups, BCNN_HIP_ACT_NONE, nullptr, 1,
                      run_mean, run_var, scales, saved_mean, saved_var, nullptr, bn_workspace, BCNN_HIP_MODE_TRAIN, nullptr,
                      nullptr, /*stats_only=*/true);
}

int bcnn_hip_conv_residual_fusable(int batch_norm, int act, int res_act, int mode, const float* bn_workspace,
                                   const float* res, const float* res_out) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return batch_norm && mode == BCNN_HIP_MODE_TRAIN && bn_workspace && act == BCNN_HIP_ACT_NONE && act_is_cheap(res_act) &&
           act_bwd_is_cheap(res_act) && res_act != BCNN_HIP_ACT_PRELU && al16(res) && al16(res_out) && al16(bn_workspace);
}

void bcnn_hip_conv_forward_residual(const float* x, const float* w, const float* bias, int n, int c, int h, int wd, int f,
                                    int k, int stride, int pad, int groups, float* run_mean, float* run_var,
                                    const float* scales, float* saved_mean, float* saved_var, float* bn_workspace,
                                    const float* res, size_t res_count, int res_act, float* res_out) {
    BnResidual r{res, res_count, res_act};
    conv_forward_impl(x, w, bias, /*y=*/nullptr, n, c, h, wd, f, k, stride, pad, groups, BCNN_HIP_ACT_NONE, nullptr, 1,
                      run_mean, run_var, scales, saved_mean, saved_var, nullptr, bn_workspace, BCNN_HIP_MODE_TRAIN, &r,
                      res_out);
}

struct ConvResidualBwd {
    const float* out;   // the folded eltwise node's output
    const float* dout;  // and its gradient (read only)
    const float* res;   // the eltwise node's second operand
    float* dres;        // and its gradient (accumulated), may be NULL
    size_t res_count;
    int act;
};
static void conv_backward_impl(const float* x, const float* w, const float* bias, const float* y, float* dy, float* dx,
                               float* dw, float* dbias, int n, int c, int h, int wd, int f, int k, int stride, int pad,
                               int groups, int act, const float* slopes, float* dslopes, int batch_norm,
                               const float* scales, float* dscales, const float* saved_mean,
                               const float* saved_var, float* dmean, float* dvar, const float* x_norm,
                               const float* bn_workspace, float* workspace, size_t workspace_elems,
                               const ConvResidualBwd* rb, DxBnSums* bs = nullptr, const float* own_sums = nullptr,
                               int own_splits = 0, bool bn_done = false) {
    const ConvShape s = make_conv_shape(n, c, h, wd, f, k, stride, pad, groups);
    const size_t ysize = (size_t)n * f * s.OHOW;
    const BnFold fold = take_fold();  // x then is the INPUT of the batch-norm in front: d/dW of W diag(a) is (dy x^T) diag(a)
    if (fold.mean && (!batch_norm || !bnfold_shape_ok(s))) {
        fprintf(stderr, "[bcnn_hip] conv backward: a batch-norm fold was announced for a layer that cannot take it\n");
        exit(1);
    }
    if (rb) {
        // dy <- batch-norm backward of dout * act'(out): the eltwise node's backward and this node's batch-norm backward
        // in the two sweeps the latter takes alone
        batchnorm_backward_residual(rb->dout, rb->out, rb->act, rb->res, rb->dres, rb->res_count, dy, scales, dscales, dbias,
                                    bias, saved_mean, saved_var, dmean, dvar, bn_workspace, n, f, s.OHOW);
    } else if (batch_norm && bn_done) {
        // dy already is the gradient of the pre-normalisation output (bcnn_hip_maxpool_bn_backward wrote it)
    } else if (batch_norm) {
        int fused_act = act;
        if (act == BCNN_HIP_ACT_PRELU) {
            bcnn_hip_activation_backward(y, dy, ysize, act, slopes, dslopes, s.OHOW, f);
            fused_act = BCNN_HIP_ACT_NONE;
        }
        // `bias` lets the batch-norm backward recompute the forward output from bn_workspace (no read of y)
        (void)x_norm;
        if (own_sums && own_splits > 0 && fused_act == act && act_bwd_is_cheap(act))  // whoever wrote dy left the sums
            batchnorm_backward_presummed(dy, y, act, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar, bn_workspace, n,
                                         f, s.OHOW, bias, own_sums, own_splits);
        else
            batchnorm_backward_impl(dy, nullptr, y, fused_act, scales, dscales, dbias, saved_mean, saved_var, dmean, dvar,
                                    bn_workspace, n, f, s.OHOW, bias);
    } else {
        bcnn_hip_activation_backward(y, dy, ysize, act, slopes, dslopes, s.OHOW, f);
    }
    // dW and dX only share their input dy, so the weight gradient CAN run on a private side stream to fill the
    // CUs the other kernel leaves idle in its last round (the side stream joins the caller's stream before
    // this function returns). Measured on ResNet-18 N=128 it is 1.7 % SLOWER than running them back to back
    // (18.15 vs 17.86 ms/step): both GEMMs are MFMA-bound and evict each other's L2 working set. Kept as an
    // opt-in experiment (BCNN_HIP_SIDE_STREAM=1), off by default.
    static const int side_env = BCNN_EXP_ENV("BCNN_HIP_SIDE_STREAM") ? 1 : 0;
    const int side_mode = g_side_mode ? g_side_mode : side_env;
    // deferred mode: EVERY weight gradient (the per-net workspace of split partials then belongs to the side stream alone)
    SideStream* side = (side_mode == 2 || (side_mode == 1 && dx)) ? side_stream() : nullptr;
    hipStream_t main_stream = current_stream();
    if (side) {
        HIP_CHECK(hipEventRecord(side->ready, main_stream));
        HIP_CHECK(hipStreamWaitEvent(side->stream, side->ready, 0));
        set_current_stream(side->stream);
    }
    bool bias_done;
    static const int dma_on = BCNN_EXP_ENV("BCNN_HIP_NO_DMA") ? 0 : 1;  // A/B switch for profiling
    static const int window_on = BCNN_EXP_ENV("BCNN_HIP_NO_WINDOW") ? 0 : 1;
    if (fold.mean) {
        // (the term b (x) sum_q dy of the exact derivative is left out: dy here is the gradient of a batch-norm's input, whose
        // sum over the batch is zero up to rounding -- in the reference too, where it multiplies the same b)
        trace_kernel("bnfold:dw");
        if (!conv_backward_weights_dma_timed(x, dy, dw, s, workspace, workspace_elems, &fold)) {
            fprintf(stderr, "[bcnn_hip] conv backward: the LDS-DMA weight-gradient kernel refused a folded layer\n");
            exit(1);
        }
        bias_done = false;
    } else if (window_on && conv_backward_weights_window(x, dy, dw, batch_norm ? nullptr : dbias, s, workspace, workspace_elems))
        bias_done = true;
    else if (window_on && conv_backward_weights_stem(x, dy, dw, batch_norm ? nullptr : dbias, s, workspace, workspace_elems))
        bias_done = true;
    else if (conv_backward_weights_direct(x, dy, dw, batch_norm ? nullptr : dbias, s, workspace, workspace_elems))
        bias_done = true;
    else if (conv_backward_weights_winograd43(x, dy, dw, s, workspace, workspace_elems))
        bias_done = false;
    else if (conv_backward_weights_winograd_fused(x, dy, dw, s, workspace, workspace_elems))
        bias_done = false;
    else if (conv_backward_weights_winograd(x, dy, dw, s, workspace, workspace_elems))
        bias_done = false;
    else if (dma_on && conv_backward_weights_dma_timed(x, dy, dw, s, workspace, workspace_elems))
        bias_done = false;
    else if (dma_on && conv_dw_small_c_workspace_floats(s) > 0) {
        KTimer kt(K_CONV_DW, 2.0 * (double)s.total_q * s.Mg * s.K * s.groups,
                  4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW));
        conv_backward_weights_small_c(x, dy, dw, s, workspace, workspace_elems);
        bias_done = false;
    }
    else
        bias_done = conv_backward_weights(x, dy, dw, dbias, s, workspace, workspace_elems,
                                          /*want_bias=*/!batch_norm);
    if (side) {
        HIP_CHECK(hipEventRecord(side->done, side->stream));
        set_current_stream(main_stream);
    }
    if (!batch_norm && !bias_done) bcnn_hip_grad_bias(dbias, dy, n, f, s.OHOW);  // uses the shared reduce scratch
    if (bs) bs->splits = 0;
    if (dx && !conv_backward_data_winograd43(w, dy, dx, s) && !conv_backward_data_winograd_fused(w, dy, dx, s) &&
        !conv_backward_data_winograd(w, dy, dx, s))
        conv_backward_data(w, dy, dx, s, bs);
    if (side) {
        if (side_mode == 2) side->pending = true;
        else HIP_CHECK(hipStreamWaitEvent(main_stream, side->done, 0));
    }
}

int bcnn_hip_conv_side_stream_mode(int mode) {
    const int prev = g_side_mode;
    if (mode >= 0 && mode <= 2) g_side_mode = mode;
    return prev;
}

void bcnn_hip_conv_side_join(void) {
    SideStream* ss = side_stream();
    if (!ss->pending) return;
    ss->pending = false;
    HIP_CHECK(hipStreamWaitEvent(current_stream(), ss->done, 0));  // `done` was recorded behind the last weight-gradient launch
}

void bcnn_hip_conv_backward(const float* x, const float* w, const float* bias, const float* y, float* dy, float* dx,
                            float* dw, float* dbias, int n, int c, int h, int wd, int f, int k, int stride, int pad,
                            int groups, int act, const float* slopes, float* dslopes, int batch_norm,
                            const float* scales, float* dscales, const float* saved_mean,
                            const float* saved_var, float* dmean, float* dvar, const float* x_norm,
                            const float* bn_workspace, float* workspace, size_t workspace_elems) {
    conv_backward_impl(x, w, bias, y, dy, dx, dw, dbias, n, c, h, wd, f, k, stride, pad, groups, act, slopes, dslopes,
                       batch_norm, scales, dscales, saved_mean, saved_var, dmean, dvar, x_norm, bn_workspace, workspace,
                       workspace_elems, nullptr);
}

size_t bcnn_hip_conv_bnsums_size(int n, int c, int h, int wd) {
    return (size_t)c * (size_t)ceil_div((long long)n * h * wd, 64) * 2;
}

int bcnn_hip_conv_backward_bnsums(const float* x, const float* w, const float* bias, const float* y, float* dy, float* dx,
                                  float* dw, float* dbias, int n, int c, int h, int wd, int f, int k, int stride, int pad,
                                  int groups, int act, const float* slopes, float* dslopes, int batch_norm,
                                  const float* scales, float* dscales, const float* saved_mean, const float* saved_var,
                                  float* dmean, float* dvar, const float* x_norm, const float* bn_workspace,
                                  float* workspace, size_t workspace_elems, const float* prev_y, const float* prev_mean,
                                  float* sums, size_t sums_floats) {
    DxBnSums bs{prev_y, prev_mean, sums, sums_floats, 0};
    conv_backward_impl(x, w, bias, y, dy, dx, dw, dbias, n, c, h, wd, f, k, stride, pad, groups, act, slopes, dslopes,
                       batch_norm, scales, dscales, saved_mean, saved_var, dmean, dvar, x_norm, bn_workspace, workspace,
                       workspace_elems, nullptr, (sums && prev_y && prev_mean) ? &bs : nullptr);
    return bs.splits;
}

void bcnn_hip_conv_backward_bn_done(const float* x, const float* w, float* dy, float* dx, float* dw, int n, int c, int h, int wd,
                                    int f, int k, int stride, int pad, int groups, float* workspace, size_t workspace_elems) {
    conv_backward_impl(x, w, nullptr, nullptr, dy, dx, dw, nullptr, n, c, h, wd, f, k, stride, pad, groups, BCNN_HIP_ACT_NONE,
                       nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace,
                       workspace_elems, nullptr, nullptr, nullptr, 0, /*bn_done=*/true);
}

int bcnn_hip_conv_backward_presummed(const float* x, const float* w, const float* bias, const float* y, float* dy, float* dx,
                                     float* dw, float* dbias, int n, int c, int h, int wd, int f, int k, int stride, int pad,
                                     int groups, int act, const float* slopes, float* dslopes, int batch_norm,
                                     const float* scales, float* dscales, const float* saved_mean, const float* saved_var,
                                     float* dmean, float* dvar, const float* x_norm, const float* bn_workspace,
                                     float* workspace, size_t workspace_elems, const float* own_sums, int own_splits,
                                     const float* prev_y, const float* prev_mean, float* prev_sums, size_t prev_sums_floats) {
    DxBnSums bs{prev_y, prev_mean, prev_sums, prev_sums_floats, 0};
    conv_backward_impl(x, w, bias, y, dy, dx, dw, dbias, n, c, h, wd, f, k, stride, pad, groups, act, slopes, dslopes,
                       batch_norm, scales, dscales, saved_mean, saved_var, dmean, dvar, x_norm, bn_workspace, workspace,
                       workspace_elems, nullptr, (prev_sums && prev_y && prev_mean) ? &bs : nullptr, own_sums, own_splits);
    return bs.splits;
}

void bcnn_hip_conv_backward_residual(const float* x, const float* w, const float* bias, float* dy, float* dx, float* dw,
                                     float* dbias, int n, int c, int h, int wd, int f, int k, int stride, int pad,
                                     int groups, const float* scales, float* dscales, const float* saved_mean,
                                     const float* saved_var, float* dmean, float* dvar, const float* bn_workspace,
                                     float* workspace, size_t workspace_elems, const float* res_out,
                                     const float* dres_out, int res_act, const float* res, float* dres, size_t res_count) {
    ConvResidualBwd rb{res_out, dres_out, res, dres, res_count, res_act};
    conv_backward_impl(x, w, bias, nullptr, dy, dx, dw, dbias, n, c, h, wd, f, k, stride, pad, groups, BCNN_HIP_ACT_NONE,
                       nullptr, nullptr, 1, scales, dscales, saved_mean, saved_var, dmean, dvar, nullptr, bn_workspace,
                       workspace, workspace_elems, &rb);
}

}  // extern "C"
