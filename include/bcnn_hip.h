/*
 * bcnn_hip.h -- C-ABI of the MI355X (gfx950) back-end for bcnn's conv/GEMM hot path.
 *
 * This is the drop-in boundary: a C99 host (bcnn_net / bcnn_node, the reference's own layer code)
 * links against libbcnn_hip.so and calls these entry points where the reference's CUDA build calls
 * its `bcnn_cuda_*` helper family and per-layer `*_gpu` workers. Only plain pointers, ints, floats
 * and size_t cross the boundary; every pointer named `*_d` / documented "device" is HBM memory
 * obtained from bcnn_hip_malloc_*. Unlike the reference helpers, operator entry points take
 * WHOLE-BATCH shapes (one launch per direction, not one cublasSgemm per image).
 *
 * Error convention (reference: src/bcnn_utils.h:174-195): a failing HIP call prints
 * file:line + the HIP error string to stderr and exit()s; entry points therefore return void.
 *
 * Each declaration cites the reference interface it replaces (file:line in jnbraun/bcnn).
 * All tensors: dense NCHW fp32, idx(n,c,h,w) = ((n*C + c)*H + h)*W + w.
 */
#ifndef BCNN_HIP_H
#define BCNN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum bcnn_activation values (inc/bcnn/bcnn.h:164-175) -- passed as int `act`. */
enum {
    BCNN_HIP_ACT_NONE = 0, BCNN_HIP_ACT_TANH, BCNN_HIP_ACT_RELU, BCNN_HIP_ACT_RAMP,
    BCNN_HIP_ACT_SOFTPLUS, BCNN_HIP_ACT_LRELU, BCNN_HIP_ACT_ABS, BCNN_HIP_ACT_CLAMP,
    BCNN_HIP_ACT_PRELU, BCNN_HIP_ACT_LOGISTIC
};
/* enum bcnn_mode values (inc/bcnn/bcnn.h:105-112) -- passed as int `mode`. */
enum { BCNN_HIP_MODE_PREDICT = 0, BCNN_HIP_MODE_TRAIN = 1, BCNN_HIP_MODE_VALID = 2 };

/* ---------------------------------------------------------------------------------------------
 * Runtime: device, memory, stream, events.  Replaces src/bcnn_utils.h:197-213 / bcnn_utils.c:101-201
 * (bcnn_cuda_set_device, bcnn_cuda_malloc_f32/i32, bcnn_cuda_memcpy_*, bcnn_cuda_free,
 *  bcnn_cuda_fill_f32) -- no library-global handle singletons, one explicit stream per thread.
 * ------------------------------------------------------------------------------------------- */
int bcnn_hip_device_count(void);
void bcnn_hip_set_device(int id);                 /* bcnn_cuda_set_device, bcnn_utils.c:201 */
int bcnn_hip_get_device(void);
const char *bcnn_hip_device_name(void);           /* gcnArchName of the current device, e.g. "gfx950:..." */
float *bcnn_hip_malloc_f32(size_t n);             /* bcnn_cuda_malloc_f32, bcnn_utils.c:133-140; zero-filled here */
int *bcnn_hip_malloc_i32(size_t n);               /* bcnn_cuda_malloc_i32, bcnn_utils.c:124-131; zero-filled here */
void bcnn_hip_free(void *p_d);                    /* bcnn_cuda_free, bcnn_utils.c:182-185 */
void bcnn_hip_memcpy_h2d(void *dst_d, const void *src_h, size_t bytes); /* bcnn_cuda_memcpy_host2dev */
void bcnn_hip_memcpy_d2h(void *dst_h, const void *src_d, size_t bytes); /* bcnn_cuda_memcpy_dev2host */
void bcnn_hip_memcpy_d2d(void *dst_d, const void *src_d, size_t bytes);
void bcnn_hip_fill_f32(float *x_d, size_t n, float value); /* bcnn_cuda_fill_f32, bcnn_mat.cu:52-60 */
void bcnn_hip_sync(void);                         /* waits for the calling thread's stream */
/* The stream all entry points launch on (per host thread). NULL = the device's null stream, which
 * is also PyTorch-ROCm's default stream. bcnn_hip_stream_create returns an opaque hipStream_t. */
void *bcnn_hip_stream_create(void);
void bcnn_hip_stream_destroy(void *stream);
void bcnn_hip_set_stream(void *stream);
void *bcnn_hip_get_stream(void);
/* HIP events recorded on the CURRENT stream (used by bench.py for per-kernel durations). */
void *bcnn_hip_event_create(void);
void bcnn_hip_event_destroy(void *ev);
void bcnn_hip_event_record(void *ev);
void bcnn_hip_event_sync(void *ev);
float bcnn_hip_event_elapsed_ms(void *start, void *stop);

/* Per-kernel-class timing (measurement aid, off by default): while enabled every launch of a classified
 * kernel is bracketed by HIP events on the launch stream together with its ALGORITHMIC flops / bytes;
 * read() sums them per class since the last reset(). Classes: bcnn_hip_profile_class_name(i). */
void bcnn_hip_profile_enable(int on);
void bcnn_hip_profile_reset(void);
int bcnn_hip_profile_num_classes(void);
const char *bcnn_hip_profile_class_name(int cls);
void bcnn_hip_profile_read(int cls, double *ms, long long *launches, double *flops, double *bytes);
/* the part of the class's `flops` that is not tile padding (Winograd classes on odd-sized planes; otherwise == flops) */
double bcnn_hip_profile_read_useful_flops(int cls);

/* Dispatch trace (test aid, off by default): while enabled every dispatcher decision of the calling thread appends the name
 * of the kernel family it launched ("wino43_kernel", "wino43_tail_fixup", "wino_fused_kernel", "maxpool_bwd_pair_bn", ...),
 * one per line, so a parity test can assert WHICH kernels produced the tensors it compares instead of assuming the
 * size-dependent dispatch rules. enable(1) clears the log; read() copies at most cap - 1 bytes + a terminating 0 and
 * returns the full length of the log. */
/* Weight gradients on a side stream (per host thread). mode 0: off (default); 1: bcnn_hip_conv_backward* queue the
 * weight-gradient kernels of a layer that also has a data gradient on a private stream and order the caller's stream behind
 * them before returning; 2: ... and do NOT order it: the caller does, with bcnn_hip_conv_side_join(), before anything reads
 * the weight gradients (bcnn_backward: at its end, and before every gradient-ready callback). Returns the previous mode. */
int bcnn_hip_conv_side_stream_mode(int mode);
void bcnn_hip_conv_side_join(void);

void bcnn_hip_trace_enable(int on);
size_t bcnn_hip_trace_read(char *buf, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * BLAS-1 / per-channel helpers.  Replaces bcnn_cuda_axpy/scal/copy (bcnn_mat.cu:44-100),
 * bcnn_cuda_add_bias / bcnn_cuda_grad_bias (bcnn_mat.cu:348-391), bcnn_scales_gpu /
 * bcnn_grad_scales_gpu (bcnn_mat.cu:393-438); CPU semantics bcnn_mat.c:52-115, 319-412, 761-811.
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_axpy(size_t n, float a, const float *x_d, float *y_d);   /* y += a*x */
void bcnn_hip_scal(size_t n, float a, float *x_d);                      /* x *= a (a==0 -> zero-fill) */
void bcnn_hip_copy_f32(size_t n, const float *x_d, float *y_d);
/* y[b][c][:] += bias[c]; channels whose bias is exactly 0.0f or 1.0f are left untouched, as in
 * the reference AVX build (bcnn_add_scalar, bcnn_mat.c:366-412). */
void bcnn_hip_add_bias(float *y_d, const float *bias_d, int n, int c, int hw);
void bcnn_hip_grad_bias(float *dbias_d, const float *g_d, int n, int c, int hw);  /* dbias[c] += sum */
void bcnn_hip_scales(float *y_d, const float *scales_d, int n, int c, int hw);
void bcnn_hip_grad_scales(const float *x_norm_d, const float *g_d, int n, int c, int hw,
                          float *dscales_d);                                     /* dscales[c] += sum g*x_norm */

/* ---------------------------------------------------------------------------------------------
 * GEMM / im2col / col2im.  Replaces bcnn_cuda_gemm (bcnn_mat.cu:31-42), bcnn_cuda_im2col (:477-524),
 * bcnn_cuda_col2im (:526-576); semantics of bcnn_gemm (bcnn_mat.c:2627-2650), bcnn_im2col (:817-854),
 * bcnn_col2im (:935-970, zero-fills then scatter-adds => OVERWRITES im).
 * Row-major C[m x n] = alpha * op(A) * op(B) + beta * C on the fp32 MFMA path.
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_gemm(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a_d,
                   int lda, const float *b_d, int ldb, float beta, float *c_d, int ldc);
void bcnn_hip_im2col(const float *im_d, int channels, int height, int width, int ksize, int pad,
                     int stride, float *col_d);
void bcnn_hip_col2im(const float *col_d, int channels, int height, int width, int ksize, int pad,
                     int stride, float *im_d);

/* ---------------------------------------------------------------------------------------------
 * Activation map.  Replaces bcnn_forward/backward_activation_gpu (bcnn_activation_layer.cu:32-113);
 * semantics of bcnn_forward_activation_cpu / bcnn_backward_activation_cpu
 * (bcnn_activation_layer.c:90-146, 165-226), incl. softplus/abs/prelu which the CUDA build lacks.
 * In place. `slopes_d` (PReLU, one per channel) may be NULL otherwise. Backward uses the
 * POST-activation values `x_d`; `dslopes_d` accumulates (PReLU only, may be NULL to skip).
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_activation_forward(float *x_d, size_t size, int act, const float *slopes_d,
                                 int spatial, int channels);
void bcnn_hip_activation_backward(const float *x_d, float *dx_d, size_t size, int act,
                                  const float *slopes_d, float *dslopes_d, int spatial, int channels);

/* ---------------------------------------------------------------------------------------------
 * Batch normalisation.  Replaces bcnn_forward/backward_batchnorm_gpu (bcnn_batchnorm_layer.cu:187-377);
 * semantics of the CPU path (bcnn_batchnorm_layer.c:196-242, 301-332): biased one-pass variance,
 * eps 1e-6 forward / 1e-5 backward, running = 0.9*running + 0.1*batch.
 *   forward : x_d -> y_d (may alias). TRAIN: writes saved_mean/saved_var, updates run_mean/run_var,
 *             keeps the pre-normalisation input in workspace_d (needed by backward; may alias x_d when
 *             x_d != y_d) and, if x_norm_d != NULL, the normalised values. VALID: normalises with the
 *             running statistics. PREDICT: y = x*scale + bias.
 *   backward: dy_d is transformed IN PLACE into the gradient w.r.t. the BN input (as the reference
 *             does) and copied to dx_d if dx_d != NULL && dx_d != dy_d; dbias_d/dscales_d accumulate;
 *             dmean_d/dvar_d (saved_mean.grad / saved_variance.grad) are overwritten.
 *             x_norm_d may be NULL (recomputed from workspace_d, mean, var).
 * `act`/`y_d` in backward: optional fused activation-backward (conv+BN+act nodes): pass the
 * post-activation output and its activation, or act = NONE.
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_batchnorm_forward(const float *x_d, float *y_d, float *run_mean_d, float *run_var_d,
                                const float *scales_d, const float *bias_d, float *saved_mean_d,
                                float *saved_var_d, float *x_norm_d, float *workspace_d, int n, int c,
                                int hw, int mode, int act);
void bcnn_hip_batchnorm_backward(float *dy_d, float *dx_d, const float *y_d, int act,
                                 const float *scales_d, float *dscales_d, float *dbias_d,
                                 const float *saved_mean_d, const float *saved_var_d, float *dmean_d,
                                 float *dvar_d, const float *x_norm_d, const float *workspace_d, int n,
                                 int c, int hw);

/* ---------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on fp32 MFMA; no materialised im2col buffer).
 * Replaces bcnn_forward_conv_layer_gpu / bcnn_backward_conv_layer_gpu (bcnn_conv_layer.c:589-792)
 * with the semantics of the CPU workers (bcnn_conv_layer.c:367-485, 487-587):
 *   - weights [f][c/groups][k][k], bias [f]; out size (h + 2*pad - k)/stride + 1;
 *   - 1x1 kernels read the source as a raw [c/groups][oh*ow] matrix regardless of stride/pad (:445-446);
 *   - forward: y = act(conv + bias) (bias skipped for channels with bias == 0.0f or 1.0f exactly), or
 *     with batch_norm != 0: y = act(BN(conv)) using the bn_* arguments (see batchnorm_forward);
 *   - backward: dy_d <- dy_d * act'(y_d) in place; then BN backward or dbias += sum; dw += (beta = 1);
 *     dx_d (if non-NULL) is OVERWRITTEN (col2im zero-fill semantics); 1x1 writes only the
 *     [c/groups][oh*ow] prefix of each image-group of dx_d.
 * workspace_d: scratch of at least bcnn_hip_conv_workspace_size(...) floats (split-K partials of dw;
 * the reference's per-net conv workspace, bcnn_net.c:337-352, plays the same role).
 * PReLU slopes: slopes_d (one per output channel), else NULL.
 * backward bias_d: the bias the forward pass used (the reference worker has it as node->src[2]); may be
 * NULL. With it the fused batch-norm backward recomputes the forward output from bn_workspace_d bit for
 * bit instead of reading y_d (one full-tensor read less in each of its two passes).
 * x_norm_d is accepted for signature parity with the reference layer and never touched: the normalised
 * values are recomputed from bn_workspace_d, saved_mean_d and saved_var_d.
 * ------------------------------------------------------------------------------------------- */
size_t bcnn_hip_conv_workspace_size(int n, int c, int h, int w, int f, int k, int stride, int pad,
                                    int groups);
void bcnn_hip_conv_forward(const float *x_d, const float *w_d, const float *bias_d, float *y_d, int n,
                           int c, int h, int w, int f, int k, int stride, int pad, int groups, int act,
                           const float *slopes_d,
                           /* fused batch-norm (batch_norm == 0: all NULL) */
                           int batch_norm, float *run_mean_d, float *run_var_d, const float *scales_d,
                           float *saved_mean_d, float *saved_var_d, float *x_norm_d,
                           float *bn_workspace_d, int mode);
void bcnn_hip_conv_backward(const float *x_d, const float *w_d, const float *bias_d, const float *y_d,
                            float *dy_d, float *dx_d, float *dw_d, float *dbias_d, int n, int c, int h, int w, int f,
                            int k, int stride, int pad, int groups, int act, const float *slopes_d,
                            float *dslopes_d, int batch_norm, const float *scales_d, float *dscales_d,
                            const float *saved_mean_d, const float *saved_var_d, float *dmean_d,
                            float *dvar_d, const float *x_norm_d, const float *bn_workspace_d,
                            float *workspace_d, size_t workspace_elems);

/* ---------------------------------------------------------------------------------------------
 * A convolution node with batch-norm (no activation) whose output is the first operand of the eltwise node that
 * follows it (the residual block: bcnn_conv_layer.c:367-485 then bcnn_eltwise_layer.c:82-113; backward
 * bcnn_eltwise_layer.c:115-152 then bcnn_conv_layer.c:487-587). For an executor that runs whole passes:
 *   bcnn_hip_conv_residual_fusable   non-zero when the pair below may replace the two workers (TRAIN mode, batch-norm
 *                                    with its pre-normalisation workspace, cheap eltwise activation, aligned tensors)
 *   bcnn_hip_conv_forward_residual   bcnn_hip_conv_forward whose batch-norm apply pass also adds res_d[0 .. res_count)
 *                                    (the reference adds its second operand to the first min_c * h * w elements only)
 *                                    and applies the eltwise activation; the result goes to res_out_d, the
 *                                    convolution node's own output tensor is NOT written (bcnn_hip_batchnorm_apply on
 *                                    bn_workspace_d produces it when somebody asks)
 *   bcnn_hip_conv_backward_residual  the eltwise backward (g = dres_out_d * act'(res_out_d), dres_d[0 .. res_count) += g)
 *                                    and bcnn_hip_conv_backward in one: dres_out_d is read, not rewritten; dy_d receives
 *                                    the batch-norm backward of g as in bcnn_hip_conv_backward. Of res_out_d only the
 *                                    first res_count elements are read: the value act' needs is recomputed from
 *                                    bn_workspace_d, bias_d and res_d with the forward's own operations
 *   bcnn_hip_batchnorm_apply         y = act((x - mean) / sqrtf(var + 1e-6) * scale + bias) with GIVEN statistics: the
 *                                    apply sweep of a TRAIN-mode forward alone, no side effects
 * ------------------------------------------------------------------------------------------- */
int bcnn_hip_conv_residual_fusable(int batch_norm, int act, int res_act, int mode, const float *bn_workspace_d,
                                   const float *res_d, const float *res_out_d);
void bcnn_hip_conv_forward_residual(const float *x_d, const float *w_d, const float *bias_d, int n, int c, int h, int w,
                                    int f, int k, int stride, int pad, int groups, float *run_mean_d, float *run_var_d,
                                    const float *scales_d, float *saved_mean_d, float *saved_var_d,
                                    float *bn_workspace_d, const float *res_d, size_t res_count, int res_act,
                                    float *res_out_d);
void bcnn_hip_conv_backward_residual(const float *x_d, const float *w_d, const float *bias_d, float *dy_d, float *dx_d,
                                     float *dw_d, float *dbias_d, int n, int c, int h, int w, int f, int k, int stride,
                                     int pad, int groups, const float *scales_d, float *dscales_d,
                                     const float *saved_mean_d, const float *saved_var_d, float *dmean_d, float *dvar_d,
                                     const float *bn_workspace_d, float *workspace_d, size_t workspace_elems,
                                     const float *res_out_d, const float *dres_out_d, int res_act, const float *res_d,
                                     float *dres_d, size_t res_count);
void bcnn_hip_batchnorm_apply(const float *x_d, float *y_d, const float *scales_d, const float *bias_d,
                              const float *saved_mean_d, const float *saved_var_d, int n, int c, int hw, int act);

/* Weight packs of a whole pass in one launch, for an executor that knows its layers (bcnn_forward / bcnn_backward).
 * The convolution kernels read the filter bank re-arranged (transformed G g G^T for the Winograd kernels, A^T per tap
 * for the LDS-DMA GEMM); bcnn_hip_conv_forward / _backward make that copy right before their kernel, one small launch
 * per call. bcnn_hip_conv_prepack makes the copies of `count` layers at once -- forward form (data_gradient = 0) or
 * data-gradient form (1) -- into buffers the library keeps; the NEXT forward (resp. backward) call for each of these
 * layers (same weight pointer and shape, this thread, this stream) uses its copy instead of packing again. A copy is
 * used at most once, and a later bcnn_hip_conv_prepack call discards every unused copy, so the caller only has to leave
 * the weights unwritten between the prepack call and the calls that consume it. Layers whose kernel needs no copy are
 * skipped. bcnn_hip_conv_prepack_reset frees the buffers (synchronises the device). */
typedef struct {
    const float *w_d;
    int n, c, h, w, f, k, stride, pad, groups;
} bcnn_hip_conv_desc;
void bcnn_hip_conv_prepack(const bcnn_hip_conv_desc *layers, int count, int data_gradient);
void bcnn_hip_conv_prepack_reset(void);
/* every copy made so far and not consumed is stale from here on (end of a pass: the weights may be rewritten next) */
void bcnn_hip_conv_prepack_discard(void);

/* ---------------------------------------------------------------------------------------------
 * Pooling.  Replaces bcnn_forward/backward_maxpool_layer_gpu (bcnn_maxpool_layer.cu:28-166) and
 * bcnn_forward/backward_avgpool_layer_gpu (bcnn_avgpool_layer.cu:29-90); CPU semantics
 * bcnn_maxpool_layer.c:145-191, 258-273 and bcnn_avgpool_layer.c:82-125.
 *   maxpool: window at (i*stride, j*stride) (bottom/right padding only), rows outer / cols inner,
 *   first strict maximum wins, NaN never wins, indexes = flat int32 offset into the whole source
 *   tensor -- BIT-EXACT with the CPU path. backward: dx[indexes[o]] += dy[o], applied in ascending
 *   output order per source element (deterministic, same order as the CPU loop). overwrite != 0:
 *   the caller guarantees dx is semantically all-zero (it skipped the zero fill of bcnn_net.c:361-375
 *   because this node is the tensor's only gradient writer): dx is assigned 0 + the same sums.
 *   avgpool: global mean per (n,c); backward dx += dy/(h*w).
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_maxpool_forward(const float *x_d, float *y_d, int *indexes_d, int n, int c, int h, int w,
                              int out_h, int out_w, int size, int stride);
void bcnn_hip_maxpool_backward(const float *dy_d, const int *indexes_d, float *dx_d, int n, int c,
                               int h, int w, int out_h, int out_w, int size, int stride, int overwrite);
/* A 1x1 convolution node whose input is the output of a stand-alone batch-norm node (MobileNet: [batchnorm] -> [conv 1x1]).
 * bcnn_hip_conv_backward_bnsums is bcnn_hip_conv_backward whose data-gradient kernel also emits, from the tile it stores,
 * the per-channel partial sums the batch-norm node's backward starts with (S1 = sum dz, S2 = sum dz * (prev_y - prev_mean),
 * bcnn_batchnorm_layer.c:263-281; prev_y_d = that node's INPUT, prev_mean_d its saved mean) into sums_d
 * (bcnn_hip_conv_bnsums_size floats). Returns the number of partials per channel, 0 when this shape's kernel does not
 * emit them (then run bcnn_hip_batchnorm_backward_sums). bcnn_hip_batchnorm_backward_finalize turns the partials into
 * dbias / dscales (accumulated) and dmean / dvar (written), like the finalize step of bcnn_hip_batchnorm_backward_sums. */
size_t bcnn_hip_conv_bnsums_size(int n, int c, int h, int w);
int bcnn_hip_conv_backward_bnsums(const float *x_d, const float *w_d, const float *bias_d, const float *y_d, float *dy_d,
                                  float *dx_d, float *dw_d, float *dbias_d, int n, int c, int h, int w, int f, int k,
                                  int stride, int pad, int groups, int act, const float *slopes_d, float *dslopes_d,
                                  int batch_norm, const float *scales_d, float *dscales_d, const float *saved_mean_d,
                                  const float *saved_var_d, float *dmean_d, float *dvar_d, const float *x_norm_d,
                                  const float *bn_workspace_d, float *workspace_d, size_t workspace_elems,
                                  const float *prev_y_d, const float *prev_mean_d, float *sums_d, size_t sums_floats);
/* The other direction of the same idea, for a convolution node with batch-norm whose output feeds a depthwise node that
 * reads its pre-normalisation tensor (bcnn_hip_depthwise_backward_bnin / _bn_bnin below): that depthwise kernel writes the
 * gradient of the convolution node's OUTPUT and holds the pre-normalisation values it belongs to, so
 * bcnn_hip_depthwise_backward_bnin_sums (bn_mean_d == NULL: no batch-norm node behind the depthwise node, dy_d is updated
 * in place; else dy_d is that node's output gradient, read only) also leaves S1 = sum g, S2 = sum g * (raw - mean) with
 * g = dx * act'(act(bn(raw))) per channel in in_sums_d (bcnn_hip_depthwise_insums_size floats) and returns the number of
 * partials per channel (0: not emitted -- overwrite == 0, or the buffer is too small). bcnn_hip_conv_backward_presummed is
 * bcnn_hip_conv_backward_bnsums (prev_* may be NULL) that takes those partials (own_sums_d, own_splits > 0) instead of
 * sweeping (dy, pre-normalisation tensor) for them; with own_splits == 0 it is bcnn_hip_conv_backward_bnsums. */
size_t bcnn_hip_depthwise_insums_size(int n, int c, int h, int w, int k, int stride, int pad);
int bcnn_hip_depthwise_backward_bnin_sums(const float *x_raw_d, const float *w_d, const float *y_d, float *dy_d, float *dx_d,
                                          float *dw_d, float *dbias_d, int n, int c, int h, int w, int k, int stride, int pad,
                                          int act, int overwrite, const float *bn_mean_d, const float *bn_var_d,
                                          const float *bn_scales_d, const float *bn_dmean_d, const float *bn_dvar_d,
                                          const float *in_mean_d, const float *in_var_d, const float *in_scale_d,
                                          const float *in_bias_d, int in_act, float *in_sums_d, size_t in_sums_floats);
int bcnn_hip_conv_backward_presummed(const float *x_d, const float *w_d, const float *bias_d, const float *y_d, float *dy_d,
                                     float *dx_d, float *dw_d, float *dbias_d, int n, int c, int h, int w, int f, int k,
                                     int stride, int pad, int groups, int act, const float *slopes_d, float *dslopes_d,
                                     int batch_norm, const float *scales_d, float *dscales_d, const float *saved_mean_d,
                                     const float *saved_var_d, float *dmean_d, float *dvar_d, const float *x_norm_d,
                                     const float *bn_workspace_d, float *workspace_d, size_t workspace_elems,
                                     const float *own_sums_d, int own_splits, const float *prev_y_d,
                                     const float *prev_mean_d, float *prev_sums_d, size_t prev_sums_floats);
void bcnn_hip_batchnorm_backward_finalize(const float *sums_d, int splits, const float *scales_d, float *dscales_d,
                                          float *dbias_d, const float *saved_var_d, float *dmean_d, float *dvar_d, int c);

/* The pooling node behind a convolution node with batch-norm (the ResNet stem), for an executor that runs whole passes:
 * bcnn_hip_maxpool_forward over act(batch-norm(x_d)) computed on the fly from the convolution node's pre-normalisation
 * output and saved statistics -- the values, scan order and indexes of bcnn_hip_batchnorm_apply followed by
 * bcnn_hip_maxpool_forward, without writing and re-reading the normalised tensor. Ask bcnn_hip_maxpool_bn_fusable
 * (stride 2, size 2 or 3, w % 4 == 0, cheap activation). bcnn_hip_conv_forward_stats_only is the convolution node's part:
 * bcnn_hip_conv_forward (batch-norm, TRAIN mode) up to and including the batch statistics, without the apply sweep. */
void bcnn_hip_conv_forward_stats_only(const float *x_d, const float *w_d, const float *bias_d, int n, int c, int h, int w,
                                      int f, int k, int stride, int pad, int groups, float *run_mean_d, float *run_var_d,
                                      const float *scales_d, float *saved_mean_d, float *saved_var_d,
                                      float *bn_workspace_d);
int bcnn_hip_maxpool_bn_fusable(int n, int c, int h, int w, int out_h, int out_w, int size, int stride, int act,
                                const float *x_d);
void bcnn_hip_maxpool_forward_bn(const float *x_d, float *y_d, int *indexes_d, int n, int c, int h, int w, int out_h,
                                 int out_w, int size, int stride, const float *scales_d, const float *bias_d,
                                 const float *saved_mean_d, const float *saved_var_d, int act);
/* ... and its backward. bcnn_hip_maxpool_forward_bn_keep is bcnn_hip_maxpool_forward_bn that also keeps, per pooled
 * element, the pre-normalisation value that won its window (raw_at_max_d, the pooled shape). With it
 * bcnn_hip_maxpool_bn_backward does the pooling node's backward AND the convolution node's batch-norm backward in a sweep
 * over the pooled tensors plus ONE sweep over the un-pooled ones: the sums S1, S2 of bcnn_batchnorm_layer.c:263-281 are
 * taken over (dpool_d, raw_at_max_d) -- the un-pooled gradient is zero except at the winning places, where it is the sum
 * of the pooled gradients that selected them -- into dscales_d / dbias_d (accumulated) and dmean_d / dvar_d (written),
 * and one kernel then gathers each un-pooled gradient value like bcnn_hip_maxpool_backward (overwrite form) and applies
 * :292-296 to it in registers, writing dx_d = the gradient of the convolution's PRE-NORMALISATION output (what
 * bcnn_hip_conv_backward leaves in dy_d after its batch-norm step). bcnn_hip_conv_backward_bn_done is the rest of that
 * node's backward: weight gradient and, if dx_d != NULL, data gradient from such a dy_d (no bias gradient: the bias of a
 * batch-norm convolution is the batch-norm's). Ask bcnn_hip_maxpool_bn_backward_fusable (3x3 / stride 2, out_w == w / 2,
 * cheap activation, aligned pointers; raw_d = the convolution's pre-normalisation output). */
void bcnn_hip_maxpool_forward_bn_keep(const float *x_d, float *y_d, int *indexes_d, int n, int c, int h, int w, int out_h,
                                      int out_w, int size, int stride, const float *scales_d, const float *bias_d,
                                      const float *saved_mean_d, const float *saved_var_d, int act, float *raw_at_max_d);
int bcnn_hip_maxpool_bn_backward_fusable(int n, int c, int h, int w, int out_h, int out_w, int size, int stride, int act,
                                         const float *raw_d, const float *dpool_d, const int *indexes_d, const float *dx_d);
void bcnn_hip_maxpool_bn_backward(const float *dpool_d, const int *indexes_d, const float *raw_at_max_d, const float *raw_d,
                                  float *dx_d, int n, int c, int h, int w, int out_h, int out_w, int size, int stride,
                                  const float *scales_d, float *dscales_d, const float *bias_d, float *dbias_d,
                                  const float *saved_mean_d, const float *saved_var_d, float *dmean_d, float *dvar_d, int act);
void bcnn_hip_conv_backward_bn_done(const float *x_d, const float *w_d, float *dy_d, float *dx_d, float *dw_d, int n, int c,
                                    int h, int w, int f, int k, int stride, int pad, int groups, float *workspace_d,
                                    size_t workspace_elems);
void bcnn_hip_avgpool_forward(const float *x_d, float *y_d, int n, int c, int h, int w);
void bcnn_hip_avgpool_backward(const float *dy_d, float *dx_d, int n, int c, int h, int w);

/* ---------------------------------------------------------------------------------------------
 * Depthwise convolution.  Replaces the kernels of bcnn_depthwise_conv_layer.cu:33-200; CPU semantics
 * bcnn_depthwise_conv_layer.c:165-293, 295-547: weights [c][k][k]; y = act(dwconv + bias);
 * backward: dy *= act'(y) in place, dbias += ..., and ONLY IF dx_d != NULL: dw += ..., dx += ...
 * (accumulating, no zero-fill). No unsynchronised `+=` races (the CUDA kernel at :113 has one).
 * overwrite != 0 (same contract as bcnn_hip_maxpool_backward): the caller guarantees dx is semantically all-zero
 * (it skipped the executor's zero fill because this node is the tensor's only gradient writer): dx is assigned
 * 0 + the same sums, without being read.
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_depthwise_forward(const float *x_d, const float *w_d, const float *bias_d, float *y_d,
                                int n, int c, int h, int w, int k, int stride, int pad, int act);
void bcnn_hip_depthwise_backward(const float *x_d, const float *w_d, const float *y_d, float *dy_d,
                                 float *dx_d, float *dw_d, float *dbias_d, int n, int c, int h, int w,
                                 int k, int stride, int pad, int act, int overwrite);

/* ---------------------------------------------------------------------------------------------
 * A depthwise layer whose output feeds a stand-alone batch-norm node (the MobileNet block, reference
 * examples/... mobilenet configs: [depthwise-conv] -> [batchnorm] -> [conv 1x1]). The reference runs the two nodes'
 * workers back to back (bcnn_depthwise_conv_layer.c:165-293 then bcnn_batchnorm_layer.c:196-242; backward
 * bcnn_batchnorm_layer.c:301-332 then bcnn_depthwise_conv_layer.c:295-547); these entry points let an executor that
 * runs whole passes share work between the two workers. Results are those of the separate calls (forward, dx: the
 * same operations in the same order; statistics and gradient sums: fixed-order two-level sums).
 *   bcnn_hip_depthwise_stats_size      floats the statistics scratch of this layer shape needs; 0 = this shape's
 *                                      forward kernel emits no statistics
 *   bcnn_hip_depthwise_forward_stats   bcnn_hip_depthwise_forward that also leaves per-channel partial sums
 *                                      (sum, sum of squares of y) in stats_d; returns the number of partials per
 *                                      channel, 0 when none were written
 *   bcnn_hip_batchnorm_forward_stats   bcnn_hip_batchnorm_forward (TRAIN mode) that takes those partials instead of
 *                                      reading x_d for its statistics (splits == 0: plain bcnn_hip_batchnorm_forward)
 *   bcnn_hip_depthwise_bn_fusable      non-zero when bcnn_hip_depthwise_backward_bn handles this layer
 *   bcnn_hip_batchnorm_backward_sums   first half of bcnn_hip_batchnorm_backward: dbias / dscales accumulate, dmean /
 *                                      dvar are written; dy_d is NOT rewritten (x_d: the batch-norm input)
 *   bcnn_hip_batchnorm_backward_apply  second half of bcnn_hip_batchnorm_backward alone (dmean / dvar from the first):
 *                                      dy_d rewritten in place and copied to dx_d -- what the executor runs when a
 *                                      caller asks for a gradient tensor the fused pass did not have to write
 *   bcnn_hip_depthwise_backward_bn     bcnn_hip_depthwise_backward on g = BNbackward(dz_d) (second half of
 *                                      bcnn_hip_batchnorm_backward, bcnn_batchnorm_layer.c:292-296, applied on the fly);
 *                                      neither dz_d nor the depthwise output gradient is written
 * ------------------------------------------------------------------------------------------- */
size_t bcnn_hip_depthwise_stats_size(int n, int c, int h, int w, int k, int stride, int pad);
int bcnn_hip_depthwise_forward_stats(const float *x_d, const float *w_d, const float *bias_d, float *y_d, int n, int c,
                                     int h, int w, int k, int stride, int pad, int act, float *stats_d,
                                     size_t stats_floats);
void bcnn_hip_batchnorm_forward_stats(const float *x_d, float *y_d, float *run_mean_d, float *run_var_d,
                                      const float *scales_d, const float *bias_d, float *saved_mean_d,
                                      float *saved_var_d, float *x_norm_d, float *workspace_d, int n, int c, int hw,
                                      int mode, int act, const float *stats_d, int splits);
/* A stand-alone batch-norm node (no activation) whose only consumer is a 1x1 / stride 1 / one-group convolution with its own
 * fused batch-norm, TRAIN mode (MobileNet: [depthwise] -> [batchnorm] -> [conv 1x1 + BN]): z = a y + b per channel, so
 * W z = (W diag(a)) y + W b and the batch-norm's apply sweep (bcnn_batchnorm_layer.c:226-241) folds into the GEMM.
 *   bcnn_hip_batchnorm_forward_stats_only  the batch-norm node's part: saved and running statistics, nothing written
 *   bcnn_hip_conv_bnfold_fusable           1: the convolution's kernels take the folded form for this shape
 *   bcnn_hip_conv_set_input_bnfold         announces, to the NEXT bcnn_hip_conv_forward* / bcnn_hip_conv_backward* call of
 *                                          this thread (which consumes it), that its x_d is the batch-norm's INPUT y and
 *                                          the batch-norm is (mean, var, scales, bias): forward packs W diag(a) and shifts
 *                                          only the running mean of the convolution's own batch-norm by W b (the stored
 *                                          pre-normalisation values are W diag(a) y: every later use subtracts the batch
 *                                          mean); backward returns (dy y^T) diag(a) as the weight gradient. */
void bcnn_hip_batchnorm_forward_stats_only(const float *x_d, float *run_mean_d, float *run_var_d, const float *scales_d,
                                           const float *bias_d, float *saved_mean_d, float *saved_var_d, int n, int c,
                                           int hw, const float *stats_d, int splits);
int bcnn_hip_conv_bnfold_fusable(int n, int c, int h, int w, int f);
void bcnn_hip_conv_set_input_bnfold(const float *mean_d, const float *var_d, const float *scales_d, const float *bias_d);
int bcnn_hip_depthwise_bn_fusable(int n, int c, int h, int w, int k, int stride, int pad, int act);
void bcnn_hip_batchnorm_backward_sums(const float *dy_d, const float *scales_d, float *dscales_d, float *dbias_d,
                                      const float *saved_mean_d, const float *saved_var_d, float *dmean_d,
                                      float *dvar_d, const float *x_d, int n, int c, int hw);
void bcnn_hip_batchnorm_backward_apply(float *dy_d, float *dx_d, const float *scales_d, const float *saved_mean_d,
                                       const float *saved_var_d, const float *dmean_d, const float *dvar_d,
                                       const float *x_d, int n, int c, int hw);
void bcnn_hip_depthwise_backward_bn(const float *x_d, const float *w_d, const float *y_d, const float *dz_d,
                                    float *dx_d, float *dw_d, float *dbias_d, int n, int c, int h, int w, int k,
                                    int stride, int pad, int act, int overwrite, const float *bn_mean_d,
                                    const float *bn_var_d, const float *bn_scales_d, const float *bn_dmean_d,
                                    const float *bn_dvar_d);

/* A depthwise layer whose INPUT is the output of a convolution node with batch-norm (+ cheap activation) that has no other
 * consumer (MobileNet: every 1x1 convolution feeds the next block's depthwise layer). For an executor that runs whole
 * passes: the convolution node stops after its batch statistics (bcnn_hip_conv_forward_stats_only) and these three read
 * its PRE-NORMALISATION output x_raw_d, applying act(batch-norm(.)) with the in_* vectors (saved mean / variance, scales,
 * bias of that node) to every element on its way in -- the values the producer's apply sweep would have written
 * (bcnn_batchnorm_layer.c:226-241), so results are those of bcnn_hip_depthwise_forward_stats / _backward / _backward_bn.
 * Ask bcnn_hip_depthwise_bnin_fusable first. */
int bcnn_hip_depthwise_bnin_fusable(int n, int c, int h, int w, int k, int stride, int pad, int act, int in_act);
int bcnn_hip_depthwise_forward_bnin(const float *x_raw_d, const float *w_d, const float *bias_d, float *y_d, int n, int c,
                                    int h, int w, int k, int stride, int pad, int act, float *stats_d, size_t stats_floats,
                                    const float *in_mean_d, const float *in_var_d, const float *in_scale_d,
                                    const float *in_bias_d, int in_act);
void bcnn_hip_depthwise_backward_bnin(const float *x_raw_d, const float *w_d, const float *y_d, float *dy_d, float *dx_d,
                                      float *dw_d, float *dbias_d, int n, int c, int h, int w, int k, int stride, int pad,
                                      int act, int overwrite, const float *in_mean_d, const float *in_var_d,
                                      const float *in_scale_d, const float *in_bias_d, int in_act);
void bcnn_hip_depthwise_backward_bn_bnin(const float *x_raw_d, const float *w_d, const float *y_d, const float *dz_d,
                                         float *dx_d, float *dw_d, float *dbias_d, int n, int c, int h, int w, int k,
                                         int stride, int pad, int act, int overwrite, const float *bn_mean_d,
                                         const float *bn_var_d, const float *bn_scales_d, const float *bn_dmean_d,
                                         const float *bn_dvar_d, const float *in_mean_d, const float *in_var_d,
                                         const float *in_scale_d, const float *in_bias_d, int in_act);

/* ---------------------------------------------------------------------------------------------
 * SGD step on a parameter arena.  Replaces bcnn_sgd_update_gpu (bcnn_learner.c:86-104); semantics
 * of bcnn_sgd_update_cpu (:67-83) fused into one pass per buffer:
 *   b -= (lr/batch)*db; db *= momentum; dw += (decay*batch)*w; w -= (lr/batch)*dw; dw *= momentum.
 * Either pair may be NULL.
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_sgd_update(float *w_d, float *b_d, float *dw_d, float *db_d, size_t w_size,
                         size_t b_size, int batch_size, float learning_rate, float momentum,
                         float decay);

/* Adam step.  Replaces bcnn_adam_update_gpu (bcnn_learner.c:133-163) with the semantics of
 * bcnn_adam_update_cpu (:106-131): biases take the momentum-SGD step; weights
 *   dw += (decay*batch)*w; m = (1-b1)*dw + b1*m; v = (1-b2)*dw^2 + b2*v;
 *   w -= (lr/batch)*mu * m/(sqrt(v)+1e-7); dw = 0,  mu = sqrt(1-b2^(iter+1))/(1-b1^(iter+1)).
 * `iter` is what the reference passes: learner->seen (samples, not steps). adam_m_d / adam_v_d are
 * w_size floats each, zero before the first step, owned by the caller. */
void bcnn_hip_adam_update(float *w_d, float *b_d, float *dw_d, float *db_d, float *adam_m_d,
                          float *adam_v_d, size_t w_size, size_t b_size, int batch_size, int iter,
                          float beta1, float beta2, float learning_rate, float momentum, float decay);

/* The same step for MANY buffers in one launch (a net's update loop, bcnn_net.c:316-326, issues one
 * bcnn_sgd_update per node: ~40 tiny launches for ResNet-18). `chunks_d` is a device array of
 * bcnn_hip_sgd_chunk, each at most BCNN_HIP_SGD_CHUNK elements of one buffer; `use_decay` selects the
 * weights rule (decay term) or the biases rule. The caller builds the table once per net. */
#define BCNN_HIP_SGD_CHUNK 2048
typedef struct bcnn_hip_sgd_chunk {
    float *w_d;
    float *g_d;
    unsigned int count;
    unsigned int use_decay;
} bcnn_hip_sgd_chunk;
void bcnn_hip_sgd_update_chunks(const bcnn_hip_sgd_chunk *chunks_d, int num_chunks, int batch_size,
                                float learning_rate, float momentum, float decay);

/* Zero fill of MANY buffers in one launch: the executor resets every destination gradient before the node runs
 * (bcnn_net.c:361-375), ~40 small hipMemset-sized launches per ResNet-18 step; nothing reads a gradient during the
 * forward pass, so the fills that are not provably dead are issued together at its start. `chunks_d`: device array,
 * each entry at most BCNN_HIP_FILL_CHUNK floats of one buffer. */
#define BCNN_HIP_FILL_CHUNK 16384
typedef struct bcnn_hip_fill_chunk {
    float *p_d;
    unsigned int count;
    unsigned int reserved;
} bcnn_hip_fill_chunk;
void bcnn_hip_zero_chunks(const bcnn_hip_fill_chunk *chunks_d, int num_chunks);

/* ---------------------------------------------------------------------------------------------
 * "Next" rows: the nodes either side of the hot path that a ResNet-style
 * graph needs, kept on the device so that a training step has no host round trip.
 *   axpy_strided : bcnn_axpy_strided (bcnn_mat.c:159-177), the residual add of the eltwise node for
 *                  operands of different spatial size / depth (bcnn_eltwise_layer.c:119-150);
 *                  y[n][k][j*sy][i*sy] += a * x[n][k][j*sx][i*sx] over min_dim = {c,h,w}.
 *   add_rowvec   : y[r][:] += v[:] for r < rows (bias add of the full-connected node,
 *                  bcnn_fc_layer.c:170-172; plain axpy, no 0/1 quirk).
 *   softmax      : bcnn_forward_softmax_layer_cpu (bcnn_softmax_layer.c:88-155), log-sum-exp form,
 *                  over c for every (n, spatial position).
 *   eltwise_*    : the same-shape path of the eltwise node in one pass each way
 *                  (bcnn_eltwise_layer.c:112-121 copy + axpy + activation; :136-147 activation backward +
 *                  two axpys): y = act(a + b) where b only covers the first b_count elements (the
 *                  reference's stride-1 path adds the second operand to IMAGE 0 only -- quirk 5);
 *                  backward: g = dy * act'(y) stored over dy, da += g, db[i] += g[i] for i < b_count.
 *                  da_d / db_d may be NULL. overwrite_a != 0: da is assigned 0 + g instead of da += g
 *                  (same contract as bcnn_hip_maxpool_backward's overwrite).
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_eltwise_forward(const float *a_d, const float *b_d, float *y_d, size_t n, size_t b_count, int act);
void bcnn_hip_eltwise_backward(const float *y_d, float *dy_d, float *da_d, float *db_d, size_t n, size_t b_count,
                               int act, int overwrite_a);
/* The cost node's scalar metric computed on the device (bcnn_compute_error, bcnn_cost_layer.c:142-244, which
 * reads whole tensors back to the host): metric = bcnn_loss_metric value (0 error rate, 1 logloss, 2 SSE, 3 MSE,
 * 4 CRPS, 5 dice); pred_d / label_d / grad_d are [batch][per]; the result is written to out_d[0]. */
void bcnn_hip_cost_metric(int metric, const float *pred_d, const float *label_d, const float *grad_d, int batch,
                          int per, float *out_d);
void bcnn_hip_axpy_strided(int num_batches, float a, const float *x_d, float *y_d, int stride_y,
                           int stride_x, int x_c, int x_h, int x_w, int y_c, int y_h, int y_w, int min_c,
                           int min_h, int min_w);
void bcnn_hip_add_rowvec(float *y_d, const float *v_d, int rows, int cols);
void bcnn_hip_softmax_forward(const float *x_d, float *y_d, int n, int c, int hw);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel exchange (RCCL over xGMI).  No reference counterpart (the reference is single-device); process
 * model = the reference's one device per process (bcnn_cuda_set_device once in main, src/cli/bcnn_cl.c:281-285,
 * src/bcnn_utils.c:201): N processes, each after bcnn_hip_set_device(local rank), form one communicator.
 *   comm_init     : rank 0 creates the RCCL unique id and publishes it at `id_path` (a file all ranks can see;
 *                   atomic rename), the others wait for it (<= 120 s), then ncclCommInitRank on the current device;
 *                   rank 0 unlinks the file once that collective call has returned. A record left behind by a job
 *                   that died in between is told from this job's by BCNN_HIP_JOB_NONCE (environment, any string,
 *                   the same on every rank of a job and different per job: records with another nonce are ignored)
 *                   and by its age (older than BCNN_HIP_ID_MAX_AGE_S, default 600 s, when the wait starts = stale).
 *                   world == 1 may pass NULL. RCCL is dlopen'ed here, not linked. One communicator per process,
 *                   reference-counted: comm_retain adds a holder, comm_destroy drops one and tears the
 *                   communicator down with the last (two nets of one process may train over it).
 *   rendezvous_publish / _fetch : the file rendezvous by itself (what comm_init uses for the ncclUniqueId): publish
 *                   stores up to 256 bytes + world size + nonce + time at `path` (0 = ok, -1 = I/O error); fetch
 *                   polls for a record of this job (0 = fetched, 1 = timed out, 2 = published for another world size).
 *   allreduce_sum : in-place sum of n floats across the ranks, queued on the communicator's own stream AFTER the
 *                   work queued so far on the calling thread's stream (event ordering, no host block). Every rank
 *                   must issue the same sequence of calls.
 *   broadcast     : rank `root`'s n floats replace every other rank's, same ordering rules (a data-parallel job
 *                   starts from rank 0's parameters whatever each rank's own initialisation drew).
 *   comm_join     : the calling thread's stream waits for every collective queued so far (no host block).
 * Any RCCL / HIP failure prints and exit()s like every other device error.
 * ------------------------------------------------------------------------------------------- */
void bcnn_hip_comm_init(int rank, int world, const char *id_path);
void bcnn_hip_comm_retain(void);
void bcnn_hip_comm_destroy(void);
int bcnn_hip_rendezvous_publish(const char *path, const void *blob, size_t n, int world);
int bcnn_hip_rendezvous_fetch(const char *path, void *blob, size_t n, int world, int timeout_ms);
int bcnn_hip_comm_world(void);   /* 0 while no communicator exists */
int bcnn_hip_comm_rank(void);
void bcnn_hip_allreduce_sum(float *buf_d, size_t n);
void bcnn_hip_broadcast(float *buf_d, size_t n, int root);
void bcnn_hip_comm_join(void);

#ifdef __cplusplus
}
#endif
#endif /* BCNN_HIP_H */
