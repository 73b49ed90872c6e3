/*
 * bcnn/bcnn.h -- public C API of the MI355X build of bcnn's conv/GEMM hot path.
 *
 * Source-compatible with the reference's inc/bcnn/bcnn.h (jnbraun/bcnn): same enumerators (values
 * matter for INI files and saved models), same `struct bcnn_tensor` members, same 52 entry points with
 * identical signatures, so `bcnn-cl` and the examples compile against it unchanged.
 * Differences: the device mirror members are guarded by BCNN_USE_HIP (reference: BCNN_USE_CUDA,
 * bcnn.h:251-254) and a few data-parallel helpers are appended at the end (new, no reference
 * counterpart). Entry points outside the hot path return
 * BCNN_INVALID_PARAMETER with a log line; they are listed in INTEGRATION.md.
 */
#ifndef BCNN_H
#define BCNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__) && defined(BCNN_BUILD_SHARED)
#define BCNN_API __attribute__((visibility("default")))
#else
#define BCNN_API
#endif

#define BCNN_VERSION_MAJOR 0
#define BCNN_VERSION_MINOR 2
#define BCNN_VERSION_PATCH 0

typedef struct bcnn_net bcnn_net;
typedef struct bcnn_tensor bcnn_tensor;
typedef struct bcnn_output_detection bcnn_output_detection;

/* ---- enumerations (values identical to the reference, bcnn.h:90-242) ---- */
typedef enum {
    BCNN_SUCCESS, BCNN_INVALID_PARAMETER, BCNN_INVALID_DATA, BCNN_INVALID_MODEL, BCNN_FAILED_ALLOC,
    BCNN_INTERNAL_ERROR, BCNN_CUDA_FAILED_ALLOC, BCNN_UNKNOWN_ERROR
} bcnn_status;

typedef enum { BCNN_MODE_PREDICT, BCNN_MODE_TRAIN, BCNN_MODE_VALID } bcnn_mode;

typedef enum {
    BCNN_LOAD_MNIST, BCNN_LOAD_CIFAR10, BCNN_LOAD_CLASSIFICATION_LIST, BCNN_LOAD_REGRESSION_LIST,
    BCNN_LOAD_DETECTION_LIST, BCNN_NUM_LOADERS
} bcnn_loader_type;

typedef enum {
    BCNN_LR_DECAY_CONSTANT, BCNN_LR_DECAY_STEP, BCNN_LR_DECAY_INV, BCNN_LR_DECAY_EXP, BCNN_LR_DECAY_POLY,
    BCNN_LR_DECAY_SIGMOID
} bcnn_lr_decay;

typedef enum {
    BCNN_LAYER_CONV2D, BCNN_LAYER_TRANSPOSE_CONV2D, BCNN_LAYER_DEPTHWISE_CONV2D, BCNN_LAYER_ACTIVATION,
    BCNN_LAYER_FULL_CONNECTED, BCNN_LAYER_MAXPOOL, BCNN_LAYER_AVGPOOL, BCNN_LAYER_SOFTMAX, BCNN_LAYER_DROPOUT,
    BCNN_LAYER_BATCHNORM, BCNN_LAYER_LRN, BCNN_LAYER_CONCAT, BCNN_LAYER_ELTWISE, BCNN_LAYER_UPSAMPLE,
    BCNN_LAYER_YOLOV3, BCNN_LAYER_RESHAPE, BCNN_LAYER_COST
} bcnn_layer_type;

typedef enum {
    BCNN_ACT_NONE, BCNN_ACT_TANH, BCNN_ACT_RELU, BCNN_ACT_RAMP, BCNN_ACT_SOFTPLUS,
    BCNN_ACT_LRELU, /* negative slope 0.1 in the code (the reference's comment says 0.01) */
    BCNN_ACT_ABS, BCNN_ACT_CLAMP, BCNN_ACT_PRELU, BCNN_ACT_LOGISTIC
} bcnn_activation;

typedef enum { BCNN_LOSS_EUCLIDEAN, BCNN_LOSS_LIFTED_STRUCT } bcnn_loss;

typedef enum {
    BCNN_METRIC_ERROR_RATE, BCNN_METRIC_LOGLOSS, BCNN_METRIC_SSE, BCNN_METRIC_MSE, BCNN_METRIC_CRPS,
    BCNN_METRIC_DICE
} bcnn_loss_metric;

typedef enum { BCNN_PADDING_SAME, BCNN_PADDING_VALID, BCNN_PADDING_CAFFE } bcnn_padding;

typedef enum { BCNN_OPTIM_SGD, BCNN_OPTIM_ADAM } bcnn_optimizer;

typedef enum { BCNN_LOG_INFO = 0, BCNN_LOG_WARNING = 1, BCNN_LOG_ERROR = 2, BCNN_LOG_SILENT = 3 } bcnn_log_level;

typedef enum bcnn_filler_type { BCNN_FILLER_FIXED, BCNN_FILLER_XAVIER, BCNN_FILLER_MSRA } bcnn_filler_type;

#define BCNN_DETECTION_MAX_BOXES 50

typedef void (*bcnn_log_callback)(const char *fmt, ...);

/* Dense NCHW fp32 tensor. Host buffers are 32-byte aligned and zero-initialised; with BCNN_USE_HIP
 * every tensor also owns device mirrors that the kernels work on (sync points: INTEGRATION.md). */
struct bcnn_tensor {
    int n;            /* batch */
    int c;            /* channels */
    int h;            /* height */
    int w;            /* width */
    int has_grad;     /* carries a gradient buffer in TRAIN/VALID nets */
    char *name;
    float *data;
    float *grad_data;
#ifdef BCNN_USE_HIP
    float *data_gpu;
    float *grad_data_gpu;
#endif
};

struct bcnn_output_detection {
    int num_classes;
    float x, y, w, h;
    float *prob;
    float *mask;
    float objectness;
};

/* ---- net life cycle ---- */
BCNN_API bcnn_status bcnn_init_net(bcnn_net **net, bcnn_mode mode);
BCNN_API void bcnn_end_net(bcnn_net **net);
BCNN_API void bcnn_set_log_context(bcnn_net *net, bcnn_log_callback fct, bcnn_log_level level);
BCNN_API bcnn_status bcnn_set_num_threads(bcnn_net *net, int num_threads, const int *cpu_ids);
BCNN_API int bcnn_get_num_threads(bcnn_net *net);
BCNN_API void bcnn_set_input_shape(bcnn_net *net, int width, int height, int channels, int batch_size);
BCNN_API bcnn_status bcnn_add_input(bcnn_net *net, int width, int height, int channels, const char *name);
BCNN_API int bcnn_get_batch_size(bcnn_net *net);
BCNN_API bcnn_status bcnn_resize_net(bcnn_net *net, int w, int h, int c, int need_realloc);
BCNN_API bcnn_status bcnn_compile_net(bcnn_net *net);
BCNN_API bcnn_status bcnn_load_weights(bcnn_net *net, const char *model_path);
BCNN_API bcnn_status bcnn_load_net(bcnn_net *net, const char *config_path, const char *model_path);
BCNN_API bcnn_status bcnn_save_weights(bcnn_net *net, const char *filename);

/* ---- data ---- */
BCNN_API bcnn_status bcnn_set_data_loader(bcnn_net *net, bcnn_loader_type type, const char *train_path_data,
                                          const char *train_path_extra, const char *test_path_data,
                                          const char *test_path_extra);
BCNN_API void bcnn_augment_data_with_shift(bcnn_net *net, int width_shift_range, int height_shift_range);
BCNN_API void bcnn_augment_data_with_scale(bcnn_net *net, float min_scale, float max_scale);
BCNN_API void bcnn_augment_data_with_rotation(bcnn_net *net, float rotation_range);
BCNN_API void bcnn_augment_data_with_flip(bcnn_net *net, int horizontal_flip, int vertical_flip);
BCNN_API void bcnn_augment_data_with_color_adjustment(bcnn_net *net, int min_brightness, int max_brightness,
                                                      float min_constrast, float max_contrast);
BCNN_API void bcnn_augment_data_with_blobs(bcnn_net *net, int max_blobs);
BCNN_API void bcnn_augment_data_with_distortion(bcnn_net *net, float distortion);
BCNN_API bcnn_status bcnn_fill_tensor_with_image(bcnn_net *net, const uint8_t *src, int w, int h, int c,
                                                 float norm_coeff, int swap_to_bgr, float mean_r, float mean_g,
                                                 float mean_b, int tensor_index, int batch_index);

/* ---- training set-up ---- */
BCNN_API bcnn_status bcnn_set_mode(bcnn_net *net, bcnn_mode mode);
BCNN_API void bcnn_set_adam_optimizer(bcnn_net *net, float learning_rate, float beta1, float beta2);
BCNN_API void bcnn_set_sgd_optimizer(bcnn_net *net, float learning_rate, float momentum);
BCNN_API void bcnn_set_learning_rate_policy(bcnn_net *net, bcnn_lr_decay decay_type, float gamma, float scale,
                                            float power, int max_batches, int step);
BCNN_API void bcnn_set_weight_regularizer(bcnn_net *net, float weight_decay);

/* ---- execution ---- */
BCNN_API void bcnn_forward(bcnn_net *net);
BCNN_API void bcnn_backward(bcnn_net *net);
BCNN_API void bcnn_update(bcnn_net *net);
BCNN_API float bcnn_train_on_batch(bcnn_net *net);
BCNN_API float bcnn_predict_on_batch(bcnn_net *net, bcnn_tensor **out);
BCNN_API bcnn_output_detection *bcnn_yolo_get_detections(bcnn_net *net, int batch, int width, int height, int netw,
                                                         int neth, float thresh, int relative, int *num_dets);
BCNN_API int bcnn_get_tensor_index_by_name(bcnn_net *net, const char *name);
BCNN_API bcnn_tensor *bcnn_get_tensor_by_index(bcnn_net *net, int index);
BCNN_API bcnn_tensor *bcnn_get_tensor_by_name(bcnn_net *net, const char *name);

/* ---- layer builders ---- */
BCNN_API bcnn_status bcnn_add_convolutional_layer(bcnn_net *net, int num_filters, int size, int stride, int pad,
                                                  int num_groups, int batch_norm, bcnn_filler_type init,
                                                  bcnn_activation activation, int quantize, const char *src_id,
                                                  const char *dst_id);
BCNN_API bcnn_status bcnn_add_deconvolutional_layer(bcnn_net *net, int num_filters, int size, int stride, int pad,
                                                    bcnn_filler_type init, bcnn_activation activation,
                                                    const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_depthwise_conv_layer(bcnn_net *net, int size, int stride, int pad, int batch_norm,
                                                   bcnn_filler_type init, bcnn_activation activation,
                                                   const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_batchnorm_layer(bcnn_net *net, const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_lrn_layer(bcnn_net *net, int local_size, float alpha, float beta, float k,
                                        const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_fullc_layer(bcnn_net *net, int output_size, bcnn_filler_type init,
                                          bcnn_activation activation, int quantize, const char *src_id,
                                          const char *dst_id);
BCNN_API bcnn_status bcnn_add_activation_layer(bcnn_net *net, bcnn_activation type, const char *id);
BCNN_API bcnn_status bcnn_add_softmax_layer(bcnn_net *net, const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_maxpool_layer(bcnn_net *net, int size, int stride, bcnn_padding padding,
                                            const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_avgpool_layer(bcnn_net *net, const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_concat_layer(bcnn_net *net, int num_src, char *const *src_ids, const char *dst_id);
BCNN_API bcnn_status bcnn_add_eltwise_layer(bcnn_net *net, bcnn_activation activation, const char *src_id1,
                                            const char *src_id2, const char *dst_id);
BCNN_API bcnn_status bcnn_add_dropout_layer(bcnn_net *net, float rate, const char *id);
BCNN_API bcnn_status bcnn_add_upsample_layer(bcnn_net *net, int size, const char *src_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_cost_layer(bcnn_net *net, bcnn_loss loss, bcnn_loss_metric loss_metric, float scale,
                                         const char *src_id, const char *label_id, const char *dst_id);
BCNN_API bcnn_status bcnn_add_yolo_layer(bcnn_net *net, int num_boxes_per_cell, int num_classes, int coords,
                                         int total, int *mask, float *anchors, const char *src_id,
                                         const char *dst_id);

/* ---------------------------------------------------------------------------------------------
 * Additions of the MI355X build (no reference counterpart).
 *  - bcnn_upload_tensor / bcnn_download_tensor: push a user-written host buffer to its device mirror
 *    and back (the reference has no public call for this; its CUDA build only syncs inside the
 *    data loader, bcnn_data.c:413-425).
 *  - data parallel: one process per GPU. After bcnn_compile_net every weight/bias gradient lives in
 *    ONE contiguous device arena; a launcher all-reduces (sum) that arena over RCCL between
 *    bcnn_backward and bcnn_update and tells the net the world size so the SGD step divides by the
 *    global batch and rescales the momentum carry ("data parallel").
 * ------------------------------------------------------------------------------------------- */
BCNN_API bcnn_status bcnn_upload_tensor(bcnn_net *net, int tensor_index, int with_grad);
BCNN_API bcnn_status bcnn_download_tensor(bcnn_net *net, int tensor_index, int with_grad);
BCNN_API bcnn_status bcnn_set_data_parallel(bcnn_net *net, int rank, int world_size);
/* The same, with the collective INSIDE the library (RCCL over xGMI, include/bcnn_hip.h): for a plain C program run as
 * one process per GPU (the reference's process model, src/cli/bcnn_cl.c:281-285). Call after bcnn_hip_set_device /
 * before training; `id_path` names a file every rank can reach, unique per job, through which rank 0 hands out the
 * RCCL id (world_size == 1 may pass NULL). From then on bcnn_backward all-reduces (sum) the weight-gradient arena
 * itself -- in ~8 MB buckets as the owning nodes finish, on the communicator's stream, overlapped with the rest of
 * backward -- and bcnn_update is ordered behind the last bucket; bcnn_train_on_batch needs no other change. The
 * communicator is destroyed by bcnn_end_net. A launcher that runs the collective itself (bench.py through
 * torch.distributed) keeps using bcnn_set_data_parallel + bcnn_get_gradient_arena. */
BCNN_API bcnn_status bcnn_set_data_parallel_comm(bcnn_net *net, int rank, int world_size, const char *id_path);
/* bcnn_backward queues the weight-gradient kernels of a pass on a second stream of the library, next to the sweeps and
 * data gradients of the layers in front, and joins it at the end of the pass (include/bcnn_hip.h:
 * bcnn_hip_conv_side_stream_mode). enable = 0 keeps everything on the caller's stream (a profiler
 * that wants every kernel alone, bench.py's `roofline.alone` leg); the results are the same bit for bit either way. */
BCNN_API void bcnn_set_weight_gradient_stream(bcnn_net *net, int enable);
BCNN_API float *bcnn_get_gradient_arena(bcnn_net *net, size_t *num_floats);  /* device pointer */
BCNN_API float *bcnn_get_parameter_arena(bcnn_net *net, size_t *num_floats); /* device pointer */
/* Overlap of the gradient all-reduce with backward. Parameters sit in the arena in node order and backward
 * visits the nodes in reverse, so the finished gradients always form a growing TAIL of the arena: `fn` is
 * called from inside bcnn_backward (same thread, after the owning node's work has been queued on the
 * stream) with the newly completed range [first_float, first_float + num_floats). The caller may queue a
 * collective on that range that depends on the stream's work so far. NULL removes the callback. */
typedef void (*bcnn_gradient_ready_fn)(size_t first_float, size_t num_floats, void *user);
BCNN_API void bcnn_set_gradient_ready_callback(bcnn_net *net, bcnn_gradient_ready_fn fn, void *user);
BCNN_API void bcnn_synchronize(bcnn_net *net);
/* borrowed pointer to tensor `index` WITHOUT the device->host refresh bcnn_get_tensor_by_index performs */
BCNN_API bcnn_tensor *bcnn_peek_tensor(bcnn_net *net, int index);
BCNN_API int bcnn_get_num_nodes(bcnn_net *net);
BCNN_API int bcnn_get_node_tensor(bcnn_net *net, int node, int is_dst, int slot); /* -1 if out of range */
/* layer-private device state needed by parity tests: which = 0 maxpool indexes (int*), 1 saved_mean,
 * 2 saved_variance, 3 d(saved_mean), 4 d(saved_variance), 5 the pre-normalisation values a batch-norm's backward works
 * from (the reference's param->workspace: a fused-BN convolution's raw output / a batch-norm node's kept input);
 * returns a DEVICE pointer or NULL */
BCNN_API void *bcnn_get_node_state(bcnn_net *net, int node, int which);
/* Run ONE node's forward / backward worker on whatever its tensors currently hold (no executor bookkeeping:
 * no zero fill of the dst gradients, no dead-fill elision -- a sole-writer gradient is accumulated like in the
 * reference). Used by the teacher-forced parity walk, which feeds every node the REFERENCE's inputs. */
BCNN_API bcnn_status bcnn_forward_node(bcnn_net *net, int node);
BCNN_API bcnn_status bcnn_backward_node(bcnn_net *net, int node);

#ifdef __cplusplus
}
#endif
#endif /* BCNN_H */
