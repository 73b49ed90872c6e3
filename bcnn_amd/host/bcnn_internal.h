/*
 * bcnn_internal.h -- host-side (C99) runtime structures of the MI355X build.
 *
 * Struct tags and member names follow the reference's internal headers, because unchanged consumers
 * reach into them (src/cli/bcnn_cl.c:106-199 reads net->learner->max_batches, net->batch_size,
 * net->log_ctx, net->tensors[i], net->nodes[i].{dst,type,param}, net->data_loader->type):
 *   struct bcnn_net      reference src/bcnn_net.h:47-67
 *   struct bcnn_node     reference src/bcnn_node.h:36-48   (the operator plug-in signature)
 *   bcnn_learner         reference src/bcnn_learner.h:29-44
 *   bcnn_loader, bcnn_data_augmenter   reference src/bcnn_data.h:34-103
 *   bcnn_log_context + BCNN_CHECK_* macros   reference src/bcnn_utils.h:48-100
 * The headers bcnn_net.h, bcnn_node.h, bcnn_tensor.h, bcnn_utils.h, bcnn_learner.h, bcnn_data.h and the
 * per-layer headers in this directory are thin includes of this file so that `#include "bcnn_net.h"`
 * etc. keep working. Binary layout is free (consumers are recompiled).
 */
#ifndef BCNN_INTERNAL_H
#define BCNN_INTERNAL_H

#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>

#include <bcnn/bcnn.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- logging / status macros ---------------------------------------------------------------- */
typedef struct {
    bcnn_log_callback fct;
    bcnn_log_level lvl;
} bcnn_log_context;

void bcnn_log(bcnn_log_context ctx, bcnn_log_level level, const char *fmt, ...);

#define BCNN_CHECK(exp, err) do { if (!(exp)) return (err); } while (0)
#define BCNN_CHECK_AND_LOG(ctx, exp, err, fmt, ...) \
    do { if (!(exp)) { bcnn_log((ctx), BCNN_LOG_ERROR, (fmt), ##__VA_ARGS__); return (err); } } while (0)
#define BCNN_CHECK_STATUS(s) do { bcnn_status ret_ = (s); if (ret_ != BCNN_SUCCESS) return ret_; } while (0)
#define BCNN_ERROR(ctx, err, fmt, ...) do { bcnn_log((ctx), BCNN_LOG_ERROR, (fmt), ##__VA_ARGS__); return (err); } while (0)
#define BCNN_INFO(ctx, fmt, ...) bcnn_log((ctx), BCNN_LOG_INFO, (fmt), ##__VA_ARGS__)
#define BCNN_WARNING(ctx, fmt, ...) bcnn_log((ctx), BCNN_LOG_WARNING, (fmt), ##__VA_ARGS__)

/* debugging switches of the host runtime exist only in an experiment build (-DBCNN_HIP_EXPERIMENT) */
#ifdef BCNN_HIP_EXPERIMENT
#define BCNN_EXP_ENV(name) getenv(name)
#else
#define BCNN_EXP_ENV(name) ((const char *)NULL)
#endif

typedef struct {
    int state;
    float r;
} bcnn_gauss_gen;
float bcnn_rng_gaussian(bcnn_gauss_gen *g);

/* ---- node: the operator plug-in --------------------------------------------------------------- */
struct bcnn_node {
    int num_src;
    int num_dst;
    bcnn_layer_type type;
    size_t param_size;
    int *src; /* indices into net->tensors (stable; pointers into the array are not) */
    int *dst;
    void *param;
    void (*forward)(struct bcnn_net *net, struct bcnn_node *node);
    void (*backward)(struct bcnn_net *net, struct bcnn_node *node);
    void (*update)(struct bcnn_net *net, struct bcnn_node *node);
    void (*release_param)(struct bcnn_node *node);
};
typedef struct bcnn_node bcnn_node;

/* ---- learner ------------------------------------------------------------------------------------ */
typedef struct {
    int step;
    int seen;
    int max_batches;
    float momentum;
    float decay;
    float base_learning_rate;
    float learning_rate;
    float gamma;
    float scale;
    float power;
    float beta1;
    float beta2;
    bcnn_optimizer optimizer;
    bcnn_lr_decay decay_type;
} bcnn_learner;

/* ---- data loader / augmenter (host only: bcnn_data.c; reference src/bcnn_data.h:34-131) ---------- */
typedef struct {
    int input_width;
    int input_height;
    int input_depth;
    bool has_extra_data;
    bcnn_loader_type type;
    uint8_t *input_uchar;
    uint8_t *input_net;
    FILE *f_train;
    FILE *f_train_extra;
    FILE *f_test;
    FILE *f_test_extra;
    FILE *f_current;
    FILE *f_current_extra;
} bcnn_loader;

typedef struct {
    int range_shift_x, range_shift_y, random_fliph, min_brightness, max_brightness, swap_to_bgr, no_input_norm,
        max_random_spots;
    float min_scale, max_scale, rotation_range, min_contrast, max_contrast, max_distortion, mean_r, mean_g, mean_b;
    int use_precomputed, brightness, apply_fliph, shift_x, shift_y;
    float rotation, scale, contrast, distortion, distortion_kx, distortion_ky;
} bcnn_data_augmenter;

struct bcnn_net;
bcnn_status bcnn_loader_next(struct bcnn_net *net);
void bcnn_convert_img_to_float(const uint8_t *src, int w, int h, int c, float norm_coeff, int swap_to_bgr, float mean_r,
                               float mean_g, float mean_b, float *dst);
bcnn_status bcnn_apply_data_augmentation(unsigned char *img, int width, int height, int depth, bcnn_data_augmenter *param,
                                         unsigned char *buffer);
bcnn_status bcnn_open_dataset(bcnn_loader *iter, struct bcnn_net *net, const char *train_path, const char *train_path_extra,
                              const char *test_path, const char *test_path_extra, bool has_extra);
bcnn_status bcnn_switch_data_handles(struct bcnn_net *net, bcnn_loader *iter);
void bcnn_fill_input_tensor(struct bcnn_net *net, bcnn_loader *iter, char *path_img, int idx);
void bcnn_destroy_data_loader(struct bcnn_net *net);

/* ---- device context of a net (reference analogue: bcnn_cuda_context, src/bcnn_net.h:37-42) ------ */
typedef struct bcnn_hip_context {
    size_t workspace_size;  /* floats; shared conv backward scratch, sized at compile time */
    float *workspace_gpu;
    /* parameter / gradient arenas built by bcnn_compile_net (one all-reduce per step) */
    float *param_arena_gpu;
    float *grad_arena_gpu;
    size_t arena_size;  /* floats */
    int *param_ids;     /* tensor indices of trainable parameters, in creation order */
    int num_params;
    int arena_members; /* how many of param_ids[] already live inside the arenas (bcnn_compile_net re-packs when it grows) */
    int dp_rank;
    int dp_world;
    int compiled;
    /* update loop folded into one launch: table of (buffer, gradient, count, rule) chunks, built by the
     * first bcnn_update from what the nodes' update() workers ask for */
    /* gradient-ready callback (bcnn_set_gradient_ready_callback): node_grad_first[i] = arena offset of the first
     * parameter owned by node i, or (size_t)-1 */
    void (*grad_ready_fn)(size_t, size_t, void *);
    void *grad_ready_user;
    size_t *node_grad_first;
    /* per tensor: 1 = the zero-fill of its gradient before forward is dead (see mark_dead_grad_fills) */
    unsigned char *grad_fill_dead;
    int grad_fill_count; /* entries of grad_fill_dead[] (tensors that existed at compile time) */
    struct bcnn_hip_sgd_chunk *sgd_chunks_host;
    void *sgd_chunks_gpu;
    int num_sgd_chunks, cap_sgd_chunks;
    int sgd_collecting;
    /* the gradient zero fills of a TRAIN-mode forward pass gathered into one launch (built by the first forward after
     * a compile; dropped with the SGD table when the graph changes) */
    void *fill_chunks_gpu;
    int num_fill_chunks;
    /* RCCL inside the library (bcnn_set_data_parallel_comm): finished tail ranges of the gradient arena are gathered
     * into buckets of comm_bucket floats and all-reduced on the communicator's stream while backward continues */
    int comm_active;
    size_t comm_bucket, comm_lo, comm_hi;
    /* 1 inside bcnn_forward / 2 inside bcnn_backward: node workers may rely on their neighbours having run in this pass
     * (bcnn_forward_node / bcnn_backward_node run one worker alone: 0) */
    int in_pass;
    int no_side_stream; /* bcnn_set_weight_gradient_stream(net, 0) */
} bcnn_hip_context;

/* ---- net ------------------------------------------------------------------------------------------ */
struct bcnn_net {
    int batch_size;
    int num_nodes;
    int num_tensors;
    int num_inputs;
    int *inputs;
    bcnn_mode mode;
    bcnn_log_context log_ctx;
    bcnn_node *nodes;
    bcnn_tensor *tensors;
    bcnn_learner *learner;
    bcnn_loader *data_loader;
    bcnn_data_augmenter *data_aug;
    void *gemm_ctx; /* unused here (the reference's CPU gemm scratch) */
#ifdef BCNN_USE_HIP
    void *hip_ctx; /* bcnn_hip_context* */
#endif
    int num_threads;
};

/* ---- net / node / tensor helpers (reference: bcnn_net.h:69-76, bcnn_node.h:50-51, bcnn_tensor.h:37-63) */
bcnn_status bcnn_net_add_node(bcnn_net *net, bcnn_node node);
bcnn_status bcnn_net_add_tensor(bcnn_net *net, bcnn_tensor tensor);
bcnn_status bcnn_node_add_input(bcnn_net *net, bcnn_node *node, int index);
bcnn_status bcnn_node_add_output(bcnn_net *net, bcnn_node *node, int index);

typedef struct tensor_filler {
    int range;
    float value;
    bcnn_filler_type type;
} bcnn_tensor_filler;

void bcnn_tensor_create(bcnn_tensor *t, int n, int c, int h, int w, int has_grad, const char *name, int net_state);
void bcnn_tensor_fill(bcnn_tensor *t, bcnn_tensor_filler filler);
void bcnn_tensor_destroy(bcnn_tensor *t);
void bcnn_tensor_set_shape(bcnn_tensor *t, int n, int c, int h, int w, int has_grad);
bcnn_status bcnn_tensor_allocate_buffer(bcnn_tensor *t, int net_state, size_t size);
bcnn_status bcnn_tensor_allocate(bcnn_tensor *t, int net_state);
void bcnn_tensor_free(bcnn_tensor *t);
int bcnn_tensor_size(const bcnn_tensor *t);
int bcnn_tensor_size3d(const bcnn_tensor *t);
int bcnn_tensor_size2d(const bcnn_tensor *t);

/* marks tensor `index` as a trainable parameter (member of the gradient arena) */
void bcnn_net_register_param(bcnn_net *net, int index);
/* finds the most recently created tensor called `name`; -1 if absent (reference builders scan newest-first) */
int bcnn_net_find_tensor(bcnn_net *net, const char *name);

bcnn_status bcnn_loader_next(bcnn_net *net);
void bcnn_convert_img_to_float(const uint8_t *src, int w, int h, int c, float norm_coeff, int swap_to_bgr,
                               float mean_r, float mean_g, float mean_b, float *dst);
void bcnn_draw_color_box(unsigned char *img, int w_img, int h_img, float cx_box, float cy_box, float w_box,
                         float h_box, unsigned char color[3]);

static inline const char *bcnn_act2str(bcnn_activation a) {
    switch (a) {
        case BCNN_ACT_TANH: return "Tanh";
        case BCNN_ACT_RELU: return "ReLU";
        case BCNN_ACT_RAMP: return "Ramp";
        case BCNN_ACT_SOFTPLUS: return "Softplus";
        case BCNN_ACT_LRELU: return "Leaky-ReLU";
        case BCNN_ACT_ABS: return "AbsVal";
        case BCNN_ACT_CLAMP: return "Clamp";
        case BCNN_ACT_PRELU: return "PReLU";
        case BCNN_ACT_LOGISTIC: return "Logistic";
        default: return "None";
    }
}

/* ---- layer parameter blocks (member names follow src/layers/<layer>.h of the reference) ---------- */
typedef struct bcnn_conv_param {
    int num, size, stride, pad, num_groups, batch_norm, post_func;
    size_t workspace_size;
    bcnn_activation activation;
    bcnn_tensor saved_mean;     /* batch statistics (data) and their gradients (grad_data) */
    bcnn_tensor saved_variance;
    float *conv_workspace;      /* unused on the device path (no materialised im2col) */
    float *workspace;
    float *x_norm;
    float *adam_m, *adam_v;
#ifdef BCNN_USE_HIP
    float *conv_workspace_gpu;  /* = net hip_ctx workspace (dW split-K partials) */
    float *bn_workspace_gpu;    /* pre-normalisation conv output, kept for backward */
    float *x_norm_gpu;          /* NULL: recomputed in backward */
    float *adam_m_gpu, *adam_v_gpu; /* weight moments, allocated by the first Adam step */
    /* set by bcnn_compile_net when the node that follows is an eltwise node adding something to this node's (batch-norm,
     * no activation) output (bcnn_link_conv_eltwise): inside a whole pass the eltwise work rides on this node's
     * batch-norm sweeps */
    int elt_node;      /* index of that eltwise node, -1: none */
    int pool_node;     /* index of the max-pooling node that is this node's only consumer and normalises this node's
                        * pre-normalisation output on the fly inside a forward pass (bcnn_link_conv_maxpool), -1: none */
    int bnsums_node;   /* the stand-alone batch-norm node in front of this (1x1) node whose backward sums this node's
                        * data-gradient kernel emits inside a backward pass (bcnn_link_batchnorm_conv), -1: none */
    int dw_node;       /* index of the depthwise node that is this node's only consumer and normalises this node's
                        * pre-normalisation output while staging it (bcnn_link_conv_depthwise), -1: none */
    int apply_skipped; /* this node left its batch-norm apply sweep to its consumer in the running forward pass */
    int fold_bn;       /* the stand-alone batch-norm node in front of this (1x1) node that leaves its apply sweep to this node's
                        * packed weights inside a forward pass (bcnn_link_batchnorm_conv), -1: none */
    int folded;        /* the running pass's forward took that route: the backward reads the batch-norm's INPUT */
    int pool_bwd_pending; /* pool_node >= 0, inside a backward pass: that node left its backward to this node's (one kernel
                           * does the pooling backward and this node's batch-norm backward: bcnn_hip_maxpool_bn_backward) */
    float *insums_gpu;   /* dw_node >= 0: partial backward sums of this node's batch-norm, left by that depthwise node's */
    size_t insums_floats; /* backward kernel (bcnn_hip_depthwise_backward_bnin_sums) in the running backward pass */
    int insums_splits;   /* > 0: insums_gpu holds them (partials per channel) */
    int data_pending;  /* the last forward pass did not write this node's output tensor (nobody inside a pass reads it):
                        * bcnn_materialize_data produces it from bn_workspace_gpu on demand */
#endif
} bcnn_conv_param;

typedef struct bcnn_depthwise_conv_param {
    int size, stride, pad, batch_norm;
    bcnn_activation activation;
    float *adam_m, *adam_v;
#ifdef BCNN_USE_HIP
    float *adam_m_gpu, *adam_v_gpu;
    /* set by bcnn_compile_net when this node's output feeds a stand-alone batch-norm node (bcnn_link_depthwise_batchnorm):
     * inside a whole forward / backward pass the two workers share work */
    int bn_node;         /* index of that batch-norm node, -1: none */
    int bn_fused_bwd;    /* the batch-norm node leaves its apply sweep to this node's backward kernel */
    float *stats_gpu;    /* per-channel statistics partials of the last forward (TRAIN) */
    size_t stats_floats;
    int stats_splits;    /* > 0: stats_gpu holds the statistics of the output written by the last forward of this pass */
    int conv_node;       /* the convolution node whose batch-norm this node applies to its input on the fly, -1: none */
    int raw_input;       /* the last forward pass read that node's pre-normalisation output: so must its backward */
    int grads_pending;   /* the last backward pass did not write the gradient of this node's output nor rewrite the
                          * batch-norm node's output gradient (nothing inside a pass reads them): they are produced on
                          * demand by bcnn_materialize_gradients, as the reference's two workers leave them */
#endif
} bcnn_depthwise_conv_param;

typedef struct bcnn_batchnorm_param {
    bcnn_tensor saved_mean;
    bcnn_tensor saved_variance;
    float *workspace;
    float *x_norm;
#ifdef BCNN_USE_HIP
    float *workspace_gpu;
    float *x_norm_gpu;
    int dw_node;         /* the depthwise node that produces this node's input (see bcnn_depthwise_conv_param), -1: none */
    int dw_fused_bwd;    /* that node applies this node's backward to the gradient it consumes */
    int sums_conv;       /* the 1x1 convolution node that is this node's only consumer: inside a backward pass its
                          * data-gradient kernel emits the partial sums this node's backward starts with, -1: none */
    float *bsums_gpu;    /* those partials (bcnn_hip_conv_bnsums_size floats) */
    size_t bsums_floats;
    int bsums_splits;    /* > 0: bsums_gpu holds the sums of the gradient written in the running backward pass */
    int input_kept;      /* the input tensor is not this node's output and has no other consumer: it IS the copy of the
                          * input the reference keeps in `workspace` (bcnn_batchnorm_layer.c:208) */
    int fold_conv;       /* the 1x1 convolution node that is this node's only consumer and folds this node's affine map into
                          * its weights inside a forward pass (bcnn_hip_conv_set_input_bnfold), -1: none */
    int apply_skipped;   /* the running forward pass left this node's output tensor unwritten for that node ... */
    int data_pending;    /* ... and nobody has asked for it since (bcnn_materialize_data produces it on demand) */
#endif
} bcnn_batchnorm_param;

typedef struct bcnn_maxpool_param {
    int size, stride;
    bcnn_padding padding;
    int *indexes;
#ifdef BCNN_USE_HIP
    int *indexes_gpu;
    int conv_node;  /* the convolution node whose batch-norm this node applies on the fly (see bcnn_conv_param), -1: none */
    float *raw_at_max_gpu; /* conv_node >= 0: per pooled element the pre-normalisation value that won its window, kept by
                            * the forward pass for bcnn_hip_maxpool_bn_backward (the output tensor's shape) */
    int raw_fwd;    /* the running pass's forward took that route (raw_at_max_gpu is current) */
#endif
} bcnn_maxpool_param;

typedef struct bcnn_activation_param {
    bcnn_activation activation;
} bcnn_activation_param;

typedef struct bcnn_eltwise_param {
    bcnn_activation activation;
    int stride[2];
    int min_dim[3];
#ifdef BCNN_USE_HIP
    int conv_node;     /* the convolution node that does this node's work inside a pass (see bcnn_conv_param), -1: none */
    int done_forward;  /* that node already wrote this node's output in the running forward pass */
    int deferred;      /* this node's backward was left to that node in the running backward pass */
    int grad_pending;  /* the last backward pass did not rewrite this node's output gradient (dy *= act'(y),
                        * bcnn_eltwise_layer.c:124-127): bcnn_materialize_gradients does on demand */
#endif
} bcnn_eltwise_param;

typedef struct bcnn_fullc_param {
    bcnn_activation activation;
    float *adam_m, *adam_v;
#ifdef BCNN_USE_HIP
    float *adam_m_gpu, *adam_v_gpu;
#endif
} bcnn_fullc_param;

typedef struct bcnn_cost_param {
    float scale;
    bcnn_loss loss;
    bcnn_loss_metric loss_metric;
} bcnn_cost_param;

/* only `classes` is read by consumers (src/cli/bcnn_cl.c:199); the YOLO head itself is out of scope */
typedef struct bcnn_yolo_param {
    int num, classes, coords, total;
    int *mask;
    float *biases;
    float *cost;
} bcnn_yolo_param;

/* hot-path node workers (installed into bcnn_node) */
void bcnn_forward_conv_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_conv_layer(bcnn_net *net, bcnn_node *node);
void bcnn_update_conv_layer(bcnn_net *net, bcnn_node *node);
void bcnn_release_param_conv_layer(bcnn_node *node);
void bcnn_forward_depthwise_conv_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_depthwise_conv_layer(bcnn_net *net, bcnn_node *node);
void bcnn_update_depthwise_conv_layer(bcnn_net *net, bcnn_node *node);
void bcnn_release_param_depthwise_conv_layer(bcnn_node *node);
void bcnn_forward_batchnorm_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_batchnorm_layer(bcnn_net *net, bcnn_node *node);
void bcnn_release_param_batchnorm_layer(bcnn_node *node);
void bcnn_forward_maxpool_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_maxpool_layer(bcnn_net *net, bcnn_node *node);
void bcnn_release_param_maxpool_layer(bcnn_node *node);
void bcnn_forward_avgpool_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_avgpool_layer(bcnn_net *net, bcnn_node *node);
void bcnn_forward_activation_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_activation_layer(bcnn_net *net, bcnn_node *node);
void bcnn_update_activation_layer(bcnn_net *net, bcnn_node *node);
void bcnn_forward_eltwise_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_eltwise_layer(bcnn_net *net, bcnn_node *node);
void bcnn_forward_fullc_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_fullc_layer(bcnn_net *net, bcnn_node *node);
void bcnn_update_fullc_layer(bcnn_net *net, bcnn_node *node);
void bcnn_release_param_fullc_layer(bcnn_node *node);
void bcnn_forward_softmax_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_softmax_layer(bcnn_net *net, bcnn_node *node);
void bcnn_forward_cost_layer(bcnn_net *net, bcnn_node *node);
void bcnn_backward_cost_layer(bcnn_net *net, bcnn_node *node);

/* SGD step on one node's parameters (bcnn_learner.c:67-104 in the reference) */
void bcnn_link_depthwise_batchnorm(bcnn_net *net); /* bcnn_layers_hot.c; called by bcnn_compile_net */
void bcnn_link_conv_eltwise(bcnn_net *net);        /* bcnn_layers_hot.c; called by bcnn_compile_net */
void bcnn_link_conv_maxpool(bcnn_net *net);
void bcnn_link_conv_depthwise(bcnn_net *net);
void bcnn_link_batchnorm_conv(bcnn_net *net);
void bcnn_materialize_data(bcnn_net *net, int tensor);      /* tensor < 0: every pending one */
void bcnn_materialize_gradients(bcnn_net *net, int tensor); /* tensor < 0: every pending one */
void bcnn_drop_pending_gradients(bcnn_net *net);
void bcnn_prepack_conv_weights(bcnn_net *net, int data_gradient); /* bcnn_layers_hot.c */
int bcnn_grad_sole_writer(bcnn_net *net, int tensor); /* 1: this gradient's zero fill was skipped, assign instead of += */
void bcnn_node_sgd_step(bcnn_net *net, bcnn_tensor *weights, bcnn_tensor *biases);
void bcnn_node_optim_step(bcnn_net *net, bcnn_tensor *weights, bcnn_tensor *biases, float **adam_m_gpu,
                          float **adam_v_gpu); /* SGD or Adam according to net->learner->optimizer */

#ifdef __cplusplus
}
#endif
#endif /* BCNN_INTERNAL_H */
