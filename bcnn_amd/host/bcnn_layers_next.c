/*
 * bcnn_layers_next.c -- the nodes either side of the hot path that ResNet / LeNet graphs need
 * (SURVEY.md section 8f): eltwise add, full-connected, softmax, cost. Device-resident so a training
 * step has no host round trip except the scalar loss/metric read-back the reference also does.
 *
 * Reference behaviour: bcnn_eltwise_layer.c:35-152, bcnn_fc_layer.c:38-226, bcnn_softmax_layer.c:36-166,
 * bcnn_cost_layer.c:36-284.
 */
#include <float.h>
#include <math.h>
#include <string.h>

#include "bcnn_internal.h"
#include "../../include/bcnn_hip.h"

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static bcnn_status new_output(bcnn_net *net, bcnn_node *node, int n, int c, int h, int w, const char *dst_id) {
    bcnn_tensor t = {0};
    bcnn_tensor_set_shape(&t, n, c, h, w, 1);
    BCNN_CHECK_STATUS(bcnn_tensor_allocate(&t, net->mode));
    t.name = (char *)malloc(strlen(dst_id) + 1);
    strcpy(t.name, dst_id);
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, t));
    return bcnn_node_add_output(net, node, net->num_tensors - 1);
}

static bcnn_hip_context *hctx(bcnn_net *net) { return (bcnn_hip_context *)net->hip_ctx; }

/* ================================================================================================
 * eltwise add (+ fused activation)
 * Two reference quirks are kept: the operand found LATER in the tensor list
 * becomes src[0] (the scan runs newest-first and pushes matches as it meets them), and when both
 * operands have the same spatial size the second one is added to the first `min_c * H * W` elements
 * only, i.e. to image 0 of the batch.
 * ============================================================================================== */
bcnn_status bcnn_add_eltwise_layer(bcnn_net *net, bcnn_activation activation, const char *src_id1,
                                   const char *src_id2, const char *dst_id) {
    bcnn_node node = {0};
    int found1 = 0, found2 = 0;
    for (int i = net->num_tensors - 1; i >= 0 && !(found1 && found2); --i) {
        const char *nm = net->tensors[i].name;
        if (!nm) continue;
        if (strcmp(nm, src_id1) == 0) { bcnn_node_add_input(net, &node, i); found1 = 1; }
        if (strcmp(nm, src_id2) == 0) { bcnn_node_add_input(net, &node, i); found2 = 1; }
    }
    BCNN_CHECK_AND_LOG(net->log_ctx, found1, BCNN_INVALID_PARAMETER, "Eltwise layer: invalid input node name %s\n", src_id1);
    BCNN_CHECK_AND_LOG(net->log_ctx, found2, BCNN_INVALID_PARAMETER, "Eltwise layer: invalid input node name %s\n", src_id2);
    const bcnn_tensor a = net->tensors[node.src[0]], b = net->tensors[node.src[1]];
    const int st0 = a.w / b.w, st1 = b.w / a.w;
    BCNN_CHECK_AND_LOG(net->log_ctx, st0 == a.h / b.h && st1 == b.h / a.h, BCNN_INVALID_PARAMETER,
                       "Eltwise layer: inconsistent spatial size between tensor %s and tensor %s\n", src_id1, src_id2);
    node.type = BCNN_LAYER_ELTWISE;
    node.param_size = sizeof(bcnn_eltwise_param);
    bcnn_eltwise_param *param = (bcnn_eltwise_param *)calloc(1, node.param_size);
    node.param = param;
    param->activation = activation;
    param->conv_node = -1;
    param->min_dim[0] = imin(a.c, b.c); param->min_dim[1] = imin(a.h, b.h); param->min_dim[2] = imin(a.w, b.w);
    param->stride[0] = imax(1, st0); param->stride[1] = imax(1, st1);
    node.forward = bcnn_forward_eltwise_layer;
    node.backward = bcnn_backward_eltwise_layer;
    BCNN_CHECK_STATUS(new_output(net, &node, a.n, a.c, a.h, a.w, dst_id));
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[EltWiseAdd] %-8s , %-8s -> %-8s (%4d x%4d x%4d)\n", a.name, b.name, dst_id, a.w, a.h, a.c);
    return BCNN_SUCCESS;
}

void bcnn_forward_eltwise_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_eltwise_param *p = (bcnn_eltwise_param *)node->param;
    if (p->done_forward) { /* the convolution node before this one wrote the output in this pass (bcnn_link_conv_eltwise) */
        p->done_forward = 0;
        if (hctx(net)->in_pass == 1) return;
    }
    bcnn_tensor *a = &net->tensors[node->src[0]], *b = &net->tensors[node->src[1]], *y = &net->tensors[node->dst[0]];
    const size_t sz = (size_t)bcnn_tensor_size(y);
    if (p->stride[0] == 1 && p->stride[1] == 1) { /* one fused pass; the second operand reaches image 0 only (quirk 5) */
        bcnn_hip_eltwise_forward(a->data_gpu, b->data_gpu, y->data_gpu, sz, (size_t)p->min_dim[0] * y->h * y->w,
                                 (int)p->activation);
        return;
    }
    bcnn_hip_copy_f32(sz, a->data_gpu, y->data_gpu);
    {
        bcnn_hip_axpy_strided(a->n, 1.0f, b->data_gpu, y->data_gpu, p->stride[0], p->stride[1], b->c, b->h, b->w, y->c,
                              y->h, y->w, p->min_dim[0], p->min_dim[1], p->min_dim[2]);
    }
    bcnn_hip_activation_forward(y->data_gpu, sz, (int)p->activation, NULL, y->w * y->h, y->c);
}

void bcnn_backward_eltwise_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_eltwise_param *p = (bcnn_eltwise_param *)node->param;
    if (hctx(net)->in_pass == 2 && p->conv_node >= 0) { /* the convolution node that runs next in this pass takes it */
        p->deferred = 1;
        return;
    }
    bcnn_tensor *a = &net->tensors[node->src[0]], *b = &net->tensors[node->src[1]], *y = &net->tensors[node->dst[0]];
    const size_t sz = (size_t)bcnn_tensor_size(y);
    if (p->stride[0] == 1 && p->stride[1] == 1) {
        bcnn_hip_eltwise_backward(y->data_gpu, y->grad_data_gpu, a->grad_data_gpu, b->grad_data_gpu, sz,
                                  (size_t)p->min_dim[0] * y->h * y->w, (int)p->activation,
                                  bcnn_grad_sole_writer(net, node->src[0]));
        return;
    }
    bcnn_hip_activation_backward(y->data_gpu, y->grad_data_gpu, sz, (int)p->activation, NULL, NULL, y->w * y->h, y->c);
    if (a->grad_data_gpu) bcnn_hip_axpy(sz, 1.0f, y->grad_data_gpu, a->grad_data_gpu);
    if (!b->grad_data_gpu) return;
    {
        bcnn_hip_axpy_strided(a->n, 1.0f, y->grad_data_gpu, b->grad_data_gpu, p->stride[1], p->stride[0], y->c, y->h,
                              y->w, b->c, b->h, b->w, p->min_dim[0], p->min_dim[1], p->min_dim[2]);
    }
}

/* ================================================================================================
 * full-connected (+ fused activation): y[B x P] = x[B x S] * W[P x S]^T + b
 * ============================================================================================== */
bcnn_status bcnn_add_fullc_layer(bcnn_net *net, int output_size, bcnn_filler_type init, bcnn_activation activation,
                                 int quantize, const char *src_id, const char *dst_id) {
    (void)quantize;
    bcnn_node node = {0};
    if (net->num_nodes > 0) {
        const int idx = bcnn_net_find_tensor(net, src_id);
        BCNN_CHECK_AND_LOG(net->log_ctx, idx >= 0, BCNN_INVALID_PARAMETER,
                           "Full-connected layer: invalid input node name %s\n", src_id);
        bcnn_node_add_input(net, &node, idx);
    } else {
        bcnn_node_add_input(net, &node, 0);
    }
    const bcnn_tensor s = net->tensors[node.src[0]];
    char name[256];
    snprintf(name, sizeof(name), "%s_w", src_id);
    bcnn_tensor weights = {0};
    bcnn_tensor_create(&weights, output_size, s.c, s.h, s.w, 1, name, net->mode);
    bcnn_tensor_filler wf = {.range = bcnn_tensor_size3d(&s), .type = init};
    bcnn_tensor_fill(&weights, wf);
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, weights));
    BCNN_CHECK_STATUS(bcnn_node_add_input(net, &node, net->num_tensors - 1));
    bcnn_net_register_param(net, net->num_tensors - 1);
    snprintf(name, sizeof(name), "%s_b", src_id);
    bcnn_tensor biases = {0};
    bcnn_tensor_create(&biases, 1, 1, 1, output_size, 1, name, net->mode);
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, biases));
    BCNN_CHECK_STATUS(bcnn_node_add_input(net, &node, net->num_tensors - 1));
    bcnn_net_register_param(net, net->num_tensors - 1);
    BCNN_CHECK_STATUS(new_output(net, &node, s.n, output_size, 1, 1, dst_id));
    node.type = BCNN_LAYER_FULL_CONNECTED;
    node.param_size = sizeof(bcnn_fullc_param);
    bcnn_fullc_param *param = (bcnn_fullc_param *)calloc(1, node.param_size);
    node.param = param;
    param->activation = activation;
    node.forward = bcnn_forward_fullc_layer;
    node.backward = bcnn_backward_fullc_layer;
    node.update = bcnn_update_fullc_layer;
    node.release_param = bcnn_release_param_fullc_layer;
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[Dense] %-8s (%4d x%4d x%4d) -> %-8s (%d)\n", src_id, s.w, s.h, s.c, dst_id, output_size);
    return BCNN_SUCCESS;
}

void bcnn_forward_fullc_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_fullc_param *p = (bcnn_fullc_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *w = &net->tensors[node->src[1]];
    bcnn_tensor *b = &net->tensors[node->src[2]], *y = &net->tensors[node->dst[0]];
    const int B = y->n, S = bcnn_tensor_size3d(x), P = y->c;
    bcnn_hip_gemm(0, 1, B, P, S, 1.0f, x->data_gpu, S, w->data_gpu, S, 0.0f, y->data_gpu, P);
    bcnn_hip_add_rowvec(y->data_gpu, b->data_gpu, B, P);
    bcnn_hip_activation_forward(y->data_gpu, (size_t)B * P, (int)p->activation, NULL, 1, P);
}

void bcnn_backward_fullc_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_fullc_param *p = (bcnn_fullc_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *w = &net->tensors[node->src[1]];
    bcnn_tensor *b = &net->tensors[node->src[2]], *y = &net->tensors[node->dst[0]];
    const int B = y->n, S = bcnn_tensor_size3d(x), P = y->c;
    bcnn_hip_activation_backward(y->data_gpu, y->grad_data_gpu, (size_t)B * P, (int)p->activation, NULL, NULL, 1, P);
    bcnn_hip_grad_bias(b->grad_data_gpu, y->grad_data_gpu, B, P, 1);                       /* db += sum_b dy */
    bcnn_hip_gemm(1, 0, P, S, B, 1.0f, y->grad_data_gpu, P, x->data_gpu, S, 1.0f, w->grad_data_gpu, S); /* dW += dy^T x */
    if (x->grad_data_gpu)
        bcnn_hip_gemm(0, 0, B, S, P, 1.0f, y->grad_data_gpu, P, w->data_gpu, S, 1.0f, x->grad_data_gpu, S); /* dx += dy W */
}

void bcnn_update_fullc_layer(bcnn_net *net, bcnn_node *node) { /* reference bcnn_fc_layer.c:303-348 */
    bcnn_fullc_param *p = (bcnn_fullc_param *)node->param;
    bcnn_node_optim_step(net, &net->tensors[node->src[1]], &net->tensors[node->src[2]], &p->adam_m_gpu, &p->adam_v_gpu);
}

void bcnn_release_param_fullc_layer(bcnn_node *node) {
    bcnn_fullc_param *p = (bcnn_fullc_param *)node->param;
    bcnn_hip_free(p->adam_m_gpu);
    bcnn_hip_free(p->adam_v_gpu);
}

/* ================================================================================================
 * softmax: forward log-sum-exp over channels; backward passes the gradient through (`+=`)
 * ============================================================================================== */
bcnn_status bcnn_add_softmax_layer(bcnn_net *net, const char *src_id, const char *dst_id) {
    bcnn_node node = {0};
    if (net->num_nodes > 0) {
        const int idx = bcnn_net_find_tensor(net, src_id);
        BCNN_CHECK_AND_LOG(net->log_ctx, idx >= 0, BCNN_INVALID_PARAMETER, "Softmax layer: invalid input node name %s\n", src_id);
        bcnn_node_add_input(net, &node, idx);
    } else {
        bcnn_node_add_input(net, &node, 0);
    }
    const bcnn_tensor s = net->tensors[node.src[0]];
    BCNN_CHECK_STATUS(new_output(net, &node, s.n, s.c, s.h, s.w, dst_id));
    node.type = BCNN_LAYER_SOFTMAX;
    node.forward = bcnn_forward_softmax_layer;
    node.backward = bcnn_backward_softmax_layer;
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[Softmax] %-8s -> %-8s (%4d x%4d x%4d)\n", src_id, dst_id, s.w, s.h, s.c);
    return BCNN_SUCCESS;
}

void bcnn_forward_softmax_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    bcnn_hip_softmax_forward(x->data_gpu, y->data_gpu, x->n, x->c, x->h * x->w);
}

void bcnn_backward_softmax_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    if (x->grad_data_gpu) bcnn_hip_axpy((size_t)bcnn_tensor_size(x), 1.0f, y->grad_data_gpu, x->grad_data_gpu);
}

/* ================================================================================================
 * cost: grad = prediction - label on the device; the scalar metric is computed on the host from
 * the copies the reference CUDA build also reads back (bcnn_cost_layer.c:147-158).
 * ============================================================================================== */
bcnn_status bcnn_add_cost_layer(bcnn_net *net, bcnn_loss loss, bcnn_loss_metric loss_metric, float scale,
                                const char *src_id, const char *label_id, const char *dst_id) {
    (void)label_id; /* the label is always tensor 1 */
    bcnn_node node = {0};
    BCNN_CHECK_AND_LOG(net->log_ctx, net->num_nodes >= 1, BCNN_INVALID_PARAMETER,
                       "Cost layer can't be the first layer of the network\n");
    BCNN_CHECK_AND_LOG(net->log_ctx, loss == BCNN_LOSS_EUCLIDEAN, BCNN_INVALID_PARAMETER,
                       "Cost layer: only the euclidean loss is built in the MI355X hot-path build\n");
    const int idx = bcnn_net_find_tensor(net, src_id);
    BCNN_CHECK_AND_LOG(net->log_ctx, idx >= 0, BCNN_INVALID_PARAMETER, "Cost layer: invalid input node name %s\n", src_id);
    bcnn_node_add_input(net, &node, idx);
    node.type = BCNN_LAYER_COST;
    node.param_size = sizeof(bcnn_cost_param);
    bcnn_cost_param *param = (bcnn_cost_param *)calloc(1, node.param_size);
    node.param = param;
    param->scale = scale; param->loss = loss; param->loss_metric = loss_metric;
    node.forward = bcnn_forward_cost_layer;
    node.backward = bcnn_backward_cost_layer;
    const bcnn_tensor s = net->tensors[idx];
    bcnn_tensor_set_shape(&net->tensors[1], s.n, s.c, s.h, s.w, 0);
    BCNN_CHECK_STATUS(bcnn_tensor_allocate(&net->tensors[1], net->mode));
    bcnn_node_add_input(net, &node, 1);
    BCNN_CHECK_STATUS(new_output(net, &node, s.n, s.c, s.h, s.w, dst_id));
    return bcnn_net_add_node(net, node);
}

static float host_metric(const bcnn_cost_param *p, const bcnn_tensor *pred, const bcnn_tensor *label,
                         const bcnn_tensor *dst) {
    const int per = pred->w * pred->h * pred->c, B = pred->n, sz = per * B;
    double acc = 0.0;
    switch (p->loss_metric) {
        case BCNN_METRIC_ERROR_RATE:
            for (int i = 0; i < B; ++i) {
                float pm = FLT_MIN;
                int best = 0;
                for (int j = 0; j < per; ++j)
                    if (pred->data[i * per + j] > pm) { pm = pred->data[i * per + j]; best = j; }
                if (label->data[i * per + best] == 0) acc += 1.0;
            }
            return (float)acc;
        case BCNN_METRIC_LOGLOSS:
            for (int i = 0; i < sz; ++i)
                if (label->data[i] > 0.0f) {
                    float q = pred->data[i];
                    q = q < 1e-8f ? 1e-8f : (q > 1.0f - 1e-8f ? 1.0f - 1e-8f : q);
                    acc += -log(q);
                }
            return (float)acc;
        case BCNN_METRIC_MSE:
        case BCNN_METRIC_SSE:
        case BCNN_METRIC_CRPS:
            for (int i = 0; i < sz; ++i) acc += (double)dst->grad_data[i] * dst->grad_data[i];
            return (float)(p->loss_metric == BCNN_METRIC_MSE ? acc / per : acc);
        case BCNN_METRIC_DICE:
            for (int i = 0; i < B; ++i) {
                int n = 0, d = 0;
                for (int j = 0; j < per; ++j) {
                    n += (int)(label->data[i * per + j] * (pred->data[i * per + j] > 0.5f));
                    d += (int)(label->data[i * per + j] + (pred->data[i * per + j] > 0.5f));
                }
                acc += (2.0f * n + 1.0f) / (d + 1.0f);
            }
            return (float)acc;
    }
    return 0.f;
}

void bcnn_forward_cost_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_cost_param *p = (bcnn_cost_param *)node->param;
    bcnn_tensor *pred = &net->tensors[node->src[0]], *label = &net->tensors[1], *dst = &net->tensors[node->dst[0]];
    if (!label->data) return;
    const size_t sz = (size_t)bcnn_tensor_size(pred);
    if (dst->grad_data_gpu) {
        bcnn_hip_copy_f32(sz, pred->data_gpu, dst->grad_data_gpu);
        bcnn_hip_axpy(sz, -1.0f, label->data_gpu, dst->grad_data_gpu);
    }
    if (net->mode == BCNN_MODE_PREDICT) return;
    /* The scalar metric: the reference's CUDA build reads prediction, label and gradient back and loops on the
     * host (bcnn_cost_layer.c:142-244); here one small kernel computes it from the device copies (the label's
     * device copy is what the loader uploaded, bcnn_data.c:413-425) and only the 4-byte result is read back, so
     * dst->data[0] is valid when bcnn_forward returns, as it is there. BCNN_HOST_COST_METRIC=1 keeps the host loop. */
    static int host_loop = -1;
    if (host_loop < 0) host_loop = BCNN_EXP_ENV("BCNN_HOST_COST_METRIC") != NULL;
    if (host_loop || !label->data_gpu || !dst->grad_data_gpu) {
        if (dst->grad_data_gpu) bcnn_hip_memcpy_d2h(dst->grad_data, dst->grad_data_gpu, sz * sizeof(float));
        bcnn_hip_memcpy_d2h(pred->data, pred->data_gpu, sz * sizeof(float));
        dst->data[0] = host_metric(p, pred, label, dst);
        bcnn_hip_memcpy_h2d(dst->data_gpu, dst->data, sizeof(float)); /* keep the mirror coherent */
        return;
    }
    bcnn_hip_cost_metric((int)p->loss_metric, pred->data_gpu, label->data_gpu, dst->grad_data_gpu, pred->n,
                         pred->w * pred->h * pred->c, dst->data_gpu);
    bcnn_hip_memcpy_d2h(dst->data, dst->data_gpu, sizeof(float));
}

void bcnn_backward_cost_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_cost_param *p = (bcnn_cost_param *)node->param;
    bcnn_tensor *pred = &net->tensors[node->src[0]], *dst = &net->tensors[node->dst[0]];
    if (pred->grad_data_gpu)
        bcnn_hip_axpy((size_t)bcnn_tensor_size(pred), p->scale, dst->grad_data_gpu, pred->grad_data_gpu);
}
