/*
 * bcnn_layers_hot.c -- builders and node workers of the hot-path operators (conv, depthwise conv,
 * batch-norm, max/avg pooling, activation) of the MI355X build. Each worker has the reference's
 * plug-in signature `void f(bcnn_net*, bcnn_node*)` and tensor-index conventions
 * (conv: src = {x, W, b [, run_mean, run_var, scales] [, prelu slopes]}, bcnn_conv_layer.c:368-387;
 *  batch-norm: src = {x, run_mean, run_var, scales, biases}, bcnn_batchnorm_layer.c:247-250)
 * and makes ONE whole-batch call into the C-ABI (include/bcnn_hip.h) per direction.
 *
 * Builder behaviour follows: bcnn_conv_layer.c:45-365, bcnn_depthwise_conv_layer.c:42-163,
 * bcnn_batchnorm_layer.c:36-145, bcnn_maxpool_layer.c:40-143, bcnn_avgpool_layer.c:33-80,
 * bcnn_activation_layer.c:36-88 (tensor names "<src>_w", "<src>_b", "<src>_run_mean", ... included,
 * because bcnn_get_tensor_by_name and the weight files rely on them).
 */
#include <math.h>
#include <string.h>

#include "bcnn_internal.h"
#include "../../include/bcnn_hip.h"

static bcnn_hip_context *hctx(bcnn_net *net) { return (bcnn_hip_context *)net->hip_ctx; }

/* Resolve the source tensor of a new node: tensor 0 for the first node, else newest tensor of that name. */
static bcnn_status attach_source(bcnn_net *net, bcnn_node *node, const char *src_id, const char *what) {
    if (net->num_nodes == 0) return bcnn_node_add_input(net, node, 0);
    const int idx = bcnn_net_find_tensor(net, src_id);
    BCNN_CHECK_AND_LOG(net->log_ctx, idx >= 0, BCNN_INVALID_PARAMETER, "%s layer: invalid input node name %s\n",
                       what, src_id);
    return bcnn_node_add_input(net, node, idx);
}

/* creates a [1,1,1,count] parameter tensor, registers it with the node (and the arena if trainable) */
static bcnn_status add_vector_param(bcnn_net *net, bcnn_node *node, int count, int has_grad, const char *src_id,
                                    const char *suffix, float fill, int trainable) {
    char name[256];
    snprintf(name, sizeof(name), "%s_%s", src_id, suffix);
    bcnn_tensor t = {0};
    bcnn_tensor_create(&t, 1, 1, 1, count, has_grad, name, net->mode);
    if (fill != 0.0f) {
        bcnn_tensor_filler f = {.value = fill, .type = BCNN_FILLER_FIXED};
        bcnn_tensor_fill(&t, f);
    }
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, t));
    BCNN_CHECK_STATUS(bcnn_node_add_input(net, node, net->num_tensors - 1));
    if (trainable) bcnn_net_register_param(net, net->num_tensors - 1);
    return BCNN_SUCCESS;
}

static bcnn_status add_output(bcnn_net *net, bcnn_node *node, int n, int c, int h, int w, const char *dst_id) {
    bcnn_tensor t = {0};
    bcnn_tensor_set_shape(&t, n, c, h, w, 1);
    BCNN_CHECK_STATUS(bcnn_tensor_allocate(&t, net->mode));
    t.name = (char *)malloc(strlen(dst_id) + 1);
    strcpy(t.name, dst_id);
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, t));
    return bcnn_node_add_output(net, node, net->num_tensors - 1);
}

/* ================================================================================================
 * convolution (+ fused batch-norm, + fused activation)
 * ============================================================================================== */
bcnn_status bcnn_add_convolutional_layer(bcnn_net *net, int num_filters, int size, int stride, int pad,
                                         int num_groups, int batch_norm, bcnn_filler_type init,
                                         bcnn_activation activation, int quantize, const char *src_id,
                                         const char *dst_id) {
    (void)quantize;
    bcnn_node node = {0};
    if (net->num_nodes == 0)
        BCNN_CHECK_AND_LOG(net->log_ctx, bcnn_tensor_size(&net->tensors[0]) > 0, BCNN_INVALID_PARAMETER,
                           "Invalid input size of the network. Hint: use 'bcnn_set_input_shape'\n");
    BCNN_CHECK_STATUS(attach_source(net, &node, src_id, "Convolution"));
    const int sn = net->tensors[node.src[0]].n, sc = net->tensors[node.src[0]].c;
    const int sh = net->tensors[node.src[0]].h, sw = net->tensors[node.src[0]].w;
    BCNN_CHECK_AND_LOG(net->log_ctx, num_groups > 0 && sc % num_groups == 0, BCNN_INVALID_PARAMETER,
                       "Number of input channels has to be a multiple of the number of groups\n");
    BCNN_CHECK_AND_LOG(net->log_ctx, num_filters % num_groups == 0, BCNN_INVALID_PARAMETER,
                       "Number of output channels has to be a multiple of the number of groups\n");
    const int cg = sc / num_groups;
    const int oh = (sh + 2 * pad - size) / stride + 1, ow = (sw + 2 * pad - size) / stride + 1;
    /* A 1x1 kernel reads its source as a raw [C/g][OH*OW] matrix whatever stride / pad say (quirk 1,
     * bcnn_conv_layer.c:445-446). With padding that makes OH*OW > H*W the reference walks past the end of the
     * last image in host memory; on the device that would be an out-of-bounds read of x and WRITE of dx.
     * Deliberate deviation: such a layer is refused here. */
    if (size == 1 && oh * ow > sh * sw) {
        free(node.src);
        BCNN_CHECK_AND_LOG(net->log_ctx, 0, BCNN_INVALID_PARAMETER,
                           "Convolution layer %s: a 1x1 kernel with pad %d gives %d x %d outputs from %d x %d inputs; "
                           "the raw-view addressing of 1x1 kernels would run past the source tensor\n",
                           dst_id, pad, ow, oh, sw, sh);
    }

    char name[256];
    snprintf(name, sizeof(name), "%s_w", src_id);
    bcnn_tensor weights = {0};
    bcnn_tensor_create(&weights, num_filters, cg, size, size, 1, name, net->mode);
    bcnn_tensor_filler wf = {.range = size * size * cg, .type = init};
    bcnn_tensor_fill(&weights, wf);
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, weights));
    BCNN_CHECK_STATUS(bcnn_node_add_input(net, &node, net->num_tensors - 1));
    bcnn_net_register_param(net, net->num_tensors - 1);
    BCNN_CHECK_STATUS(add_vector_param(net, &node, num_filters, 1, src_id, "b", 0.0f, 1));

    node.type = BCNN_LAYER_CONV2D;
    node.param_size = sizeof(bcnn_conv_param);
    bcnn_conv_param *param = (bcnn_conv_param *)calloc(1, node.param_size);
    node.param = param;
    param->activation = activation;
    param->pad = pad; param->num = num_filters; param->size = size; param->stride = stride;
    param->num_groups = num_groups;
    param->elt_node = -1;
    param->pool_node = -1;
    param->dw_node = -1;
    param->bnsums_node = -1;
    param->fold_bn = -1;
    node.forward = bcnn_forward_conv_layer;
    node.backward = bcnn_backward_conv_layer;
    node.update = bcnn_update_conv_layer;
    node.release_param = bcnn_release_param_conv_layer;

    BCNN_CHECK_STATUS(add_output(net, &node, sn, num_filters, oh, ow, dst_id));
    if (batch_norm) {
        param->batch_norm = 1;
        snprintf(name, sizeof(name), "%s_sav_mean", src_id);
        bcnn_tensor_create(&param->saved_mean, 1, 1, 1, num_filters, 1, name, net->mode);
        snprintf(name, sizeof(name), "%s_sav_var", src_id);
        bcnn_tensor_create(&param->saved_variance, 1, 1, 1, num_filters, 1, name, net->mode);
        BCNN_CHECK_STATUS(add_vector_param(net, &node, num_filters, 0, src_id, "run_mean", 0.0f, 0));
        BCNN_CHECK_STATUS(add_vector_param(net, &node, num_filters, 0, src_id, "run_var", 0.0f, 0));
        /* scales accumulate a gradient but no update function ever applies it (reference
         * bcnn_conv_layer.c:810-855 updates weights and biases only) -- kept out of the arena */
        BCNN_CHECK_STATUS(add_vector_param(net, &node, num_filters, 1, src_id, "scales", 1.0f, 0));
        if (net->mode != BCNN_MODE_PREDICT)
            param->bn_workspace_gpu = bcnn_hip_malloc_f32((size_t)sn * num_filters * oh * ow);
    }
    if (activation == BCNN_ACT_PRELU)
        BCNN_CHECK_STATUS(add_vector_param(net, &node, num_filters, 0, src_id, "prelu_slopes", 0.0f, 0));
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[Conv2D%s%s] %-8s (%4d x%4d x%4d) -> %-8s (%4d x%4d x%4d) %d x %d / %d pad %d groups %d\n",
              batch_norm ? "+BN" : "", activation != BCNN_ACT_NONE ? "+act" : "", src_id, sw, sh, sc, dst_id, ow, oh,
              num_filters, size, size, stride, pad, num_groups);
    return BCNN_SUCCESS;
}

typedef struct {
    bcnn_tensor *x, *w, *b, *y, *run_mean, *run_var, *scales, *slopes;
} conv_io;

static conv_io conv_tensors(bcnn_net *net, bcnn_node *node) {
    conv_io io = {0};
    bcnn_conv_param *p = (bcnn_conv_param *)node->param;
    io.x = &net->tensors[node->src[0]];
    io.w = &net->tensors[node->src[1]];
    io.b = &net->tensors[node->src[2]];
    io.y = &net->tensors[node->dst[0]];
    if (p->batch_norm) {
        io.run_mean = &net->tensors[node->src[3]];
        io.run_var = &net->tensors[node->src[4]];
        io.scales = &net->tensors[node->src[5]];
    }
    if (p->activation == BCNN_ACT_PRELU) io.slopes = &net->tensors[node->src[3 + 3 * p->batch_norm]];
    return io;
}

/* the stand-alone batch-norm node in front of a node whose forward folded it (cp->folded), for the calls that follow */
static const float *announce_bnfold(bcnn_net *net, const bcnn_conv_param *p) {
    const bcnn_node *bn = &net->nodes[p->fold_bn];
    const bcnn_batchnorm_param *bp = (const bcnn_batchnorm_param *)bn->param;
    bcnn_hip_conv_set_input_bnfold(bp->saved_mean.data_gpu, bp->saved_variance.data_gpu, net->tensors[bn->src[3]].data_gpu,
                                   net->tensors[bn->src[4]].data_gpu);
    return net->tensors[bn->src[0]].data_gpu; /* the batch-norm's INPUT: what the convolution multiplies */
}

void bcnn_forward_conv_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_conv_param *p = (bcnn_conv_param *)node->param;
    conv_io io = conv_tensors(net, node);
    p->data_pending = 0;
    p->apply_skipped = 0;
    p->folded = 0;
    const float *xin = io.x->data_gpu;
    if (hctx(net)->in_pass == 1 && net->mode == BCNN_MODE_TRAIN && p->fold_bn >= 0 &&
        ((bcnn_batchnorm_param *)net->nodes[p->fold_bn].param)->apply_skipped) {
        /* the batch-norm node that ran just before in this pass stopped after its statistics: whichever forward entry runs
         * below reads ITS input and packs weights that carry its affine map */
        ((bcnn_batchnorm_param *)net->nodes[p->fold_bn].param)->apply_skipped = 0;
        xin = announce_bnfold(net, p);
        p->folded = 1;
    }
    if (hctx(net)->in_pass == 1 && (p->pool_node >= 0 || p->dw_node >= 0) && net->mode == BCNN_MODE_TRAIN) {
        /* the max-pooling / depthwise node that runs next in this pass normalises this node's pre-normalisation output on
         * the fly: convolution and batch statistics only; this node's own output tensor is not written */
        bcnn_hip_conv_forward_stats_only(xin, io.w->data_gpu, io.b->data_gpu, io.x->n, io.x->c, io.x->h, io.x->w,
                                         p->num, p->size, p->stride, p->pad, p->num_groups, io.run_mean->data_gpu,
                                         io.run_var->data_gpu, io.scales->data_gpu, p->saved_mean.data_gpu,
                                         p->saved_variance.data_gpu, p->bn_workspace_gpu);
        p->apply_skipped = 1;
        p->data_pending = 1;
        return;
    }
    if (hctx(net)->in_pass == 1 && p->elt_node >= 0 && net->mode == BCNN_MODE_TRAIN) {
        /* the eltwise node that runs next in this pass: its add + activation ride on this node's batch-norm apply sweep,
         * whose result goes straight to the eltwise output; this node's own output tensor is not written */
        bcnn_node *en = &net->nodes[p->elt_node];
        bcnn_eltwise_param *ep = (bcnn_eltwise_param *)en->param;
        bcnn_tensor *r = &net->tensors[en->src[1]], *out = &net->tensors[en->dst[0]];
        bcnn_hip_conv_forward_residual(xin, io.w->data_gpu, io.b->data_gpu, io.x->n, io.x->c, io.x->h, io.x->w,
                                       p->num, p->size, p->stride, p->pad, p->num_groups, io.run_mean->data_gpu,
                                       io.run_var->data_gpu, io.scales->data_gpu, p->saved_mean.data_gpu,
                                       p->saved_variance.data_gpu, p->bn_workspace_gpu, r->data_gpu,
                                       (size_t)ep->min_dim[0] * out->h * out->w, (int)ep->activation, out->data_gpu);
        ep->done_forward = 1;
        p->data_pending = 1;
        return;
    }
    bcnn_hip_conv_forward(xin, io.w->data_gpu, io.b->data_gpu, io.y->data_gpu, io.x->n, io.x->c, io.x->h,
                          io.x->w, p->num, p->size, p->stride, p->pad, p->num_groups, (int)p->activation,
                          io.slopes ? io.slopes->data_gpu : NULL, p->batch_norm,
                          io.run_mean ? io.run_mean->data_gpu : NULL, io.run_var ? io.run_var->data_gpu : NULL,
                          io.scales ? io.scales->data_gpu : NULL, p->saved_mean.data_gpu,
                          p->saved_variance.data_gpu, p->x_norm_gpu, p->bn_workspace_gpu, (int)net->mode);
}

void bcnn_backward_conv_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_conv_param *p = (bcnn_conv_param *)node->param;
    conv_io io = conv_tensors(net, node);
    const float *xin = io.x->data_gpu;
    if (hctx(net)->in_pass == 2 && p->folded && p->fold_bn >= 0) /* this pass's forward folded the batch-norm in front: */
        xin = announce_bnfold(net, p);                           /* the weight gradient is formed against ITS input */
    if (hctx(net)->in_pass == 2 && p->pool_bwd_pending) {
        /* the max-pooling node (which ran just before in this pass) left its backward to this node */
        bcnn_node *pn = &net->nodes[p->pool_node];
        bcnn_maxpool_param *mp = (bcnn_maxpool_param *)pn->param;
        const bcnn_tensor *py = &net->tensors[pn->dst[0]];
        p->pool_bwd_pending = 0;
        bcnn_hip_maxpool_bn_backward(py->grad_data_gpu, mp->indexes_gpu, mp->raw_at_max_gpu, p->bn_workspace_gpu,
                                     io.y->grad_data_gpu, io.y->n, io.y->c, io.y->h, io.y->w, py->h, py->w, mp->size,
                                     mp->stride, io.scales->data_gpu, io.scales->grad_data_gpu, io.b->data_gpu,
                                     io.b->grad_data_gpu, p->saved_mean.data_gpu, p->saved_variance.data_gpu,
                                     p->saved_mean.grad_data_gpu, p->saved_variance.grad_data_gpu, (int)p->activation);
        bcnn_hip_conv_backward_bn_done(xin, io.w->data_gpu, io.y->grad_data_gpu, io.x->grad_data_gpu,
                                       io.w->grad_data_gpu, io.x->n, io.x->c, io.x->h, io.x->w, p->num, p->size, p->stride,
                                       p->pad, p->num_groups, p->conv_workspace_gpu, hctx(net)->workspace_size);
        return;
    }
    if (hctx(net)->in_pass == 2 && p->elt_node >= 0 && ((bcnn_eltwise_param *)net->nodes[p->elt_node].param)->deferred) {
        /* the eltwise node (which ran just before in this pass) left its backward to this node's batch-norm sweeps */
        bcnn_node *en = &net->nodes[p->elt_node];
        bcnn_eltwise_param *ep = (bcnn_eltwise_param *)en->param;
        bcnn_tensor *r = &net->tensors[en->src[1]], *out = &net->tensors[en->dst[0]];
        ep->deferred = 0;
        ep->grad_pending = 1;
        bcnn_hip_conv_backward_residual(xin, io.w->data_gpu, io.b->data_gpu, io.y->grad_data_gpu, io.x->grad_data_gpu,
                                        io.w->grad_data_gpu, io.b->grad_data_gpu, io.x->n, io.x->c, io.x->h, io.x->w,
                                        p->num, p->size, p->stride, p->pad, p->num_groups, io.scales->data_gpu,
                                        io.scales->grad_data_gpu, p->saved_mean.data_gpu, p->saved_variance.data_gpu,
                                        p->saved_mean.grad_data_gpu, p->saved_variance.grad_data_gpu, p->bn_workspace_gpu,
                                        p->conv_workspace_gpu, hctx(net)->workspace_size, out->data_gpu,
                                        out->grad_data_gpu, (int)ep->activation, r->data_gpu, r->grad_data_gpu,
                                        (size_t)ep->min_dim[0] * out->h * out->w);
        return;
    }
    /* sums of this node's own batch-norm backward that the depthwise node behind it (which ran just before in this pass)
     * left while writing this node's output gradient */
    const int own_splits = hctx(net)->in_pass == 2 ? p->insums_splits : 0;
    p->insums_splits = 0;
    const int to_bn = hctx(net)->in_pass == 2 && p->bnsums_node >= 0 && io.x->grad_data_gpu;
    if (to_bn || own_splits > 0) {
        /* to_bn: the stand-alone batch-norm node that runs next in this pass gets the partial sums of the gradient this
         * node's data-gradient kernel writes, straight from that kernel's epilogue */
        bcnn_node *bn = to_bn ? &net->nodes[p->bnsums_node] : NULL;
        bcnn_batchnorm_param *bp = bn ? (bcnn_batchnorm_param *)bn->param : NULL;
        const int splits = bcnn_hip_conv_backward_presummed(
            xin, io.w->data_gpu, io.b->data_gpu, io.y->data_gpu, io.y->grad_data_gpu, io.x->grad_data_gpu,
            io.w->grad_data_gpu, io.b->grad_data_gpu, io.x->n, io.x->c, io.x->h, io.x->w, p->num, p->size, p->stride, p->pad,
            p->num_groups, (int)p->activation, io.slopes ? io.slopes->data_gpu : NULL,
            io.slopes ? io.slopes->grad_data_gpu : NULL, p->batch_norm, io.scales ? io.scales->data_gpu : NULL,
            io.scales ? io.scales->grad_data_gpu : NULL, p->saved_mean.data_gpu, p->saved_variance.data_gpu,
            p->saved_mean.grad_data_gpu, p->saved_variance.grad_data_gpu, p->x_norm_gpu, p->bn_workspace_gpu,
            p->conv_workspace_gpu, hctx(net)->workspace_size, own_splits > 0 ? p->insums_gpu : NULL, own_splits,
            bn ? net->tensors[bn->src[0]].data_gpu : NULL, bp ? bp->saved_mean.data_gpu : NULL, bp ? bp->bsums_gpu : NULL,
            bp ? bp->bsums_floats : 0);
        if (bp) bp->bsums_splits = splits;
        return;
    }
    bcnn_hip_conv_backward(xin, io.w->data_gpu, io.b->data_gpu, io.y->data_gpu, io.y->grad_data_gpu,
                           io.x->grad_data_gpu /* NULL for the net input: no dX */, io.w->grad_data_gpu,
                           io.b->grad_data_gpu, io.x->n, io.x->c, io.x->h, io.x->w, p->num, p->size, p->stride,
                           p->pad, p->num_groups, (int)p->activation, io.slopes ? io.slopes->data_gpu : NULL,
                           io.slopes ? io.slopes->grad_data_gpu : NULL, p->batch_norm,
                           io.scales ? io.scales->data_gpu : NULL, io.scales ? io.scales->grad_data_gpu : NULL,
                           p->saved_mean.data_gpu, p->saved_variance.data_gpu, p->saved_mean.grad_data_gpu,
                           p->saved_variance.grad_data_gpu, p->x_norm_gpu, p->bn_workspace_gpu,
                           p->conv_workspace_gpu, hctx(net)->workspace_size);
}

/* The filter banks of every convolution node re-arranged for its kernel in ONE launch per kind, at the start of a pass
 * (the nodes' own calls would each pack right before their kernel: ~40 launches of ~5 us per ResNet-18 step).
 * data_gradient = 0: forward forms; 1: data-gradient forms, of the nodes whose source carries a gradient. */
void bcnn_prepack_conv_weights(bcnn_net *net, int data_gradient) {
    int n = 0;
    if (BCNN_EXP_ENV("BCNN_NO_PREPACK")) return; /* A/B switch of the experiment build: every node packs for itself */
    if (net->num_nodes < 1) return;
    bcnn_hip_conv_desc *d = (bcnn_hip_conv_desc *)malloc((size_t)net->num_nodes * sizeof(*d));
    if (!d) return; /* every node then packs for itself */
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *node = &net->nodes[i];
        if (node->type != BCNN_LAYER_CONV2D) continue;
        const bcnn_conv_param *p = (const bcnn_conv_param *)node->param;
        const bcnn_tensor *x = &net->tensors[node->src[0]], *w = &net->tensors[node->src[1]];
        if (!w->data_gpu || (data_gradient && !x->grad_data_gpu)) continue;
        /* a forward pack of a node that folds the batch-norm in front is made per call (its column factors depend on this
         * batch's statistics): nothing to prepare ahead */
        if (!data_gradient && net->mode == BCNN_MODE_TRAIN && p->fold_bn >= 0) continue;
        d[n].w_d = w->data_gpu;
        d[n].n = x->n; d[n].c = x->c; d[n].h = x->h; d[n].w = x->w;
        d[n].f = p->num; d[n].k = p->size; d[n].stride = p->stride; d[n].pad = p->pad; d[n].groups = p->num_groups;
        ++n;
    }
    bcnn_hip_conv_prepack(d, n, data_gradient); /* also with nothing to pack: a pass begins (stale packs and folds are dropped) */
    free(d);
}

void bcnn_update_conv_layer(bcnn_net *net, bcnn_node *node) { /* reference bcnn_conv_layer.c:810-855 */
    bcnn_conv_param *p = (bcnn_conv_param *)node->param;
    bcnn_node_optim_step(net, &net->tensors[node->src[1]], &net->tensors[node->src[2]], &p->adam_m_gpu, &p->adam_v_gpu);
}

void bcnn_release_param_conv_layer(bcnn_node *node) {
    bcnn_conv_param *p = (bcnn_conv_param *)node->param;
    bcnn_tensor_destroy(&p->saved_mean);
    bcnn_tensor_destroy(&p->saved_variance);
    bcnn_hip_free(p->bn_workspace_gpu);
    bcnn_hip_free(p->insums_gpu);
    bcnn_hip_free(p->x_norm_gpu);
    bcnn_hip_free(p->adam_m_gpu);
    bcnn_hip_free(p->adam_v_gpu);
}

/* ================================================================================================
 * depthwise convolution
 * ============================================================================================== */
bcnn_status bcnn_add_depthwise_conv_layer(bcnn_net *net, int size, int stride, int pad, int batch_norm,
                                          bcnn_filler_type init, bcnn_activation activation, const char *src_id,
                                          const char *dst_id) {
    (void)batch_norm; /* ignored by the reference as well (bcnn_depthwise_conv_layer.c:42-163) */
    bcnn_node node = {0};
    BCNN_CHECK_STATUS(attach_source(net, &node, src_id, "Depthwise convolution"));
    const bcnn_tensor s = net->tensors[node.src[0]];
    char name[256];
    snprintf(name, sizeof(name), "%s_w", src_id);
    bcnn_tensor weights = {0};
    bcnn_tensor_create(&weights, 1, 1, 1, s.c * size * size, 1, name, net->mode); /* laid out [C][k][k] */
    bcnn_tensor_filler wf = {.range = size * size * s.c, .type = init};
    bcnn_tensor_fill(&weights, wf);
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, weights));
    BCNN_CHECK_STATUS(bcnn_node_add_input(net, &node, net->num_tensors - 1));
    bcnn_net_register_param(net, net->num_tensors - 1);
    BCNN_CHECK_STATUS(add_vector_param(net, &node, s.c, 1, src_id, "b", 0.0f, 1));
    node.type = BCNN_LAYER_DEPTHWISE_CONV2D;
    node.param_size = sizeof(bcnn_depthwise_conv_param);
    bcnn_depthwise_conv_param *param = (bcnn_depthwise_conv_param *)calloc(1, node.param_size);
    node.param = param;
    param->activation = activation; param->size = size; param->stride = stride; param->pad = pad;
    param->bn_node = -1;
    param->conv_node = -1;
    node.forward = bcnn_forward_depthwise_conv_layer;
    node.backward = bcnn_backward_depthwise_conv_layer;
    node.update = bcnn_update_depthwise_conv_layer;
    node.release_param = bcnn_release_param_depthwise_conv_layer;
    const int oh = (s.h + 2 * pad - size) / stride + 1, ow = (s.w + 2 * pad - size) / stride + 1;
    BCNN_CHECK_STATUS(add_output(net, &node, s.n, s.c, oh, ow, dst_id));
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[DepthwiseConv2D] %-8s (%4d x%4d x%4d) -> %-8s (%4d x%4d x%4d) %d x %d / %d\n", src_id,
              s.w, s.h, s.c, dst_id, ow, oh, s.c, size, size, stride);
    return BCNN_SUCCESS;
}

void bcnn_forward_depthwise_conv_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_depthwise_conv_param *p = (bcnn_depthwise_conv_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *w = &net->tensors[node->src[1]];
    bcnn_tensor *b = &net->tensors[node->src[2]], *y = &net->tensors[node->dst[0]];
    p->stats_splits = 0;
    p->raw_input = 0;
    if (hctx(net)->in_pass == 1 && p->conv_node >= 0 && ((bcnn_conv_param *)net->nodes[p->conv_node].param)->apply_skipped) {
        /* the convolution node before this one stopped after its batch statistics: its pre-normalisation output is
         * normalised while it is staged (and again by this node's backward, which needs the same values) */
        bcnn_node *cn = &net->nodes[p->conv_node];
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        const int want_stats = net->mode == BCNN_MODE_TRAIN && p->bn_node >= 0 && p->stats_gpu;
        cp->apply_skipped = 0;
        p->raw_input = 1;
        p->stats_splits = bcnn_hip_depthwise_forward_bnin(cp->bn_workspace_gpu, w->data_gpu, b->data_gpu, y->data_gpu, x->n,
                                                          x->c, x->h, x->w, p->size, p->stride, p->pad, (int)p->activation,
                                                          want_stats ? p->stats_gpu : NULL, want_stats ? p->stats_floats : 0,
                                                          cp->saved_mean.data_gpu, cp->saved_variance.data_gpu,
                                                          net->tensors[cn->src[5]].data_gpu,
                                                          net->tensors[cn->src[2]].data_gpu, (int)cp->activation);
        return;
    }
    if (hctx(net)->in_pass == 1 && net->mode == BCNN_MODE_TRAIN && p->bn_node >= 0 && p->stats_gpu) {
        /* the batch-norm node that runs next in this pass takes its statistics from this kernel's epilogue */
        p->stats_splits = bcnn_hip_depthwise_forward_stats(x->data_gpu, w->data_gpu, b->data_gpu, y->data_gpu, x->n, x->c,
                                                           x->h, x->w, p->size, p->stride, p->pad, (int)p->activation,
                                                           p->stats_gpu, p->stats_floats);
        return;
    }
    bcnn_hip_depthwise_forward(x->data_gpu, w->data_gpu, b->data_gpu, y->data_gpu, x->n, x->c, x->h, x->w, p->size,
                               p->stride, p->pad, (int)p->activation);
}

void bcnn_backward_depthwise_conv_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_depthwise_conv_param *p = (bcnn_depthwise_conv_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *w = &net->tensors[node->src[1]];
    bcnn_tensor *b = &net->tensors[node->src[2]], *y = &net->tensors[node->dst[0]];
    if (hctx(net)->in_pass == 2 && p->raw_input && p->conv_node >= 0 && x->grad_data_gpu) {
        /* this pass's forward read the producer's pre-normalisation output: the input tensor itself was never written */
        bcnn_node *cn = &net->nodes[p->conv_node];
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        const float *im = cp->saved_mean.data_gpu, *iv = cp->saved_variance.data_gpu;
        const float *is = net->tensors[cn->src[5]].data_gpu, *ib = net->tensors[cn->src[2]].data_gpu;
        /* either way the kernel also leaves the sums the producer's batch-norm backward (which runs next) starts with */
        const int want_sums = cp->insums_gpu != NULL && !BCNN_EXP_ENV("BCNN_NO_DW_INSUMS");
        if (p->bn_fused_bwd) {
            const bcnn_node *bn = &net->nodes[p->bn_node];
            const bcnn_batchnorm_param *bp = (const bcnn_batchnorm_param *)bn->param;
            cp->insums_splits = bcnn_hip_depthwise_backward_bnin_sums(
                cp->bn_workspace_gpu, w->data_gpu, y->data_gpu, net->tensors[bn->dst[0]].grad_data_gpu, x->grad_data_gpu,
                w->grad_data_gpu, b->grad_data_gpu, x->n, x->c, x->h, x->w, p->size, p->stride, p->pad, (int)p->activation,
                bcnn_grad_sole_writer(net, node->src[0]), bp->saved_mean.data_gpu, bp->saved_variance.data_gpu,
                net->tensors[bn->src[3]].data_gpu, bp->saved_mean.grad_data_gpu, bp->saved_variance.grad_data_gpu, im, iv, is,
                ib, (int)cp->activation, want_sums ? cp->insums_gpu : NULL, want_sums ? cp->insums_floats : 0);
            p->grads_pending = 1;
        } else {
            cp->insums_splits = bcnn_hip_depthwise_backward_bnin_sums(
                cp->bn_workspace_gpu, w->data_gpu, y->data_gpu, y->grad_data_gpu, x->grad_data_gpu, w->grad_data_gpu,
                b->grad_data_gpu, x->n, x->c, x->h, x->w, p->size, p->stride, p->pad, (int)p->activation,
                bcnn_grad_sole_writer(net, node->src[0]), NULL, NULL, NULL, NULL, NULL, im, iv, is, ib, (int)cp->activation,
                want_sums ? cp->insums_gpu : NULL, want_sums ? cp->insums_floats : 0);
        }
        return;
    }
    if (hctx(net)->in_pass == 2 && p->bn_fused_bwd && x->grad_data_gpu) {
        /* the batch-norm node (which ran just before in this pass) only produced its sums: its apply sweep
         * (bcnn_batchnorm_layer.c:292-296) happens inside this node's kernel, on the gradient of ITS output */
        const bcnn_node *bn = &net->nodes[p->bn_node];
        const bcnn_batchnorm_param *bp = (const bcnn_batchnorm_param *)bn->param;
        bcnn_hip_depthwise_backward_bn(x->data_gpu, w->data_gpu, y->data_gpu, net->tensors[bn->dst[0]].grad_data_gpu,
                                       x->grad_data_gpu, w->grad_data_gpu, b->grad_data_gpu, x->n, x->c, x->h, x->w,
                                       p->size, p->stride, p->pad, (int)p->activation,
                                       bcnn_grad_sole_writer(net, node->src[0]), bp->saved_mean.data_gpu,
                                       bp->saved_variance.data_gpu, net->tensors[bn->src[3]].data_gpu,
                                       bp->saved_mean.grad_data_gpu, bp->saved_variance.grad_data_gpu);
        p->grads_pending = 1;
        return;
    }
    bcnn_hip_depthwise_backward(x->data_gpu, w->data_gpu, y->data_gpu, y->grad_data_gpu, x->grad_data_gpu,
                                w->grad_data_gpu, b->grad_data_gpu, x->n, x->c, x->h, x->w, p->size, p->stride,
                                p->pad, (int)p->activation, bcnn_grad_sole_writer(net, node->src[0]));
}

void bcnn_update_depthwise_conv_layer(bcnn_net *net, bcnn_node *node) { /* bcnn_depthwise_conv_layer.c:565-610 */
    bcnn_depthwise_conv_param *p = (bcnn_depthwise_conv_param *)node->param;
    bcnn_node_optim_step(net, &net->tensors[node->src[1]], &net->tensors[node->src[2]], &p->adam_m_gpu, &p->adam_v_gpu);
}

void bcnn_release_param_depthwise_conv_layer(bcnn_node *node) {
    bcnn_depthwise_conv_param *p = (bcnn_depthwise_conv_param *)node->param;
    bcnn_hip_free(p->adam_m_gpu);
    bcnn_hip_free(p->adam_v_gpu);
    bcnn_hip_free(p->stats_gpu);
}

/* ================================================================================================
 * stand-alone batch normalisation
 * ============================================================================================== */
bcnn_status bcnn_add_batchnorm_layer(bcnn_net *net, const char *src_id, const char *dst_id) {
    bcnn_node node = {0};
    BCNN_CHECK_AND_LOG(net->log_ctx, net->num_nodes >= 1, BCNN_INVALID_PARAMETER,
                       "Batchnorm layer can't be the first layer of the network\n");
    BCNN_CHECK_STATUS(attach_source(net, &node, src_id, "Batchnorm"));
    const bcnn_tensor s = net->tensors[node.src[0]];
    BCNN_CHECK_STATUS(add_output(net, &node, s.n, s.c, s.h, s.w, dst_id));
    node.type = BCNN_LAYER_BATCHNORM;
    node.param_size = sizeof(bcnn_batchnorm_param);
    bcnn_batchnorm_param *param = (bcnn_batchnorm_param *)calloc(1, node.param_size);
    node.param = param;
    param->dw_node = -1;
    param->sums_conv = -1;
    param->fold_conv = -1;
    node.forward = bcnn_forward_batchnorm_layer;
    node.backward = bcnn_backward_batchnorm_layer;
    node.release_param = bcnn_release_param_batchnorm_layer;
    char name[256];
    snprintf(name, sizeof(name), "%s_sav_mean", src_id);
    bcnn_tensor_create(&param->saved_mean, 1, 1, 1, s.c, 1, name, net->mode);
    snprintf(name, sizeof(name), "%s_sav_var", src_id);
    bcnn_tensor_create(&param->saved_variance, 1, 1, 1, s.c, 1, name, net->mode);
    BCNN_CHECK_STATUS(add_vector_param(net, &node, s.c, 0, src_id, "run_mean", 0.0f, 0));
    BCNN_CHECK_STATUS(add_vector_param(net, &node, s.c, 0, src_id, "run_var", 0.0f, 0));
    BCNN_CHECK_STATUS(add_vector_param(net, &node, s.c, 1, src_id, "scales", 1.0f, 0));
    BCNN_CHECK_STATUS(add_vector_param(net, &node, s.c, 1, src_id, "b", 0.0f, 0));
    if (net->mode != BCNN_MODE_PREDICT) param->workspace_gpu = bcnn_hip_malloc_f32((size_t)bcnn_tensor_size(&s));
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[Batchnorm] %-8s (%4d x%4d x%4d) -> %-8s\n", src_id, s.w, s.h, s.c, dst_id);
    return BCNN_SUCCESS;
}

/* the copy of the input the reference keeps for the backward pass (bcnn_batchnorm_layer.c:208): the input tensor itself
 * when nothing can overwrite it before then (input_kept, decided by bcnn_link_depthwise_batchnorm) */
static float *batchnorm_kept_input(const bcnn_batchnorm_param *p, const bcnn_tensor *x) {
    return (p->input_kept || !p->workspace_gpu) ? x->data_gpu : p->workspace_gpu;
}

void bcnn_forward_batchnorm_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_batchnorm_param *p = (bcnn_batchnorm_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    const float *stats = NULL;
    int splits = 0;
    if (hctx(net)->in_pass == 1 && net->mode == BCNN_MODE_TRAIN && p->dw_node >= 0) {
        /* the depthwise node that ran just before in this pass left the statistics of the tensor it wrote */
        bcnn_depthwise_conv_param *dp = (bcnn_depthwise_conv_param *)net->nodes[p->dw_node].param;
        stats = dp->stats_gpu;
        splits = dp->stats_splits;
        dp->stats_splits = 0;
    }
    p->apply_skipped = 0;
    p->data_pending = 0;
    if (hctx(net)->in_pass == 1 && net->mode == BCNN_MODE_TRAIN && p->fold_conv >= 0) {
        /* the 1x1 convolution that runs next in this pass multiplies this node's INPUT by weights that carry this node's
         * affine map: statistics only, the output tensor is not written (bcnn_materialize_data produces it on demand) */
        bcnn_hip_batchnorm_forward_stats_only(x->data_gpu, net->tensors[node->src[1]].data_gpu,
                                              net->tensors[node->src[2]].data_gpu, net->tensors[node->src[3]].data_gpu,
                                              net->tensors[node->src[4]].data_gpu, p->saved_mean.data_gpu,
                                              p->saved_variance.data_gpu, x->n, x->c, x->h * x->w, stats, splits);
        p->apply_skipped = 1;
        p->data_pending = 1;
        return;
    }
    bcnn_hip_batchnorm_forward_stats(x->data_gpu, y->data_gpu, net->tensors[node->src[1]].data_gpu,
                                     net->tensors[node->src[2]].data_gpu, net->tensors[node->src[3]].data_gpu,
                                     net->tensors[node->src[4]].data_gpu, p->saved_mean.data_gpu,
                                     p->saved_variance.data_gpu, p->x_norm_gpu,
                                     net->mode == BCNN_MODE_PREDICT ? NULL : batchnorm_kept_input(p, x), x->n, x->c,
                                     x->h * x->w, (int)net->mode, BCNN_HIP_ACT_NONE, stats, splits);
}

void bcnn_backward_batchnorm_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_batchnorm_param *p = (bcnn_batchnorm_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    bcnn_tensor *scales = &net->tensors[node->src[3]], *biases = &net->tensors[node->src[4]];
    if (hctx(net)->in_pass == 2 && p->dw_fused_bwd && net->tensors[net->nodes[p->dw_node].src[0]].grad_data_gpu) {
        /* sums only: the depthwise node that runs next in this pass applies :292-296 inside its own kernel */
        if (p->bsums_splits > 0) { /* and the 1x1 convolution that ran just before already left the partial sums */
            const int splits = p->bsums_splits;
            p->bsums_splits = 0;
            bcnn_hip_batchnorm_backward_finalize(p->bsums_gpu, splits, scales->data_gpu, scales->grad_data_gpu,
                                                 biases->grad_data_gpu, p->saved_variance.data_gpu,
                                                 p->saved_mean.grad_data_gpu, p->saved_variance.grad_data_gpu, x->c);
            return;
        }
        bcnn_hip_batchnorm_backward_sums(y->grad_data_gpu, scales->data_gpu, scales->grad_data_gpu, biases->grad_data_gpu,
                                         p->saved_mean.data_gpu, p->saved_variance.data_gpu,
                                         p->saved_mean.grad_data_gpu, p->saved_variance.grad_data_gpu,
                                         batchnorm_kept_input(p, x), x->n, x->c, x->h * x->w);
        return;
    }
    p->bsums_splits = 0; /* sums left by the consumer are used by the sums-only path above alone */
    /* VALID-mode backward would use the running statistics (reference :306-309); only TRAIN is meaningful */
    bcnn_hip_batchnorm_backward(y->grad_data_gpu, x->grad_data_gpu, NULL, BCNN_HIP_ACT_NONE, scales->data_gpu,
                                scales->grad_data_gpu, biases->grad_data_gpu, p->saved_mean.data_gpu,
                                p->saved_variance.data_gpu, p->saved_mean.grad_data_gpu,
                                p->saved_variance.grad_data_gpu, p->x_norm_gpu, batchnorm_kept_input(p, x), x->n, x->c,
                                x->h * x->w);
}

/* Pairs a convolution node (batch-norm, no activation) with the eltwise node right behind it that takes the convolution
 * output as its FIRST operand (the one every element of which is used; the reference adds only min_c * h * w elements of
 * the second, bcnn_eltwise_layer.c:96-101) -- the tail of a residual block. TRAIN-mode nets only (the pre-normalisation
 * workspace has to exist). Recomputed at every compile. */
void bcnn_link_conv_eltwise(bcnn_net *net) {
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->type == BCNN_LAYER_CONV2D) {
            ((bcnn_conv_param *)nd->param)->elt_node = -1;
            ((bcnn_conv_param *)nd->param)->data_pending = 0;
        } else if (nd->type == BCNN_LAYER_ELTWISE) {
            bcnn_eltwise_param *ep = (bcnn_eltwise_param *)nd->param;
            ep->conv_node = -1;
            ep->done_forward = ep->deferred = ep->grad_pending = 0;
        }
    }
    if (BCNN_EXP_ENV("BCNN_NO_NODE_FUSION")) return; /* A/B switch of the experiment build */
    for (int e = 1; e < net->num_nodes; ++e) {
        bcnn_node *en = &net->nodes[e], *cn = &net->nodes[e - 1];
        if (en->type != BCNN_LAYER_ELTWISE || cn->type != BCNN_LAYER_CONV2D) continue;
        bcnn_eltwise_param *ep = (bcnn_eltwise_param *)en->param;
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        const int t = cn->dst[0];
        if (ep->stride[0] != 1 || ep->stride[1] != 1 || en->src[0] != t || en->src[1] == t || en->dst[0] == t ||
            en->dst[0] == en->src[1])
            continue;
        int writers = 0, consumers = 0;
        for (int i = 0; i < net->num_nodes; ++i) {
            for (int k = 0; k < net->nodes[i].num_dst; ++k) writers += net->nodes[i].dst[k] == t;
            for (int k = 0; k < net->nodes[i].num_src; ++k) consumers += net->nodes[i].src[k] == t;
        }
        if (writers != 1 || consumers != 1) continue;
        const bcnn_tensor *y = &net->tensors[t], *out = &net->tensors[en->dst[0]], *r = &net->tensors[en->src[1]];
        if (bcnn_tensor_size(y) != bcnn_tensor_size(out) || !out->grad_data_gpu || !y->grad_data_gpu) continue;
        if ((size_t)ep->min_dim[0] * out->h * out->w > (size_t)bcnn_tensor_size(r)) continue;
        if (!bcnn_hip_conv_residual_fusable(cp->batch_norm, (int)cp->activation, (int)ep->activation, (int)net->mode,
                                            cp->bn_workspace_gpu, r->data_gpu, out->data_gpu))
            continue;
        cp->elt_node = e;
        ep->conv_node = e - 1;
    }
}

/* Pairs a convolution node (batch-norm, cheap activation) with the stride-2 max-pooling node right behind it when that node
 * is the only consumer of the convolution output (the ResNet stem), TRAIN-mode nets: forward, the pooling kernel reads the
 * pre-normalisation values and normalises them on the fly (bcnn_hip_maxpool_forward_bn); the normalised tensor -- four
 * times the pooled one -- is not written (bcnn_materialize_data produces it on demand; the backward pass never reads it:
 * the activation derivative is recomputed from the pre-normalisation values). */
void bcnn_link_conv_maxpool(bcnn_net *net) {
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->type == BCNN_LAYER_CONV2D) {
            ((bcnn_conv_param *)nd->param)->pool_node = -1;
            ((bcnn_conv_param *)nd->param)->apply_skipped = 0;
            ((bcnn_conv_param *)nd->param)->pool_bwd_pending = 0;
        } else if (nd->type == BCNN_LAYER_MAXPOOL) {
            ((bcnn_maxpool_param *)nd->param)->conv_node = -1;
            ((bcnn_maxpool_param *)nd->param)->raw_fwd = 0;
        }
    }
    if (BCNN_EXP_ENV("BCNN_NO_NODE_FUSION") || net->mode != BCNN_MODE_TRAIN) return;
    for (int m = 1; m < net->num_nodes; ++m) {
        bcnn_node *pn = &net->nodes[m], *cn = &net->nodes[m - 1];
        if (pn->type != BCNN_LAYER_MAXPOOL || cn->type != BCNN_LAYER_CONV2D || pn->src[0] != cn->dst[0]) continue;
        const int t = cn->dst[0];
        int writers = 0, consumers = 0;
        for (int i = 0; i < net->num_nodes; ++i) {
            for (int k = 0; k < net->nodes[i].num_dst; ++k) writers += net->nodes[i].dst[k] == t;
            for (int k = 0; k < net->nodes[i].num_src; ++k) consumers += net->nodes[i].src[k] == t;
        }
        if (writers != 1 || consumers != 1) continue;
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        bcnn_maxpool_param *mp = (bcnn_maxpool_param *)pn->param;
        const bcnn_tensor *y = &net->tensors[t], *py = &net->tensors[pn->dst[0]];
        /* the backward pass must not need the normalised tensor: cheap activations are recomputed from the workspace */
        if (!cp->batch_norm || !cp->bn_workspace_gpu || cp->activation == BCNN_ACT_PRELU ||
            !bcnn_hip_maxpool_bn_fusable(y->n, y->c, y->h, y->w, py->h, py->w, mp->size, mp->stride, (int)cp->activation,
                                         cp->bn_workspace_gpu))
            continue;
        cp->pool_node = m;
        mp->conv_node = m - 1;
        if (!mp->raw_at_max_gpu) mp->raw_at_max_gpu = bcnn_hip_malloc_f32((size_t)bcnn_tensor_size(py));
    }
}

/* Pairs a convolution node (batch-norm, cheap activation) with the depthwise node right behind it when that node is the
 * only consumer of the convolution output and runs on the LDS-staged kernels (MobileNet: every 1x1 convolution and the
 * stem), TRAIN-mode nets: the convolution node stops after its batch statistics, the depthwise kernels normalise its
 * pre-normalisation output while staging it -- forward and backward (the weight gradient needs the same input values).
 * The convolution output tensor is not written (bcnn_materialize_data). */
void bcnn_link_conv_depthwise(bcnn_net *net) {
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->type == BCNN_LAYER_CONV2D) {
            ((bcnn_conv_param *)nd->param)->dw_node = -1;
            ((bcnn_conv_param *)nd->param)->insums_splits = 0;
        }
        else if (nd->type == BCNN_LAYER_DEPTHWISE_CONV2D) {
            ((bcnn_depthwise_conv_param *)nd->param)->conv_node = -1;
            ((bcnn_depthwise_conv_param *)nd->param)->raw_input = 0;
        }
    }
    if (BCNN_EXP_ENV("BCNN_NO_NODE_FUSION") || BCNN_EXP_ENV("BCNN_NO_CONV_DW_FUSION") || net->mode != BCNN_MODE_TRAIN) return;
    for (int d = 1; d < net->num_nodes; ++d) {
        bcnn_node *dn = &net->nodes[d], *cn = &net->nodes[d - 1];
        if (dn->type != BCNN_LAYER_DEPTHWISE_CONV2D || cn->type != BCNN_LAYER_CONV2D || dn->src[0] != cn->dst[0]) continue;
        const int t = cn->dst[0];
        int writers = 0, consumers = 0;
        for (int i = 0; i < net->num_nodes; ++i) {
            for (int k = 0; k < net->nodes[i].num_dst; ++k) writers += net->nodes[i].dst[k] == t;
            for (int k = 0; k < net->nodes[i].num_src; ++k) consumers += net->nodes[i].src[k] == t;
        }
        if (writers != 1 || consumers != 1) continue;
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        bcnn_depthwise_conv_param *dp = (bcnn_depthwise_conv_param *)dn->param;
        const bcnn_tensor *x = &net->tensors[t];
        if (!cp->batch_norm || !cp->bn_workspace_gpu || !x->grad_data_gpu ||
            ((uintptr_t)cp->bn_workspace_gpu & 15) != 0 ||
            !bcnn_hip_depthwise_bnin_fusable(x->n, x->c, x->h, x->w, dp->size, dp->stride, dp->pad, (int)dp->activation,
                                             (int)cp->activation))
            continue;
        cp->dw_node = d;
        dp->conv_node = d - 1;
        /* the depthwise node's backward kernel can leave the sums this node's batch-norm backward starts with */
        const size_t need = bcnn_hip_depthwise_insums_size(x->n, x->c, x->h, x->w, dp->size, dp->stride, dp->pad);
        if (need > cp->insums_floats) {
            bcnn_hip_sync();
            bcnn_hip_free(cp->insums_gpu);
            cp->insums_gpu = bcnn_hip_malloc_f32(need);
            cp->insums_floats = need;
        }
    }
}

/* Pairs a stand-alone batch-norm node whose backward is of the sums-only kind (dw_fused_bwd: the depthwise node in front of
 * it applies the rest) with the 1x1 / stride 1 / one-group convolution node right behind it that is its only consumer
 * (MobileNet: [depthwise] -> [batchnorm] -> [conv 1x1]): that node's data-gradient kernel then emits the sums. Call after
 * bcnn_link_depthwise_batchnorm. */
void bcnn_link_batchnorm_conv(bcnn_net *net) {
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->type == BCNN_LAYER_CONV2D) {
            ((bcnn_conv_param *)nd->param)->bnsums_node = -1;
            ((bcnn_conv_param *)nd->param)->fold_bn = -1;
            ((bcnn_conv_param *)nd->param)->folded = 0;
        } else if (nd->type == BCNN_LAYER_BATCHNORM) {
            ((bcnn_batchnorm_param *)nd->param)->sums_conv = -1;
            ((bcnn_batchnorm_param *)nd->param)->bsums_splits = 0;
            ((bcnn_batchnorm_param *)nd->param)->fold_conv = -1;
            ((bcnn_batchnorm_param *)nd->param)->apply_skipped = 0;
        }
    }
    if (BCNN_EXP_ENV("BCNN_NO_NODE_FUSION") || BCNN_EXP_ENV("BCNN_NO_BN_CONV_FUSION")) return;
    for (int c = 1; c < net->num_nodes; ++c) {
        bcnn_node *cn = &net->nodes[c], *bn = &net->nodes[c - 1];
        if (cn->type != BCNN_LAYER_CONV2D || bn->type != BCNN_LAYER_BATCHNORM || cn->src[0] != bn->dst[0]) continue;
        bcnn_batchnorm_param *bp = (bcnn_batchnorm_param *)bn->param;
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        if (!bp->dw_fused_bwd || cp->size != 1 || cp->stride != 1 || cp->pad != 0 || cp->num_groups != 1) continue;
        const int t = bn->dst[0];
        int writers = 0, consumers = 0;
        for (int i = 0; i < net->num_nodes; ++i) {
            for (int k = 0; k < net->nodes[i].num_dst; ++k) writers += net->nodes[i].dst[k] == t;
            for (int k = 0; k < net->nodes[i].num_src; ++k) consumers += net->nodes[i].src[k] == t;
        }
        const bcnn_tensor *z = &net->tensors[t];
        if (writers != 1 || consumers != 1 || !z->grad_data_gpu) continue;
        const size_t need = bcnn_hip_conv_bnsums_size(z->n, z->c, z->h, z->w);
        if (need > bp->bsums_floats) {
            bcnn_hip_sync();
            bcnn_hip_free(bp->bsums_gpu);
            bp->bsums_gpu = bcnn_hip_malloc_f32(need);
            bp->bsums_floats = need;
        }
        bp->sums_conv = c;
        cp->bnsums_node = c - 1;
        /* forward: the same pair can drop the batch-norm's apply sweep -- its affine map goes into the convolution's packed
         * weights (the parity bar is per reference-visible tensor, not operation order). Needs the
         * convolution's own batch-norm behind (it absorbs the constant W b), TRAIN mode, the batch-norm's input still in
         * place at backward time (input_kept), and kernels that take the folded form. */
        if (net->mode == BCNN_MODE_TRAIN && !BCNN_EXP_ENV("BCNN_NO_BN_FOLD") && cp->batch_norm && bp->input_kept &&
            bcnn_hip_conv_bnfold_fusable(z->n, z->c, z->h, z->w, cp->num)) {
            bp->fold_conv = c;
            cp->fold_bn = c - 1;
        }
    }
}

/* What a fused forward pass did not write: the output tensor of a convolution node whose result went straight into the
 * eltwise node behind it. Its pre-normalisation values and batch statistics are in place, so the tensor is one batch-norm
 * apply sweep away (bcnn_batchnorm_layer.c:226-241 with the saved statistics). */
void bcnn_materialize_data(bcnn_net *net, int tensor) {
    for (int i = 0; i < net->num_nodes; ++i) { /* batch-norm nodes whose apply sweep went into the next node's weights */
        bcnn_node *bn = &net->nodes[i];
        if (bn->type != BCNN_LAYER_BATCHNORM) continue;
        bcnn_batchnorm_param *bp = (bcnn_batchnorm_param *)bn->param;
        if (!bp->data_pending || (tensor >= 0 && tensor != bn->dst[0])) continue;
        bp->data_pending = 0;
        const bcnn_tensor *x = &net->tensors[bn->src[0]];
        bcnn_tensor *z = &net->tensors[bn->dst[0]];
        bcnn_hip_batchnorm_apply(x->data_gpu, z->data_gpu, net->tensors[bn->src[3]].data_gpu,
                                 net->tensors[bn->src[4]].data_gpu, bp->saved_mean.data_gpu, bp->saved_variance.data_gpu,
                                 z->n, z->c, z->h * z->w, BCNN_HIP_ACT_NONE);
    }
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *cn = &net->nodes[i];
        if (cn->type != BCNN_LAYER_CONV2D) continue;
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        if (!cp->data_pending || (tensor >= 0 && tensor != cn->dst[0])) continue;
        cp->data_pending = 0;
        bcnn_tensor *y = &net->tensors[cn->dst[0]];
        bcnn_hip_batchnorm_apply(cp->bn_workspace_gpu, y->data_gpu, net->tensors[cn->src[5]].data_gpu,
                                 net->tensors[cn->src[2]].data_gpu, cp->saved_mean.data_gpu, cp->saved_variance.data_gpu,
                                 y->n, y->c, y->h * y->w, (int)cp->activation);
    }
}

/* The fused backward of a depthwise / batch-norm pair leaves two gradient tensors unwritten that the reference's workers
 * rewrite in place: the batch-norm node's dst gradient (-> gradient w.r.t. its input, bcnn_batchnorm_layer.c:292-296)
 * and the depthwise node's dst gradient (that, times act'(y), bcnn_depthwise_conv_layer.c:311-314). Every input of those
 * two sweeps is still in place after the pass, so a caller that asks for either tensor (bcnn_get_tensor_by_*,
 * bcnn_download_tensor) gets them produced here, once, with the unfused kernels. */
void bcnn_materialize_gradients(bcnn_net *net, int tensor) {
    for (int i = 0; i < net->num_nodes; ++i) { /* eltwise nodes whose backward rode on the convolution node before them */
        bcnn_node *en = &net->nodes[i];
        if (en->type != BCNN_LAYER_ELTWISE) continue;
        bcnn_eltwise_param *ep = (bcnn_eltwise_param *)en->param;
        if (!ep->grad_pending || (tensor >= 0 && tensor != en->dst[0])) continue;
        ep->grad_pending = 0;
        bcnn_tensor *y = &net->tensors[en->dst[0]];
        if (y->grad_data_gpu)
            bcnn_hip_activation_backward(y->data_gpu, y->grad_data_gpu, (size_t)bcnn_tensor_size(y), (int)ep->activation, NULL,
                                         NULL, y->h * y->w, y->c);
    }
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *dw = &net->nodes[i];
        if (dw->type != BCNN_LAYER_DEPTHWISE_CONV2D) continue;
        bcnn_depthwise_conv_param *dp = (bcnn_depthwise_conv_param *)dw->param;
        if (!dp->grads_pending || dp->bn_node < 0 || dp->bn_node >= net->num_nodes) continue;
        bcnn_node *bn = &net->nodes[dp->bn_node];
        if (tensor >= 0 && tensor != dw->dst[0] && tensor != bn->dst[0]) continue;
        dp->grads_pending = 0;
        const bcnn_batchnorm_param *bp = (const bcnn_batchnorm_param *)bn->param;
        bcnn_tensor *y = &net->tensors[dw->dst[0]], *z = &net->tensors[bn->dst[0]];
        if (!y->grad_data_gpu || !z->grad_data_gpu) continue;
        bcnn_hip_batchnorm_backward_apply(z->grad_data_gpu, y->grad_data_gpu, net->tensors[bn->src[3]].data_gpu,
                                          bp->saved_mean.data_gpu, bp->saved_variance.data_gpu,
                                          bp->saved_mean.grad_data_gpu, bp->saved_variance.grad_data_gpu, y->data_gpu,
                                          y->n, y->c, y->h * y->w);
        bcnn_hip_activation_backward(y->data_gpu, y->grad_data_gpu, (size_t)bcnn_tensor_size(y), (int)dp->activation, NULL,
                                     NULL, y->h * y->w, y->c);
    }
}

void bcnn_drop_pending_gradients(bcnn_net *net) { /* a new forward pass: the reference zero-fills them (bcnn_net.c:361-375) */
    for (int i = 0; i < net->num_nodes; ++i) {
        if (net->nodes[i].type == BCNN_LAYER_DEPTHWISE_CONV2D)
            ((bcnn_depthwise_conv_param *)net->nodes[i].param)->grads_pending = 0;
        else if (net->nodes[i].type == BCNN_LAYER_ELTWISE)
            ((bcnn_eltwise_param *)net->nodes[i].param)->grad_pending = 0;
    }
}

/* Pairs every depthwise node with the stand-alone batch-norm node that consumes its output (bcnn_compile_net; the
 * MobileNet block of the reference's configs: [depthwise-conv] -> [batchnorm]). Recomputed from scratch at every compile:
 * node indices, not pointers (the node array moves when nodes are added). */
void bcnn_link_depthwise_batchnorm(bcnn_net *net) {
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->type == BCNN_LAYER_DEPTHWISE_CONV2D) {
            bcnn_depthwise_conv_param *dp = (bcnn_depthwise_conv_param *)nd->param;
            dp->bn_node = -1;
            dp->bn_fused_bwd = 0;
            dp->stats_splits = 0;
            dp->grads_pending = 0;
        } else if (nd->type == BCNN_LAYER_BATCHNORM) {
            bcnn_batchnorm_param *bp = (bcnn_batchnorm_param *)nd->param;
            bp->dw_node = -1;
            bp->dw_fused_bwd = 0;
            bp->input_kept = 0;
        }
    }
    if (BCNN_EXP_ENV("BCNN_NO_NODE_FUSION")) return; /* A/B switch of the experiment build */
    for (int j = 0; j < net->num_nodes; ++j) {
        bcnn_node *bn = &net->nodes[j];
        if (bn->type != BCNN_LAYER_BATCHNORM) continue;
        bcnn_batchnorm_param *bp = (bcnn_batchnorm_param *)bn->param;
        const int t = bn->src[0];
        int producer = -1, writers = 0, consumers = 0;
        for (int i = 0; i < net->num_nodes; ++i) {
            for (int k = 0; k < net->nodes[i].num_dst; ++k)
                if (net->nodes[i].dst[k] == t) {
                    producer = i;
                    ++writers;
                }
            for (int k = 0; k < net->nodes[i].num_src; ++k)
                if (net->nodes[i].src[k] == t) ++consumers;
        }
        /* nobody else reads or rewrites the input between this node's forward and its backward */
        const int private_input = writers == 1 && consumers == 1 && bn->dst[0] != t;
        bp->input_kept = private_input;
        if (private_input && bp->workspace_gpu) { /* the copy is not needed */
            bcnn_hip_sync();
            bcnn_hip_free(bp->workspace_gpu);
            bp->workspace_gpu = NULL;
        } else if (!private_input && !bp->workspace_gpu && net->mode != BCNN_MODE_PREDICT) {
            bp->workspace_gpu = bcnn_hip_malloc_f32((size_t)bcnn_tensor_size(&net->tensors[t]));
        }
        if (producer != j - 1 || writers != 1 || net->nodes[producer].type != BCNN_LAYER_DEPTHWISE_CONV2D) continue;
        bcnn_node *dw = &net->nodes[producer];
        bcnn_depthwise_conv_param *dp = (bcnn_depthwise_conv_param *)dw->param;
        const bcnn_tensor *x = &net->tensors[dw->src[0]];
        /* forward: the statistics of the depthwise output are the same whoever else reads it */
        size_t need = bcnn_hip_depthwise_stats_size(x->n, x->c, x->h, x->w, dp->size, dp->stride, dp->pad);
        if (BCNN_EXP_ENV("BCNN_NO_DW_STATS")) need = 0; /* experiment build: forward hand-off off, backward fusion on */
        if (need > dp->stats_floats) {
            bcnn_hip_sync();
            bcnn_hip_free(dp->stats_gpu);
            dp->stats_gpu = bcnn_hip_malloc_f32(need);
            dp->stats_floats = need;
        } else if (need == 0 && dp->stats_gpu) { /* no scratch = no hand-off (bcnn_forward_depthwise_conv_layer) */
            bcnn_hip_sync();
            bcnn_hip_free(dp->stats_gpu);
            dp->stats_gpu = NULL;
            dp->stats_floats = 0;
        }
        dp->bn_node = j;
        bp->dw_node = producer;
        /* backward: the gradient of the depthwise output has this node as its only writer and that node as its only
         * reader, so it does not have to exist */
        if (private_input &&
            bcnn_hip_depthwise_bn_fusable(x->n, x->c, x->h, x->w, dp->size, dp->stride, dp->pad, (int)dp->activation)) {
            dp->bn_fused_bwd = 1;
            bp->dw_fused_bwd = 1;
        }
    }
}

void bcnn_release_param_batchnorm_layer(bcnn_node *node) {
    bcnn_batchnorm_param *p = (bcnn_batchnorm_param *)node->param;
    bcnn_tensor_destroy(&p->saved_mean);
    bcnn_tensor_destroy(&p->saved_variance);
    bcnn_hip_free(p->workspace_gpu);
    bcnn_hip_free(p->x_norm_gpu);
    bcnn_hip_free(p->bsums_gpu);
}

/* ================================================================================================
 * pooling
 * ============================================================================================== */
static int pooled_extent(int in, int size, int stride, bcnn_padding padding) {
    switch (padding) { /* reference bcnn_maxpool_layer.c:62-83 */
        case BCNN_PADDING_SAME: return (in + stride - 1) / stride;
        case BCNN_PADDING_VALID: return (in - size + stride) / stride;
        case BCNN_PADDING_CAFFE: return (int)(ceil((float)(in - size) / stride)) + 1;
    }
    return 0;
}

bcnn_status bcnn_add_maxpool_layer(bcnn_net *net, int size, int stride, bcnn_padding padding, const char *src_id,
                                   const char *dst_id) {
    bcnn_node node = {0};
    BCNN_CHECK_STATUS(attach_source(net, &node, src_id, "Maxpool"));
    const bcnn_tensor s = net->tensors[node.src[0]];
    const int oh = pooled_extent(s.h, size, stride, padding), ow = pooled_extent(s.w, size, stride, padding);
    BCNN_CHECK_STATUS(add_output(net, &node, s.n, s.c, oh, ow, dst_id));
    node.type = BCNN_LAYER_MAXPOOL;
    node.param_size = sizeof(bcnn_maxpool_param);
    bcnn_maxpool_param *param = (bcnn_maxpool_param *)calloc(1, node.param_size);
    node.param = param;
    param->size = size; param->stride = stride; param->padding = padding;
    param->conv_node = -1;
    const size_t sz = (size_t)s.n * s.c * oh * ow;
    param->indexes = (int *)calloc(sz, sizeof(int));
    param->indexes_gpu = bcnn_hip_malloc_i32(sz);
    node.forward = bcnn_forward_maxpool_layer;
    node.backward = bcnn_backward_maxpool_layer;
    node.release_param = bcnn_release_param_maxpool_layer;
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[Maxpool] %-8s (%4d x%4d x%4d) -> %-8s (%4d x%4d x%4d) %d x %d / %d\n", src_id, s.w, s.h,
              s.c, dst_id, ow, oh, s.c, size, size, stride);
    return BCNN_SUCCESS;
}

void bcnn_forward_maxpool_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_maxpool_param *p = (bcnn_maxpool_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    p->raw_fwd = 0;
    if (hctx(net)->in_pass == 1 && p->conv_node >= 0) {
        bcnn_node *cn = &net->nodes[p->conv_node];
        bcnn_conv_param *cp = (bcnn_conv_param *)cn->param;
        if (cp->apply_skipped) { /* the convolution node before this one left its pre-normalisation output and statistics */
            cp->apply_skipped = 0;
            bcnn_hip_maxpool_forward_bn_keep(cp->bn_workspace_gpu, y->data_gpu, p->indexes_gpu, x->n, x->c, x->h, x->w, y->h,
                                             y->w, p->size, p->stride, net->tensors[cn->src[5]].data_gpu,
                                             net->tensors[cn->src[2]].data_gpu, cp->saved_mean.data_gpu,
                                             cp->saved_variance.data_gpu, (int)cp->activation, p->raw_at_max_gpu);
            p->raw_fwd = p->raw_at_max_gpu != NULL;
            return;
        }
    }
    bcnn_hip_maxpool_forward(x->data_gpu, y->data_gpu, p->indexes_gpu, x->n, x->c, x->h, x->w, y->h, y->w, p->size,
                             p->stride);
}

void bcnn_backward_maxpool_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_maxpool_param *p = (bcnn_maxpool_param *)node->param;
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    if (!x->grad_data_gpu) return;
    if (hctx(net)->in_pass == 2 && p->conv_node >= 0 && p->raw_fwd && !BCNN_EXP_ENV("BCNN_NO_POOL_BWD_FUSION") &&
        bcnn_grad_sole_writer(net, node->src[0])) {
        /* the convolution node that runs next in this pass: one kernel gathers this node's gradient and applies that node's
         * batch-norm backward to it in registers; the sums it needs are taken over the pooled tensors */
        bcnn_conv_param *cp = (bcnn_conv_param *)net->nodes[p->conv_node].param;
        if (bcnn_hip_maxpool_bn_backward_fusable(x->n, x->c, x->h, x->w, y->h, y->w, p->size, p->stride, (int)cp->activation,
                                                 cp->bn_workspace_gpu, y->grad_data_gpu, p->indexes_gpu, x->grad_data_gpu)) {
            cp->pool_bwd_pending = 1;
            return;
        }
    }
    bcnn_hip_maxpool_backward(y->grad_data_gpu, p->indexes_gpu, x->grad_data_gpu, x->n, x->c, x->h, x->w, y->h, y->w,
                              p->size, p->stride, bcnn_grad_sole_writer(net, node->src[0]));
}

void bcnn_release_param_maxpool_layer(bcnn_node *node) {
    bcnn_maxpool_param *p = (bcnn_maxpool_param *)node->param;
    free(p->indexes);
    bcnn_hip_free(p->indexes_gpu);
    bcnn_hip_free(p->raw_at_max_gpu);
}

bcnn_status bcnn_add_avgpool_layer(bcnn_net *net, const char *src_id, const char *dst_id) {
    bcnn_node node = {0};
    BCNN_CHECK_STATUS(attach_source(net, &node, src_id, "Avgpool"));
    const bcnn_tensor s = net->tensors[node.src[0]];
    BCNN_CHECK_STATUS(add_output(net, &node, s.n, s.c, 1, 1, dst_id));
    node.type = BCNN_LAYER_AVGPOOL;
    node.forward = bcnn_forward_avgpool_layer;
    node.backward = bcnn_backward_avgpool_layer;
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[Avgpool] %-8s (%4d x%4d x%4d) -> %-8s (1 x 1 x %d)\n", src_id, s.w, s.h, s.c, dst_id, s.c);
    return BCNN_SUCCESS;
}

void bcnn_forward_avgpool_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    bcnn_hip_avgpool_forward(x->data_gpu, y->data_gpu, x->n, x->c, x->h, x->w);
}

void bcnn_backward_avgpool_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_tensor *x = &net->tensors[node->src[0]], *y = &net->tensors[node->dst[0]];
    if (!x->grad_data_gpu) return;
    bcnn_hip_avgpool_backward(y->grad_data_gpu, x->grad_data_gpu, x->n, x->c, x->h, x->w);
}

/* ================================================================================================
 * stand-alone activation node (in place: src index == dst index, bcnn_activation_layer.c:46-47).
 * The reference's CPU worker dereferences a NULL weights tensor for every non-PReLU activation
 * (:152-160); here the node simply works for all of them.
 * ============================================================================================== */
bcnn_status bcnn_add_activation_layer(bcnn_net *net, bcnn_activation type, const char *src_id) {
    bcnn_node node = {0};
    BCNN_CHECK_AND_LOG(net->log_ctx, net->num_nodes >= 1, BCNN_INVALID_PARAMETER,
                       "Activation layer can't be the first layer of the network\n");
    const int idx = bcnn_net_find_tensor(net, src_id);
    BCNN_CHECK_AND_LOG(net->log_ctx, idx >= 0, BCNN_INVALID_PARAMETER,
                       "Activation layer: invalid input node name %s\n", src_id);
    BCNN_CHECK_STATUS(bcnn_node_add_input(net, &node, idx));
    BCNN_CHECK_STATUS(bcnn_node_add_output(net, &node, idx));
    node.type = BCNN_LAYER_ACTIVATION;
    node.param_size = sizeof(bcnn_activation_param);
    bcnn_activation_param *param = (bcnn_activation_param *)calloc(1, node.param_size);
    node.param = param;
    param->activation = type;
    node.forward = bcnn_forward_activation_layer;
    node.backward = bcnn_backward_activation_layer;
    node.update = bcnn_update_activation_layer;
    if (type == BCNN_ACT_PRELU) {
        char name[256];
        snprintf(name, sizeof(name), "%s_w_prelu", src_id);
        bcnn_tensor slopes = {0};
        bcnn_tensor_create(&slopes, 1, 1, 1, net->tensors[idx].c, 1, name, net->mode);
        BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, slopes));
        BCNN_CHECK_STATUS(bcnn_node_add_input(net, &node, net->num_tensors - 1));
        bcnn_net_register_param(net, net->num_tensors - 1);
    }
    BCNN_CHECK_STATUS(bcnn_net_add_node(net, node));
    BCNN_INFO(net->log_ctx, "[%s] %-8s (in place)\n", bcnn_act2str(type), src_id);
    return BCNN_SUCCESS;
}

void bcnn_forward_activation_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_activation_param *p = (bcnn_activation_param *)node->param;
    bcnn_tensor *t = &net->tensors[node->dst[0]];
    const float *slopes = (p->activation == BCNN_ACT_PRELU) ? net->tensors[node->src[1]].data_gpu : NULL;
    bcnn_hip_activation_forward(t->data_gpu, (size_t)bcnn_tensor_size(t), (int)p->activation, slopes, t->w * t->h, t->c);
}

void bcnn_backward_activation_layer(bcnn_net *net, bcnn_node *node) {
    bcnn_activation_param *p = (bcnn_activation_param *)node->param;
    bcnn_tensor *t = &net->tensors[node->dst[0]];
    if (!t->grad_data_gpu) return;
    bcnn_tensor *sl = (p->activation == BCNN_ACT_PRELU) ? &net->tensors[node->src[1]] : NULL;
    bcnn_hip_activation_backward(t->data_gpu, t->grad_data_gpu, (size_t)bcnn_tensor_size(t), (int)p->activation,
                                 sl ? sl->data_gpu : NULL, sl ? sl->grad_data_gpu : NULL, t->w * t->h, t->c);
}

void bcnn_update_activation_layer(bcnn_net *net, bcnn_node *node) {
    /* reference bcnn_activation_layer.c:262-291: momentum-SGD with the weights rule for both optimizers, and
     * batch_size = weights->n, which is 1 for the [1,1,1,C] slope tensor -- NOT the net's batch size. Under data
     * parallelism the all-reduced gradient already is the global sum, so the divisor stays 1; only the momentum
     * carry is split over the ranks like everywhere else (bcnn_node_sgd_step). */
    bcnn_activation_param *p = (bcnn_activation_param *)node->param;
    if (p->activation != BCNN_ACT_PRELU) return;
    bcnn_tensor *slopes = &net->tensors[node->src[1]];
    if (!slopes->data_gpu || !slopes->grad_data_gpu) return;
    const bcnn_learner *ln = net->learner;
    bcnn_hip_sgd_update(slopes->data_gpu, NULL, slopes->grad_data_gpu, NULL, (size_t)bcnn_tensor_size(slopes), 0,
                        slopes->n, ln->learning_rate, ln->momentum / (float)hctx(net)->dp_world, ln->decay);
}
