/*
 * bcnn_core.c -- graph runtime of the MI355X build: net / node / tensor bookkeeping, the three
 * executor loops, the learner, and the device-mirror sync points. C99; every device action goes
 * through the C-ABI of include/bcnn_hip.h (no HIP types here).
 *
 * Behaviour follows the reference runtime (jnbraun/bcnn):
 *   net life cycle   src/bcnn_net.c:61-157, 236-258, 280-285, 337-359
 *   executor         src/bcnn_net.c:361-375 (dst-gradient reset), 410-429, 452-481
 *   tensors          src/bcnn_tensor.c:40-145
 *   learner          src/bcnn_learner.c:29-65, 167-217
 * What is new: parameters and their gradients are re-packed into two contiguous device arenas at
 * compile time (one all-reduce per step, one place to checkpoint), and the SGD step knows the
 * data-parallel world size.
 */
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "bcnn_internal.h"
#include "../../include/bcnn_hip.h"

#ifndef BCNN_USE_HIP
#error "this runtime is the device build: compile with -DBCNN_USE_HIP"
#endif

/* ------------------------------------------------------------------------------------------------
 * logging
 * ---------------------------------------------------------------------------------------------- */
void bcnn_log(bcnn_log_context ctx, bcnn_log_level level, const char *fmt, ...) {
    if (level < ctx.lvl) return;
    char msg[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    if (ctx.fct) {
        ctx.fct("%s", msg);
        return;
    }
    static const char *tag[] = {"[INFO] ", "[WARNING] ", "[ERROR] ", ""};
    fprintf(stderr, "%s%s", tag[level > 3 ? 3 : level], msg);
}

void bcnn_set_log_context(bcnn_net *net, bcnn_log_callback fct, bcnn_log_level level) {
    net->log_ctx.fct = fct;
    net->log_ctx.lvl = level;
}

/* Box-Muller pair generator on rand(), as the MSRA filler of the reference uses (bcnn_utils.c:48-73) */
float bcnn_rng_gaussian(bcnn_gauss_gen *g) {
    if (g->state) {
        g->state = 0;
        return g->r;
    }
    float u, v, s;
    do {
        u = 2.0f * ((float)rand() / RAND_MAX) - 1.0f;
        v = 2.0f * ((float)rand() / RAND_MAX) - 1.0f;
        s = u * u + v * v;
    } while (s >= 1.0f || s == 0.0f);
    const float m = sqrtf(-2.0f * logf(s) / s);
    g->r = v * m;
    g->state = 1;
    return u * m;
}

/* ------------------------------------------------------------------------------------------------
 * tensors: 32-byte aligned zeroed host buffer + device mirror
 * ---------------------------------------------------------------------------------------------- */
static float *host_calloc_aligned(size_t count) {
    void *p = NULL;
    if (count == 0) return NULL;
    if (posix_memalign(&p, 32, count * sizeof(float)) != 0) return NULL;
    memset(p, 0, count * sizeof(float));
    return (float *)p;
}

int bcnn_tensor_size(const bcnn_tensor *t) { return t->w * t->h * t->c * t->n; }
int bcnn_tensor_size3d(const bcnn_tensor *t) { return t->w * t->h * t->c; }
int bcnn_tensor_size2d(const bcnn_tensor *t) { return t->w * t->h; }

void bcnn_tensor_set_shape(bcnn_tensor *t, int n, int c, int h, int w, int has_grad) {
    t->n = n; t->c = c; t->h = h; t->w = w; t->has_grad = has_grad;
}

void bcnn_tensor_free(bcnn_tensor *t) {
    free(t->data); t->data = NULL;
    free(t->grad_data); t->grad_data = NULL;
    bcnn_hip_free(t->data_gpu); t->data_gpu = NULL;
    bcnn_hip_free(t->grad_data_gpu); t->grad_data_gpu = NULL;
}

bcnn_status bcnn_tensor_allocate_buffer(bcnn_tensor *t, int net_state, size_t size) {
    bcnn_tensor_free(t);
    if (size == 0) return BCNN_INVALID_PARAMETER;
    t->data = host_calloc_aligned(size);
    if (!t->data) return BCNN_FAILED_ALLOC;
    t->data_gpu = bcnn_hip_malloc_f32(size); /* zero-filled, like the freshly calloc'ed host copy */
    if (t->has_grad && net_state != BCNN_MODE_PREDICT) {
        t->grad_data = host_calloc_aligned(size);
        if (!t->grad_data) return BCNN_FAILED_ALLOC;
        t->grad_data_gpu = bcnn_hip_malloc_f32(size);
    }
    return BCNN_SUCCESS;
}

bcnn_status bcnn_tensor_allocate(bcnn_tensor *t, int net_state) {
    return bcnn_tensor_allocate_buffer(t, net_state, (size_t)t->n * t->c * t->h * t->w);
}

static char *dup_name(const char *s) {
    size_t n = strlen(s) + 1;
    char *d = (char *)malloc(n);
    if (d) memcpy(d, s, n);
    return d;
}

void bcnn_tensor_create(bcnn_tensor *t, int n, int c, int h, int w, int has_grad, const char *name,
                        int net_state) {
    bcnn_tensor_set_shape(t, n, c, h, w, has_grad);
    bcnn_tensor_allocate(t, net_state);
    free(t->name);
    t->name = dup_name(name);
}

/* Xavier: sqrt(3/range)*U(-1,1) from rand(); MSRA: sqrt(2/range)*N(0,1); then host -> device */
void bcnn_tensor_fill(bcnn_tensor *t, bcnn_tensor_filler filler) {
    if (!t->data) return;
    const int sz = bcnn_tensor_size(t);
    if (filler.type == BCNN_FILLER_XAVIER) {
        const float a = sqrtf(3.0f / filler.range);
        for (int i = 0; i < sz; ++i) t->data[i] = a * (2 * ((float)rand() / RAND_MAX) - 1);
    } else if (filler.type == BCNN_FILLER_MSRA) {
        const float a = sqrtf(2.0f / filler.range);
        bcnn_gauss_gen g = {0};
        for (int i = 0; i < sz; ++i) t->data[i] = a * bcnn_rng_gaussian(&g);
    } else {
        for (int i = 0; i < sz; ++i) t->data[i] = filler.value;
    }
    if (t->data_gpu) bcnn_hip_memcpy_h2d(t->data_gpu, t->data, (size_t)sz * sizeof(float));
}

void bcnn_tensor_destroy(bcnn_tensor *t) {
    bcnn_tensor_free(t);
    bcnn_tensor_set_shape(t, 0, 0, 0, 0, 0);
    free(t->name);
    t->name = NULL;
}

/* ------------------------------------------------------------------------------------------------
 * net / node containers (realloc-grown arrays: indices are stable, pointers are not)
 * ---------------------------------------------------------------------------------------------- */
static bcnn_hip_context *hctx(bcnn_net *net) { return (bcnn_hip_context *)net->hip_ctx; }

bcnn_status bcnn_net_add_tensor(bcnn_net *net, bcnn_tensor tensor) {
    bcnn_tensor *p = (bcnn_tensor *)realloc(net->tensors, (size_t)(net->num_tensors + 1) * sizeof(bcnn_tensor));
    if (!p) return BCNN_FAILED_ALLOC;
    net->tensors = p;
    net->tensors[net->num_tensors++] = tensor;
    return BCNN_SUCCESS;
}

/* The cached SGD table (bcnn_update) names device buffers: any change of the node list or a re-pack of the
 * arenas makes it stale. */
static void invalidate_sgd_table(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    if (!hc) return;
    if (hc->sgd_chunks_gpu) {
        bcnn_hip_sync();
        bcnn_hip_free(hc->sgd_chunks_gpu);
        hc->sgd_chunks_gpu = NULL;
    }
    hc->num_sgd_chunks = 0;
    if (hc->fill_chunks_gpu) {
        bcnn_hip_sync();
        bcnn_hip_free(hc->fill_chunks_gpu);
        hc->fill_chunks_gpu = NULL;
    }
    hc->num_fill_chunks = -1; /* not built */
}

static void relink_graph(bcnn_net *net);

bcnn_status bcnn_net_add_node(bcnn_net *net, bcnn_node node) {
    bcnn_hip_context *hc = hctx(net);
    /* A compiled net carries node-to-node links that were proven on the OLD graph (a convolution node that stops after its
     * statistics because the eltwise / pooling / depthwise node behind it is its only reader, fused backward pairs, ...).
     * The new node may read one of those unwritten tensors or be a second writer of a gradient. Values a fused pass left
     * pending are produced now, under the old links; the links are re-derived below on the graph with the node in it. */
    if (hc && hc->compiled) {
        bcnn_materialize_data(net, -1);
        bcnn_materialize_gradients(net, -1);
    }
    bcnn_node *p = (bcnn_node *)realloc(net->nodes, (size_t)(net->num_nodes + 1) * sizeof(bcnn_node));
    if (!p) return BCNN_FAILED_ALLOC;
    net->nodes = p;
    net->nodes[net->num_nodes++] = node;
    invalidate_sgd_table(net); /* the new node's parameters are not in the cached one-launch table */
    if (hc) { /* per-node arena offsets are rebuilt by the next bcnn_compile_net; until then one final range */
        free(hc->node_grad_first);
        hc->node_grad_first = NULL;
        if (hc->compiled) relink_graph(net); /* dead-fill / sole-writer marks and fusion links: functions of the graph */
        else hc->grad_fill_count = 0;
    }
    return BCNN_SUCCESS;
}

static bcnn_status push_index(int **arr, int *count, int index) {
    int *p = (int *)realloc(*arr, (size_t)(*count + 1) * sizeof(int));
    if (!p) return BCNN_FAILED_ALLOC;
    p[(*count)++] = index;
    *arr = p;
    return BCNN_SUCCESS;
}

bcnn_status bcnn_node_add_input(bcnn_net *net, bcnn_node *node, int index) {
    (void)net;
    return push_index(&node->src, &node->num_src, index);
}

bcnn_status bcnn_node_add_output(bcnn_net *net, bcnn_node *node, int index) {
    (void)net;
    return push_index(&node->dst, &node->num_dst, index);
}

int bcnn_net_find_tensor(bcnn_net *net, const char *name) {
    for (int i = net->num_tensors - 1; i >= 0; --i)
        if (net->tensors[i].name && strcmp(net->tensors[i].name, name) == 0) return i;
    return -1;
}

void bcnn_net_register_param(bcnn_net *net, int index) {
    bcnn_hip_context *hc = hctx(net);
    push_index(&hc->param_ids, &hc->num_params, index);
}

bcnn_status bcnn_init_net(bcnn_net **net, bcnn_mode mode) {
    bcnn_net *p = (bcnn_net *)calloc(1, sizeof(bcnn_net));
    if (!p) return BCNN_FAILED_ALLOC;
    p->mode = mode;
    p->hip_ctx = calloc(1, sizeof(bcnn_hip_context));
    if (!p->hip_ctx) { free(p); return BCNN_FAILED_ALLOC; }
    hctx(p)->dp_world = 1;
    hctx(p)->num_fill_chunks = -1;
    /* tensor 0 = "input", tensor 1 = "label" (reference bcnn_net.c:66-76) */
    bcnn_tensor in = {0}, lab = {0};
    in.name = dup_name("input");
    lab.name = dup_name("label");
    bcnn_net_add_tensor(p, in);
    bcnn_net_add_tensor(p, lab);
    if (mode != BCNN_MODE_PREDICT) {
        p->learner = (bcnn_learner *)calloc(1, sizeof(bcnn_learner));
        p->data_aug = (bcnn_data_augmenter *)calloc(1, sizeof(bcnn_data_augmenter));
    }
    p->num_inputs = 1;
    p->inputs = (int *)calloc(1, sizeof(int));
    p->num_threads = 1;
    *net = p;
    return BCNN_SUCCESS;
}

void bcnn_end_net(bcnn_net **pnet) {
    bcnn_net *net = *pnet;
    if (!net) return;
    bcnn_hip_sync();
    bcnn_hip_conv_prepack_reset(); /* the packed copies of this net's filter banks */
    bcnn_hip_context *hc = hctx(net);
    if (hc->comm_active) bcnn_hip_comm_destroy();
    /* arena members do not own their device buffers */
    if (hc->compiled) {
        for (int i = 0; i < hc->arena_members; ++i) { /* parameters added after the last compile still own theirs */
            bcnn_tensor *t = &net->tensors[hc->param_ids[i]];
            t->data_gpu = NULL;
            t->grad_data_gpu = NULL;
        }
        bcnn_hip_free(hc->param_arena_gpu);
        bcnn_hip_free(hc->grad_arena_gpu);
        bcnn_hip_free(hc->sgd_chunks_gpu);
        bcnn_hip_free(hc->fill_chunks_gpu);
        free(hc->sgd_chunks_host);
        free(hc->grad_fill_dead);
        free(hc->node_grad_first);
    }
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->release_param) nd->release_param(nd);
        free(nd->src); free(nd->dst); free(nd->param);
    }
    free(net->nodes);
    for (int i = 0; i < net->num_tensors; ++i) bcnn_tensor_destroy(&net->tensors[i]);
    free(net->tensors);
    bcnn_hip_free(hc->workspace_gpu);
    free(hc->param_ids);
    free(hc);
    free(net->learner); free(net->data_aug); bcnn_destroy_data_loader(net); free(net->inputs);
    free(net);
    *pnet = NULL;
}

bcnn_status bcnn_set_num_threads(bcnn_net *net, int num_threads, const int *cpu_ids) {
    (void)cpu_ids;  /* host threads do no math in this build */
    net->num_threads = num_threads > 0 ? num_threads : 1;
    return BCNN_SUCCESS;
}
int bcnn_get_num_threads(bcnn_net *net) { return net->num_threads; }
int bcnn_get_batch_size(bcnn_net *net) { return net->batch_size; }

void bcnn_set_input_shape(bcnn_net *net, int width, int height, int channels, int batch_size) {
    net->batch_size = batch_size;
    bcnn_tensor_set_shape(&net->tensors[0], batch_size, channels, height, width, 0);
}

bcnn_status bcnn_add_input(bcnn_net *net, int width, int height, int channels, const char *name) {
    bcnn_tensor t = {0};
    bcnn_tensor_set_shape(&t, net->batch_size, channels, height, width, 0);
    BCNN_CHECK_STATUS(bcnn_tensor_allocate(&t, net->mode));
    t.name = dup_name(name);
    BCNN_CHECK_STATUS(bcnn_net_add_tensor(net, t));
    int *p = (int *)realloc(net->inputs, (size_t)(net->num_inputs + 1) * sizeof(int));
    if (!p) return BCNN_FAILED_ALLOC;
    net->inputs = p;
    net->inputs[net->num_inputs++] = net->num_tensors - 1;
    return BCNN_SUCCESS;
}

/* ------------------------------------------------------------------------------------------------
 * compile: input tensor, shared conv workspace, parameter / gradient arenas
 * ---------------------------------------------------------------------------------------------- */
static void build_arenas(bcnn_net *net) {
    /* (Re)pack every registered parameter and its gradient into two contiguous device arenas. Runs at the first
     * compile and again whenever parameters were registered since (nodes added to a compiled net): members that
     * already live in the old arenas are copied across, the others give up their own allocation. */
    bcnn_hip_context *hc = hctx(net);
    size_t total = 0;
    for (int i = 0; i < hc->num_params; ++i) {
        /* keep every member 16-byte aligned for the vectorised kernels */
        total += ((size_t)bcnn_tensor_size(&net->tensors[hc->param_ids[i]]) + 3) & ~(size_t)3;
    }
    if (total == 0) return;
    float *old_p = hc->param_arena_gpu, *old_g = hc->grad_arena_gpu;
    float *new_p = bcnn_hip_malloc_f32(total);
    float *new_g = (net->mode != BCNN_MODE_PREDICT) ? bcnn_hip_malloc_f32(total) : NULL;
    size_t off = 0;
    for (int i = 0; i < hc->num_params; ++i) {
        bcnn_tensor *t = &net->tensors[hc->param_ids[i]];
        const size_t sz = (size_t)bcnn_tensor_size(t);
        const int member = i < hc->arena_members;
        bcnn_hip_memcpy_d2d(new_p + off, t->data_gpu, sz * sizeof(float));
        bcnn_hip_sync();
        if (!member) bcnn_hip_free(t->data_gpu);
        t->data_gpu = new_p + off;
        if (new_g && t->grad_data_gpu) {
            bcnn_hip_memcpy_d2d(new_g + off, t->grad_data_gpu, sz * sizeof(float));
            bcnn_hip_sync();
            if (!member) bcnn_hip_free(t->grad_data_gpu);
            t->grad_data_gpu = new_g + off;
        }
        off += (sz + 3) & ~(size_t)3;
    }
    bcnn_hip_free(old_p);
    bcnn_hip_free(old_g);
    hc->param_arena_gpu = new_p;
    hc->grad_arena_gpu = new_g;
    hc->arena_size = total;
    hc->arena_members = hc->num_params;
}

/* The reference zero-fills every dst gradient before the node's forward (bcnn_net.c:361-375) because
 * pooling / eltwise / fc / softmax backward accumulate. Backward visits a tensor's consumers in reverse
 * node order, so the consumer with the LOWEST index writes the gradient last; when that consumer is a
 * convolution (or a stand-alone batch-norm) its data-gradient pass overwrites every element (col2im
 * zero-fills first, conv_layer.c:571; batch-norm assigns, batchnorm_layer.c:292-296)
 * and whatever the fill and the earlier `+=` left there is never read. Such fills are skipped -- same
 * values everywhere a reader can see, ~0.65 GB less memset traffic per ResNet-18 step. */
static void mark_dead_grad_fills(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    free(hc->grad_fill_dead);
    hc->grad_fill_dead = (unsigned char *)calloc((size_t)net->num_tensors + 1, 1);
    hc->grad_fill_count = net->num_tensors;
    if (BCNN_EXP_ENV("BCNN_KEEP_ALL_GRAD_FILLS")) return; /* debugging switch: zero every dst gradient like the reference */
    for (int t = 0; t < net->num_tensors; ++t) {
        int first = -1, uses = 0;
        for (int i = 0; i < net->num_nodes; ++i)
            for (int k = 0; k < net->nodes[i].num_src; ++k)
                if (net->nodes[i].src[k] == t) {
                    if (first < 0) first = i;
                    ++uses;
                }
        if (first < 0) continue;
        const bcnn_node *nd = &net->nodes[first];
        if (nd->src[0] != t) continue;
        /* A gradient with exactly ONE writer that touches every element once per backward pass (max-pooling's
         * gather; the full-size operand of a same-shape eltwise add) needs no fill either if that writer assigns
         * `0 + sum` instead of accumulating: mark 2, the node's backward asks bcnn_grad_sole_writer(). Holds for
         * the forward -> backward order every caller of the reference uses (bcnn_train_on_batch). */
        if (uses == 1 && (nd->type == BCNN_LAYER_MAXPOOL || nd->type == BCNN_LAYER_DEPTHWISE_CONV2D)) {
            /* max-pooling's gather and the depthwise data gradient (a gather too: one thread owns a dx element,
             * bcnn_depthwise_conv_layer.c:432-547 accumulates onto the zero fill) */
            hc->grad_fill_dead[t] = 2;
            continue;
        }
        if (uses == 1 && nd->type == BCNN_LAYER_ELTWISE) {
            const bcnn_eltwise_param *ep = (const bcnn_eltwise_param *)nd->param;
            if (ep->stride[0] == 1 && ep->stride[1] == 1 &&
                bcnn_tensor_size(&net->tensors[t]) == bcnn_tensor_size(&net->tensors[nd->dst[0]]))
                hc->grad_fill_dead[t] = 2;
            continue;
        }
        if (nd->type == BCNN_LAYER_BATCHNORM) { /* stand-alone batch-norm writes dx = f(dy, x) everywhere (:292-296) */
            hc->grad_fill_dead[t] = 1;
            continue;
        }
        if (nd->type != BCNN_LAYER_CONV2D) continue;
        const bcnn_conv_param *p = (const bcnn_conv_param *)nd->param;
        const bcnn_tensor *x = &net->tensors[t], *y = &net->tensors[nd->dst[0]];
        /* 1x1 kernels write dX through the raw [C/g][OH*OW] view: only complete when OH*OW == H*W */
        if (p->size == 1 && y->h * y->w != x->h * x->w) continue;
        hc->grad_fill_dead[t] = 1;
    }
}

int bcnn_grad_sole_writer(bcnn_net *net, int tensor) {
    const bcnn_hip_context *hc = hctx(net);
    return hc->grad_fill_dead && tensor >= 0 && tensor < hc->grad_fill_count && hc->grad_fill_dead[tensor] == 2;
}

static void comm_sync_parameters(bcnn_net *net);

bcnn_status bcnn_compile_net(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    /* (re)allocate the input tensor, as bcnn_init_workload does (reference bcnn_net.c:337-359) */
    BCNN_CHECK_STATUS(bcnn_tensor_allocate(&net->tensors[0], net->mode));
    /* one conv scratch for the whole net: the largest any conv node needs */
    size_t need = 0;
    for (int i = 0; i < net->num_nodes; ++i) {
        if (net->nodes[i].type != BCNN_LAYER_CONV2D) continue;
        bcnn_conv_param *p = (bcnn_conv_param *)net->nodes[i].param;
        const bcnn_tensor *s = &net->tensors[net->nodes[i].src[0]];
        const size_t ws = bcnn_hip_conv_workspace_size(s->n, s->c, s->h, s->w, p->num, p->size, p->stride, p->pad,
                                                       p->num_groups);
        if (ws > need) need = ws;
    }
    if (need > hc->workspace_size) {
        bcnn_hip_free(hc->workspace_gpu);
        hc->workspace_gpu = bcnn_hip_malloc_f32(need);
        hc->workspace_size = need;
    }
    for (int i = 0; i < net->num_nodes; ++i)
        if (net->nodes[i].type == BCNN_LAYER_CONV2D)
            ((bcnn_conv_param *)net->nodes[i].param)->conv_workspace_gpu = hc->workspace_gpu;
    if (!hc->compiled || hc->num_params != hc->arena_members) {
        /* first compile, or nodes were added since the last one: (re)pack the arenas; the gradient-ready offsets
         * and the cached SGD table refer to the old layout */
        invalidate_sgd_table(net);
        build_arenas(net);
        hc->compiled = 1;
        if (hc->grad_ready_fn) bcnn_set_gradient_ready_callback(net, hc->grad_ready_fn, hc->grad_ready_user);
        comm_sync_parameters(net); /* communicator installed before the arena existed / new parameters appeared */
    }
    relink_graph(net);
    bcnn_hip_sync();
    return BCNN_SUCCESS;
}

/* everything bcnn_compile_net derives from the graph alone: which gradient fills are dead, who is a sole writer, and the
 * node-to-node fusion links (each link pass starts from "no link"). Also run by bcnn_net_add_node on a compiled net. */
static void relink_graph(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    mark_dead_grad_fills(net);
    bcnn_link_depthwise_batchnorm(net);
    bcnn_link_conv_eltwise(net);
    bcnn_link_conv_maxpool(net);
    bcnn_link_conv_depthwise(net);
    bcnn_link_batchnorm_conv(net);
    if (hc->fill_chunks_gpu) { /* the table of live fills follows the dead-fill marks */
        bcnn_hip_sync();
        bcnn_hip_free(hc->fill_chunks_gpu);
        hc->fill_chunks_gpu = NULL;
    }
    hc->num_fill_chunks = -1;
}

/* Reference bcnn_net.c:287-335: new input extent (batch 1, like the reference's bcnn_set_input_shape(net, w, h, c, 1)), then
 * the first destination tensor of every node re-shaped from its first source -- convolution and max-pooling by their output
 * rules, every other node as a copy of the source's shape -- and, with need_realloc, re-allocated (host and device mirrors).
 * In the reference, layer-private buffers (batch-norm workspaces, pooling indexes) keep the size they were built with, so a
 * net resized to a LARGER extent -- the usual detection use of this call -- overruns them (there: host heap; here it
 * would be device memory). Deliberate deviation: with need_realloc the private buffers whose size follows a dst tensor
 * (pooling indexes and kept maxima, the batch-norm workspaces of convolution and batch-norm nodes) are re-allocated for the
 * new shape as well; without it, a shape that outgrew them is refused (BCNN_INVALID_PARAMETER) instead of corrupting memory.
 * What this build derives from shapes (the conv scratch, the node-to-node links and their sums buffers) is brought up to
 * date before returning; the reference's function falls off its end without a return value, here the status is BCNN_SUCCESS. */
static bcnn_status resize_private_f32(float **buf, size_t old_elems, size_t new_elems, int need_realloc) {
    if (!*buf || new_elems <= old_elems) return BCNN_SUCCESS;
    if (!need_realloc) return BCNN_INVALID_PARAMETER;
    bcnn_hip_sync();
    bcnn_hip_free(*buf);
    *buf = bcnn_hip_malloc_f32(new_elems);
    return *buf ? BCNN_SUCCESS : BCNN_FAILED_ALLOC;
}

bcnn_status bcnn_resize_net(bcnn_net *net, int w, int h, int c, int need_realloc) {
    if (!net || w <= 0 || h <= 0 || c <= 0) return BCNN_INVALID_PARAMETER;
    bcnn_set_input_shape(net, w, h, c, 1);
    for (int i = 0; i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->num_src < 1 || nd->num_dst < 1) continue;
        const bcnn_tensor *s = &net->tensors[nd->src[0]];
        bcnn_tensor *d = &net->tensors[nd->dst[0]];
        const size_t old_elems = (size_t)bcnn_tensor_size(d);
        if (nd->type == BCNN_LAYER_CONV2D) {
            const bcnn_conv_param *p = (const bcnn_conv_param *)nd->param;
            bcnn_tensor_set_shape(d, s->n, p->num, (s->h + 2 * p->pad - p->size) / p->stride + 1,
                                  (s->w + 2 * p->pad - p->size) / p->stride + 1, 1);
        } else if (nd->type == BCNN_LAYER_MAXPOOL) {
            const bcnn_maxpool_param *p = (const bcnn_maxpool_param *)nd->param;
            bcnn_tensor_set_shape(d, s->n, s->c, (s->h - 1) / p->stride + 1, (s->w - 1) / p->stride + 1, 1);
        } else {
            bcnn_tensor_set_shape(d, s->n, s->c, s->h, s->w, 1);
        }
        if (need_realloc) BCNN_CHECK_STATUS(bcnn_tensor_allocate(d, net->mode));
        const size_t new_elems = (size_t)bcnn_tensor_size(d);
        if (nd->type == BCNN_LAYER_CONV2D) {
            bcnn_conv_param *p = (bcnn_conv_param *)nd->param;
            BCNN_CHECK_STATUS(resize_private_f32(&p->bn_workspace_gpu, old_elems, new_elems, need_realloc));
            BCNN_CHECK_STATUS(resize_private_f32(&p->x_norm_gpu, old_elems, new_elems, need_realloc));
        } else if (nd->type == BCNN_LAYER_BATCHNORM) {
            bcnn_batchnorm_param *p = (bcnn_batchnorm_param *)nd->param;
            BCNN_CHECK_STATUS(resize_private_f32(&p->workspace_gpu, old_elems, new_elems, need_realloc));
            BCNN_CHECK_STATUS(resize_private_f32(&p->x_norm_gpu, old_elems, new_elems, need_realloc));
        } else if (nd->type == BCNN_LAYER_MAXPOOL && new_elems > old_elems) {
            bcnn_maxpool_param *p = (bcnn_maxpool_param *)nd->param;
            if (!need_realloc) return BCNN_INVALID_PARAMETER;
            bcnn_hip_sync();
            bcnn_hip_free(p->indexes_gpu);
            p->indexes_gpu = bcnn_hip_malloc_i32(new_elems);
            int *host = (int *)realloc(p->indexes, new_elems * sizeof(int));
            if (!host || !p->indexes_gpu) return BCNN_FAILED_ALLOC;
            p->indexes = host;
            if (p->raw_at_max_gpu) { /* re-made by bcnn_link_conv_maxpool for the new shape */
                bcnn_hip_free(p->raw_at_max_gpu);
                p->raw_at_max_gpu = NULL;
            }
        }
    }
    if (hctx(net)->compiled) return bcnn_compile_net(net); /* input tensor, conv scratch, links */
    return BCNN_SUCCESS;
}

bcnn_status bcnn_set_mode(bcnn_net *net, bcnn_mode mode) {
    if (net->mode == mode) return BCNN_SUCCESS;
    net->mode = mode;
    /* TRAIN reads the train streams, VALID / PREDICT the (rewound) test streams: reference bcnn_net.c:490-504 */
    if (net->data_loader) bcnn_switch_data_handles(net, net->data_loader);
    return BCNN_SUCCESS;
}

/* ------------------------------------------------------------------------------------------------
 * executor loops
 * ---------------------------------------------------------------------------------------------- */
/* Every dst gradient is zero-filled before its node runs (reference bcnn_net.c:361-375): pooling / eltwise / fc /
 * softmax backward ACCUMULATE into it. No forward worker reads a gradient, so the fills that are not provably dead
 * (mark_dead_grad_fills) are issued together, as ONE launch over a table, at the start of the pass. */
static void build_fill_table(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    int cap = 256, n = 0;
    bcnn_hip_fill_chunk *tab = (bcnn_hip_fill_chunk *)malloc((size_t)cap * sizeof(*tab));
    unsigned char *seen = (unsigned char *)calloc((size_t)net->num_tensors + 1, 1);
    for (int i = 0; i < net->num_nodes; ++i)
        for (int d = 0; d < net->nodes[i].num_dst; ++d) {
            const int id = net->nodes[i].dst[d];
            bcnn_tensor *t = &net->tensors[id];
            if (seen[id] || !t->grad_data_gpu) continue;
            seen[id] = 1;
            if (hc->grad_fill_dead && id < hc->grad_fill_count && hc->grad_fill_dead[id]) continue;
            const size_t sz = (size_t)bcnn_tensor_size(t);
            for (size_t off = 0; off < sz; off += BCNN_HIP_FILL_CHUNK) {
                if (n == cap) {
                    cap *= 2;
                    tab = (bcnn_hip_fill_chunk *)realloc(tab, (size_t)cap * sizeof(*tab));
                }
                tab[n].p_d = t->grad_data_gpu + off;
                tab[n].count = (unsigned int)((sz - off < BCNN_HIP_FILL_CHUNK) ? (sz - off) : BCNN_HIP_FILL_CHUNK);
                tab[n].reserved = 0;
                ++n;
            }
        }
    free(seen);
    hc->num_fill_chunks = n;
    if (n > 0) {
        hc->fill_chunks_gpu = bcnn_hip_malloc_f32(((size_t)n * sizeof(*tab) + 3) / 4);
        bcnn_hip_memcpy_h2d(hc->fill_chunks_gpu, tab, (size_t)n * sizeof(*tab));
    }
    free(tab);
}

void bcnn_forward(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    if (net->mode == BCNN_MODE_TRAIN) {
        if (hc->num_fill_chunks < 0) build_fill_table(net);
        bcnn_hip_zero_chunks((const bcnn_hip_fill_chunk *)hc->fill_chunks_gpu, hc->num_fill_chunks);
    }
    bcnn_drop_pending_gradients(net);
    bcnn_prepack_conv_weights(net, 0);
    hc->in_pass = 1;
    for (int i = 0; i < net->num_nodes; ++i) net->nodes[i].forward(net, &net->nodes[i]);
    hc->in_pass = 0;
    bcnn_hip_conv_prepack_discard(); /* a copy no node of this pass consumed must not meet later (rewritten) weights */
}

/* in-library data parallelism: the gradient-ready callback of bcnn_set_data_parallel_comm */
static void comm_flush(bcnn_hip_context *hc) {
    if (hc->comm_lo < hc->comm_hi) {
        bcnn_hip_allreduce_sum(hc->grad_arena_gpu + hc->comm_lo, hc->comm_hi - hc->comm_lo);
        hc->comm_hi = hc->comm_lo;
    }
}

static void comm_on_ready(size_t first, size_t count, void *user) {
    bcnn_hip_context *hc = (bcnn_hip_context *)user;
    (void)count; /* ranges arrive as a growing tail: [first, comm_lo) is new */
    hc->comm_lo = first;
    if (hc->comm_hi - hc->comm_lo >= hc->comm_bucket) comm_flush(hc);
}

void bcnn_backward(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    size_t ready_from = hc->arena_size;
    if (hc->comm_active) hc->comm_lo = hc->comm_hi = hc->arena_size;
    bcnn_prepack_conv_weights(net, 1);
    hc->in_pass = 2;
    /* weight gradients on a side stream, next to the sweeps and data gradients of the layers in front (joined below, and
     * before anybody is told that gradients are final) */
    const int side_prev = bcnn_hip_conv_side_stream_mode((hc->no_side_stream || BCNN_EXP_ENV("BCNN_NO_SIDE_STREAM")) ? 0 : 2);
    for (int i = net->num_nodes - 1; i >= 0; --i) {
        net->nodes[i].backward(net, &net->nodes[i]);
        if (hc->grad_ready_fn && hc->node_grad_first && hc->node_grad_first[i] < ready_from) {
            bcnn_hip_conv_side_join();
            hc->grad_ready_fn(hc->node_grad_first[i], ready_from - hc->node_grad_first[i], hc->grad_ready_user);
            ready_from = hc->node_grad_first[i];
        }
    }
    hc->in_pass = 0;
    bcnn_hip_conv_side_join();
    bcnn_hip_conv_side_stream_mode(side_prev);
    bcnn_hip_conv_prepack_discard(); /* before bcnn_update rewrites the weights an unused copy was made from */
    if (hc->grad_ready_fn && ready_from > 0 && hc->arena_size > 0)  /* members no node claims (none today) */
        hc->grad_ready_fn(0, ready_from, hc->grad_ready_user);
    if (hc->comm_active) {
        comm_flush(hc);       /* what is left below the last full bucket */
        bcnn_hip_comm_join(); /* bcnn_update (same stream) is ordered behind every bucket; the host does not block */
    }
}

void bcnn_set_weight_gradient_stream(bcnn_net *net, int enable) { hctx(net)->no_side_stream = enable ? 0 : 1; }

void bcnn_set_gradient_ready_callback(bcnn_net *net, bcnn_gradient_ready_fn fn, void *user) {
    bcnn_hip_context *hc = hctx(net);
    hc->grad_ready_fn = fn;
    hc->grad_ready_user = user;
    free(hc->node_grad_first);
    hc->node_grad_first = NULL;
    if (!fn || !hc->grad_arena_gpu) return;
    hc->node_grad_first = (size_t *)malloc((size_t)net->num_nodes * sizeof(size_t));
    for (int i = 0; i < net->num_nodes; ++i) {
        size_t first = (size_t)-1;
        for (int k = 0; k < net->nodes[i].num_src; ++k) {
            const bcnn_tensor *t = &net->tensors[net->nodes[i].src[k]];
            if (!t->grad_data_gpu || t->grad_data_gpu < hc->grad_arena_gpu ||
                t->grad_data_gpu >= hc->grad_arena_gpu + hc->arena_size)
                continue;
            const size_t off = (size_t)(t->grad_data_gpu - hc->grad_arena_gpu);
            if (off < first) first = off;
        }
        hc->node_grad_first[i] = first;
    }
}

/* learning-rate schedules, reference bcnn_learner.c:29-65 */
static void step_learning_rate(bcnn_net *net) {
    bcnn_learner *ln = net->learner;
    /* `seen` counts SAMPLES (bcnn_learner.c:31-33) and keys both the schedules and Adam's bias correction
     * (bcnn_learner.c:111-112): a data-parallel step consumes the global batch, exactly like the single-process
     * run on that batch which it has to reproduce */
    const int global_batch = net->batch_size * hctx(net)->dp_world;
    ln->seen += global_batch;
    const int iter = ln->seen / global_batch;
    switch (ln->decay_type) {
        case BCNN_LR_DECAY_STEP:
            ln->learning_rate = ln->base_learning_rate * (float)pow(ln->scale, iter / ln->step);
            break;
        case BCNN_LR_DECAY_INV:
            ln->learning_rate = ln->base_learning_rate * (float)pow(1.0f + ln->gamma * iter, -ln->power);
            break;
        case BCNN_LR_DECAY_EXP:
            ln->learning_rate = ln->base_learning_rate * (float)pow(ln->gamma, iter);
            break;
        case BCNN_LR_DECAY_POLY:
            ln->learning_rate = ln->base_learning_rate * (float)pow(1 - (float)iter / ln->max_batches, ln->power);
            break;
        case BCNN_LR_DECAY_SIGMOID:
            ln->learning_rate = ln->base_learning_rate * (1.0f / (1.0f + (float)exp(ln->gamma * (iter - ln->step))));
            break;
        default:
            break;
    }
}

/* The reference walks the nodes and issues one SGD step per node (bcnn_net.c:316-326). Here the first
 * call only RECORDS which buffers the nodes' update() workers step (and with which rule); the table goes
 * to the device once and every update is then a single launch over all of them. */
static void sgd_table_add(bcnn_hip_context *hc, float *w, float *g, size_t n, int use_decay) {
    for (size_t off = 0; off < n; off += BCNN_HIP_SGD_CHUNK) {
        if (hc->num_sgd_chunks == hc->cap_sgd_chunks) {
            hc->cap_sgd_chunks = hc->cap_sgd_chunks ? 2 * hc->cap_sgd_chunks : 1024;
            hc->sgd_chunks_host = (bcnn_hip_sgd_chunk *)realloc(hc->sgd_chunks_host,
                                                                (size_t)hc->cap_sgd_chunks * sizeof(bcnn_hip_sgd_chunk));
        }
        bcnn_hip_sgd_chunk *c = &hc->sgd_chunks_host[hc->num_sgd_chunks++];
        c->w_d = w + off;
        c->g_d = g + off;
        c->count = (unsigned int)((n - off < BCNN_HIP_SGD_CHUNK) ? (n - off) : BCNN_HIP_SGD_CHUNK);
        c->use_decay = (unsigned int)use_decay;
    }
}

/* update workers whose whole effect is bcnn_node_sgd_step on their own tensors: safe to replay from the table */
static int update_is_tabled(const bcnn_node *nd) {
    return nd->update == bcnn_update_conv_layer || nd->update == bcnn_update_depthwise_conv_layer ||
           nd->update == bcnn_update_fullc_layer;
}

void bcnn_update(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    step_learning_rate(net);
    if (net->learner->optimizer == BCNN_OPTIM_ADAM) { /* per-node launches; no one-launch table */
        for (int i = 0; i < net->num_nodes; ++i)
            if (net->nodes[i].update) net->nodes[i].update(net, &net->nodes[i]);
        return;
    }
    if (hc->sgd_chunks_gpu == NULL) {
        hc->sgd_collecting = 1;
        hc->num_sgd_chunks = 0;
        for (int i = 0; i < net->num_nodes; ++i)
            if (net->nodes[i].update && update_is_tabled(&net->nodes[i])) net->nodes[i].update(net, &net->nodes[i]);
        hc->sgd_collecting = 0;
        if (hc->num_sgd_chunks > 0) {
            hc->sgd_chunks_gpu = bcnn_hip_malloc_f32(((size_t)hc->num_sgd_chunks * sizeof(bcnn_hip_sgd_chunk) + 3) / 4);
            bcnn_hip_memcpy_h2d(hc->sgd_chunks_gpu, hc->sgd_chunks_host,
                                (size_t)hc->num_sgd_chunks * sizeof(bcnn_hip_sgd_chunk));
        }
    }
    const bcnn_learner *ln = net->learner;
    if (hc->num_sgd_chunks > 0)
        bcnn_hip_sgd_update_chunks((const bcnn_hip_sgd_chunk *)hc->sgd_chunks_gpu, hc->num_sgd_chunks,
                                   net->batch_size * hc->dp_world, ln->learning_rate,
                                   ln->momentum / (float)hc->dp_world, ln->decay);
    /* every other worker runs on every step like in the reference's loop (bcnn_net.c:316-326): the stand-alone
     * PReLU node (its own batch divisor) and anything a user plugged into node->update */
    for (int i = 0; i < net->num_nodes; ++i)
        if (net->nodes[i].update && !update_is_tabled(&net->nodes[i])) net->nodes[i].update(net, &net->nodes[i]);
}

/* One fused pass per buffer (reference: axpy/axpy/scal sequence, bcnn_learner.c:67-104). Under data
 * parallelism the all-reduced gradient is a sum over `world` ranks of (fresh gradient + momentum
 * carry): the step divides by the GLOBAL batch and leaves momentum*g/world in the buffer, so the next
 * all-reduce reconstitutes exactly one carry ("momentum trap"). */
void bcnn_node_sgd_step(bcnn_net *net, bcnn_tensor *weights, bcnn_tensor *biases) {
    const bcnn_learner *ln = net->learner;
    const int world = hctx(net)->dp_world;
    if (hctx(net)->sgd_collecting) { /* bcnn_update is building its one-launch table */
        if (biases && biases->data_gpu && biases->grad_data_gpu)
            sgd_table_add(hctx(net), biases->data_gpu, biases->grad_data_gpu, (size_t)bcnn_tensor_size(biases), 0);
        if (weights && weights->data_gpu && weights->grad_data_gpu)
            sgd_table_add(hctx(net), weights->data_gpu, weights->grad_data_gpu, (size_t)bcnn_tensor_size(weights), 1);
        return;
    }
    bcnn_hip_sgd_update(weights ? weights->data_gpu : NULL, biases ? biases->data_gpu : NULL,
                        weights ? weights->grad_data_gpu : NULL, biases ? biases->grad_data_gpu : NULL,
                        weights ? (size_t)bcnn_tensor_size(weights) : 0, biases ? (size_t)bcnn_tensor_size(biases) : 0,
                        net->batch_size * world, ln->learning_rate, ln->momentum / (float)world, ln->decay);
}

/* Adam (reference bcnn_adam_update_cpu, bcnn_learner.c:106-131; reachable only through the INI key
 * `optimizer=adam`, quirk 6). The moment buffers belong to the node's param block like the reference's
 * adam_m_gpu / adam_v_gpu; they are created by the first step (the reference creates them in the builder
 * when the learner already says Adam, and crashes otherwise). Under data parallelism the weight gradient
 * carries nothing between steps (Adam zeroes it), so the all-reduced sum with the GLOBAL batch is the
 * single-process step; the bias carry is divided by `world` like in bcnn_node_sgd_step. */
void bcnn_node_optim_step(bcnn_net *net, bcnn_tensor *weights, bcnn_tensor *biases, float **adam_m_gpu,
                          float **adam_v_gpu) {
    const bcnn_learner *ln = net->learner;
    if (ln->optimizer != BCNN_OPTIM_ADAM || adam_m_gpu == NULL) {
        bcnn_node_sgd_step(net, weights, biases);
        return;
    }
    const int world = hctx(net)->dp_world;
    const size_t wsz = (size_t)bcnn_tensor_size(weights);
    if (*adam_m_gpu == NULL) {
        *adam_m_gpu = bcnn_hip_malloc_f32(wsz); /* zero-filled */
        *adam_v_gpu = bcnn_hip_malloc_f32(wsz);
    }
    bcnn_hip_adam_update(weights->data_gpu, biases ? biases->data_gpu : NULL, weights->grad_data_gpu,
                         biases ? biases->grad_data_gpu : NULL, *adam_m_gpu, *adam_v_gpu, wsz,
                         biases ? (size_t)bcnn_tensor_size(biases) : 0, net->batch_size * world, ln->seen, ln->beta1,
                         ln->beta2, ln->learning_rate, ln->momentum / (float)world, ln->decay);
}

static float current_loss(bcnn_net *net) {
    float loss = 0.f;
    int n = 0;
    for (int i = 0; i < net->num_nodes; ++i)
        if (net->nodes[i].type == BCNN_LAYER_COST) {
            loss += net->tensors[net->nodes[i].dst[0]].data[0];
            ++n;
        }
    return n ? loss / n : 0.f;
}

/* forward / backward / update return nothing (reference bcnn_net.c:455-488); a batch that could not be read is fatal in the
 * reference's convention for unrecoverable errors (print and exit, bcnn_utils.h:174-195), not a step on stale tensors */
static void next_batch_or_die(bcnn_net *net) {
    const bcnn_status st = bcnn_loader_next(net);
    if (st != BCNN_SUCCESS) {
        fprintf(stderr, "[bcnn] the data loader could not fill a batch (status %d)\n", (int)st);
        exit(1);
    }
}

float bcnn_train_on_batch(bcnn_net *net) {
    next_batch_or_die(net);
    bcnn_forward(net);
    bcnn_backward(net);
    bcnn_update(net);
    return current_loss(net);
}

float bcnn_predict_on_batch(bcnn_net *net, bcnn_tensor **out) {
    next_batch_or_die(net);
    bcnn_forward(net);
    const bcnn_node *last = &net->nodes[net->num_nodes - 1];
    const int out_id = (last->type == BCNN_LAYER_COST) ? last->src[0] : last->dst[0];
    bcnn_tensor *t = &net->tensors[out_id];
    bcnn_hip_memcpy_d2h(t->data, t->data_gpu, (size_t)bcnn_tensor_size(t) * sizeof(float));
    *out = t;
    return current_loss(net);
}

/* ------------------------------------------------------------------------------------------------
 * tensor access + sync points
 * ---------------------------------------------------------------------------------------------- */
int bcnn_get_tensor_index_by_name(bcnn_net *net, const char *name) { return bcnn_net_find_tensor(net, name); }

bcnn_tensor *bcnn_get_tensor_by_index(bcnn_net *net, int index) {
    if (index < 0 || index >= net->num_tensors) return NULL;
    bcnn_download_tensor(net, index, 1); /* refresh host data + grad (reference bcnn_net.c:393-401) */
    return &net->tensors[index];
}

bcnn_tensor *bcnn_get_tensor_by_name(bcnn_net *net, const char *name) {
    return bcnn_get_tensor_by_index(net, bcnn_net_find_tensor(net, name));
}

bcnn_status bcnn_upload_tensor(bcnn_net *net, int index, int with_grad) {
    if (index < 0 || index >= net->num_tensors) return BCNN_INVALID_PARAMETER;
    bcnn_tensor *t = &net->tensors[index];
    const size_t bytes = (size_t)bcnn_tensor_size(t) * sizeof(float);
    bcnn_materialize_data(net, index); /* the pending values would otherwise land on top of the caller's later */
    if (t->data && t->data_gpu) bcnn_hip_memcpy_h2d(t->data_gpu, t->data, bytes);
    if (with_grad) bcnn_materialize_gradients(net, -1); /* before a caller's values can mix with pending ones */
    if (with_grad && t->grad_data && t->grad_data_gpu) bcnn_hip_memcpy_h2d(t->grad_data_gpu, t->grad_data, bytes);
    return BCNN_SUCCESS;
}

bcnn_status bcnn_download_tensor(bcnn_net *net, int index, int with_grad) {
    if (index < 0 || index >= net->num_tensors) return BCNN_INVALID_PARAMETER;
    bcnn_tensor *t = &net->tensors[index];
    const size_t bytes = (size_t)bcnn_tensor_size(t) * sizeof(float);
    bcnn_materialize_data(net, index); /* an output tensor a fused forward pass did not need to write */
    if (t->data && t->data_gpu) bcnn_hip_memcpy_d2h(t->data, t->data_gpu, bytes);
    if (with_grad) bcnn_materialize_gradients(net, index); /* gradients a fused backward pass did not need to write */
    if (with_grad && t->grad_data && t->grad_data_gpu) bcnn_hip_memcpy_d2h(t->grad_data, t->grad_data_gpu, bytes);
    return BCNN_SUCCESS;
}

bcnn_tensor *bcnn_peek_tensor(bcnn_net *net, int index) {
    return (index < 0 || index >= net->num_tensors) ? NULL : &net->tensors[index];
}

int bcnn_get_num_nodes(bcnn_net *net) { return net->num_nodes; }

int bcnn_get_node_tensor(bcnn_net *net, int node, int is_dst, int slot) {
    if (node < 0 || node >= net->num_nodes || slot < 0) return -1;
    const bcnn_node *nd = &net->nodes[node];
    if (is_dst) return slot < nd->num_dst ? nd->dst[slot] : -1;
    return slot < nd->num_src ? nd->src[slot] : -1;
}

void *bcnn_get_node_state(bcnn_net *net, int node, int which) {
    if (node < 0 || node >= net->num_nodes) return NULL;
    const bcnn_node *nd = &net->nodes[node];
    const bcnn_tensor *sm = NULL, *sv = NULL;
    if (nd->type == BCNN_LAYER_MAXPOOL) return which == 0 ? (void *)((bcnn_maxpool_param *)nd->param)->indexes_gpu : NULL;
    if (nd->type == BCNN_LAYER_CONV2D && ((bcnn_conv_param *)nd->param)->batch_norm) {
        sm = &((bcnn_conv_param *)nd->param)->saved_mean;
        sv = &((bcnn_conv_param *)nd->param)->saved_variance;
    } else if (nd->type == BCNN_LAYER_BATCHNORM) {
        sm = &((bcnn_batchnorm_param *)nd->param)->saved_mean;
        sv = &((bcnn_batchnorm_param *)nd->param)->saved_variance;
    }
    if (!sm) return NULL;
    if (which == 5) { /* the pre-normalisation values the backward pass works from (the reference's param->workspace) */
        if (nd->type == BCNN_LAYER_CONV2D) return ((bcnn_conv_param *)nd->param)->bn_workspace_gpu;
        const bcnn_batchnorm_param *bp = (const bcnn_batchnorm_param *)nd->param;
        return (bp->input_kept || !bp->workspace_gpu) ? net->tensors[nd->src[0]].data_gpu : bp->workspace_gpu;
    }
    switch (which) {
        case 1: return sm->data_gpu;
        case 2: return sv->data_gpu;
        case 3: return sm->grad_data_gpu;
        case 4: return sv->grad_data_gpu;
    }
    return NULL;
}

bcnn_status bcnn_forward_node(bcnn_net *net, int node) {
    if (node < 0 || node >= net->num_nodes || !net->nodes[node].forward) return BCNN_INVALID_PARAMETER;
    bcnn_materialize_data(net, -1); /* a single worker reads its inputs from the tensors themselves */
    net->nodes[node].forward(net, &net->nodes[node]);
    return BCNN_SUCCESS;
}

bcnn_status bcnn_backward_node(bcnn_net *net, int node) {
    if (node < 0 || node >= net->num_nodes || !net->nodes[node].backward) return BCNN_INVALID_PARAMETER;
    bcnn_hip_context *hc = hctx(net);
    bcnn_materialize_data(net, -1);
    bcnn_materialize_gradients(net, -1); /* a single worker reads and rewrites gradient tensors in place */
    /* outside the executor nobody promised that the gradients were left unfilled: accumulate like the reference */
    unsigned char *saved = hc->grad_fill_dead;
    hc->grad_fill_dead = NULL;
    net->nodes[node].backward(net, &net->nodes[node]);
    hc->grad_fill_dead = saved;
    return BCNN_SUCCESS;
}

void bcnn_synchronize(bcnn_net *net) {
    (void)net;
    bcnn_hip_sync();
}

/* ------------------------------------------------------------------------------------------------
 * data parallel helpers
 * ---------------------------------------------------------------------------------------------- */
bcnn_status bcnn_set_data_parallel(bcnn_net *net, int rank, int world_size) {
    if (world_size < 1 || rank < 0 || rank >= world_size) return BCNN_INVALID_PARAMETER;
    hctx(net)->dp_rank = rank;
    hctx(net)->dp_world = world_size;
    return BCNN_SUCCESS;
}

/* every rank continues from rank 0's parameters, whatever its own initialisation drew */
static void comm_sync_parameters(bcnn_net *net) {
    bcnn_hip_context *hc = hctx(net);
    if (!hc->comm_active || !hc->param_arena_gpu || hc->arena_size == 0) return;
    bcnn_hip_broadcast(hc->param_arena_gpu, hc->arena_size, 0);
    bcnn_hip_comm_join();
}

bcnn_status bcnn_set_data_parallel_comm(bcnn_net *net, int rank, int world_size, const char *id_path) {
    if (world_size < 1 || rank < 0 || rank >= world_size) return BCNN_INVALID_PARAMETER;
    if (world_size > 1 && (!id_path || !id_path[0])) return BCNN_INVALID_PARAMETER;
    bcnn_hip_context *hc = hctx(net);
    if (hc->comm_active) /* second call on the same net: it already holds the communicator */
        return (bcnn_hip_comm_world() == world_size && bcnn_hip_comm_rank() == rank) ? BCNN_SUCCESS
                                                                                       : BCNN_INVALID_PARAMETER;
    if (bcnn_hip_comm_world() == 0) {
        bcnn_hip_comm_init(rank, world_size, id_path); /* fatal on failure; this net is the first holder */
    } else {
        if (bcnn_hip_comm_world() != world_size || bcnn_hip_comm_rank() != rank) return BCNN_INVALID_PARAMETER;
        bcnn_hip_comm_retain(); /* one communicator per process, shared by its nets; the last bcnn_end_net destroys it */
    }
    hc->dp_rank = rank;
    hc->dp_world = world_size;
    hc->comm_active = 1;
    hc->comm_bucket = (size_t)(8u << 20) / sizeof(float);
    /* node order == arena order, backward walks it in reverse: the callback sees a growing tail (recomputed by
     * bcnn_compile_net when the arena is (re)built after this call) */
    bcnn_set_gradient_ready_callback(net, comm_on_ready, hc);
    comm_sync_parameters(net);
    return BCNN_SUCCESS;
}

float *bcnn_get_gradient_arena(bcnn_net *net, size_t *num_floats) {
    if (num_floats) *num_floats = hctx(net)->arena_size;
    return hctx(net)->grad_arena_gpu;
}

float *bcnn_get_parameter_arena(bcnn_net *net, size_t *num_floats) {
    if (num_floats) *num_floats = hctx(net)->arena_size;
    return hctx(net)->param_arena_gpu;
}

/* ------------------------------------------------------------------------------------------------
 * learner set-up (reference bcnn_learner.c:177-225; neither setter touches learner->optimizer,
 * so API users always train with SGD -- quirk 6 of SURVEY.md)
 * ---------------------------------------------------------------------------------------------- */
static bcnn_learner *learner_of(bcnn_net *net) {
    if (!net->learner) net->learner = (bcnn_learner *)calloc(1, sizeof(bcnn_learner));
    return net->learner;
}

void bcnn_set_learning_rate_policy(bcnn_net *net, bcnn_lr_decay decay_type, float gamma, float scale, float power,
                                   int max_batches, int step) {
    bcnn_learner *ln = learner_of(net);
    ln->decay_type = decay_type; ln->gamma = gamma; ln->scale = scale; ln->power = power;
    ln->max_batches = max_batches; ln->step = step;
}

void bcnn_set_adam_optimizer(bcnn_net *net, float learning_rate, float beta1, float beta2) {
    bcnn_learner *ln = learner_of(net);
    ln->base_learning_rate = ln->learning_rate = learning_rate;
    ln->beta1 = beta1; ln->beta2 = beta2;
    ln->momentum = 0.9f;
}

void bcnn_set_sgd_optimizer(bcnn_net *net, float learning_rate, float momentum) {
    bcnn_learner *ln = learner_of(net);
    ln->base_learning_rate = ln->learning_rate = learning_rate;
    ln->momentum = momentum;
}

void bcnn_set_weight_regularizer(bcnn_net *net, float weight_decay) { learner_of(net)->decay = weight_decay; }
