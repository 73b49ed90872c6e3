/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin accessors compiled INTO oracle/_ref/libbcnn_ref.so next to the unmodified reference
 * sources (see oracle/Makefile, target `ref`). The reference's public API (inc/bcnn/bcnn.h) is
 * driven straight from Python/ctypes; this file only exposes the layer-private state that the
 * public API does not reach (maxpool indexes, saved batch statistics, ...), and a timing loop
 * for bench.py's cpu_baseline leg (kind = "reference").
 *
 * Nothing here is part of the product; the product never links or loads this library.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <bcnn/bcnn.h>
#include "bcnn_net.h"
#include "bcnn_node.h"
#include "bcnn_tensor.h"
#include "bcnn_utils.h"
#include "bcnn_mat.h"
#include "bcnn_conv_layer.h"
#include "bcnn_batchnorm_layer.h"
#include "bcnn_maxpool_layer.h"
#include "bcnn_depthwise_conv_layer.h"

int ref_num_nodes(bcnn_net *net) { return net->num_nodes; }
int ref_num_tensors(bcnn_net *net) { return net->num_tensors; }
int ref_node_type(bcnn_net *net, int node) { return (int)net->nodes[node].type; }
int ref_node_num_src(bcnn_net *net, int node) { return net->nodes[node].num_src; }
int ref_node_src(bcnn_net *net, int node, int i) { return net->nodes[node].src[i]; }
int ref_node_dst(bcnn_net *net, int node, int i) { return net->nodes[node].dst[i]; }
/* raw tensor pointer (bcnn_get_tensor_by_index is equivalent on a CPU build) */
bcnn_tensor *ref_tensor(bcnn_net *net, int idx) { return &net->tensors[idx]; }
const char *ref_tensor_name(bcnn_net *net, int idx) { return net->tensors[idx].name; }
void ref_set_mode_raw(bcnn_net *net, int mode) { net->mode = (bcnn_mode)mode; }
void ref_set_threads(bcnn_net *net, int nt) { net->num_threads = nt; }
int ref_get_threads(bcnn_net *net) { return net->num_threads; }

/* one node at a time (tests/test_teacher_forced.py): the plug-in workers of src/bcnn_node.h:44-47 */
void ref_forward_node(bcnn_net *net, int node) { net->nodes[node].forward(net, &net->nodes[node]); }
void ref_backward_node(bcnn_net *net, int node) { net->nodes[node].backward(net, &net->nodes[node]); }

/* maxpool: param->indexes (src/layers/bcnn_maxpool_layer.h:34-47) */
int *ref_maxpool_indexes(bcnn_net *net, int node) {
    if (net->nodes[node].type != BCNN_LAYER_MAXPOOL) return NULL;
    return ((bcnn_maxpool_param *)net->nodes[node].param)->indexes;
}

/* batch statistics kept by a fused-BN conv node or a stand-alone BN node.
 * which: 0 saved_mean.data 1 saved_variance.data 2 saved_mean.grad_data 3 saved_variance.grad_data
 *        4 x_norm 5 workspace (pre-normalisation copy of x) */
float *ref_bn_field(bcnn_net *net, int node, int which) {
    bcnn_tensor *sm = NULL, *sv = NULL;
    float *xn = NULL, *ws = NULL;
    if (net->nodes[node].type == BCNN_LAYER_CONV2D) {
        bcnn_conv_param *p = (bcnn_conv_param *)net->nodes[node].param;
        if (!p->batch_norm) return NULL;
        sm = &p->saved_mean; sv = &p->saved_variance; xn = p->x_norm; ws = p->workspace;
    } else if (net->nodes[node].type == BCNN_LAYER_BATCHNORM) {
        bcnn_batchnorm_param *p = (bcnn_batchnorm_param *)net->nodes[node].param;
        sm = &p->saved_mean; sv = &p->saved_variance; xn = p->x_norm; ws = p->workspace;
    } else {
        return NULL;
    }
    switch (which) {
        case 0: return sm->data;
        case 1: return sv->data;
        case 2: return sm->grad_data;
        case 3: return sv->grad_data;
        case 4: return xn;
        case 5: return ws;
    }
    return NULL;
}

/* bcnn_gemm needs the net-private context (src/kernels/bcnn_mat.c:2627) */
void ref_gemm(bcnn_net *net, int ta, int tb, int m, int n, int k, float alpha, float *A, int lda,
              float *B, int ldb, float beta, float *C, int ldc) {
    bcnn_gemm(net->gemm_ctx, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc,
              net->num_threads);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* cpu_baseline timing: `iters` x (bcnn_forward + bcnn_backward), returns seconds per iteration
 * (best of iters) -- same calls the reference's bcnn_train_on_batch makes
 * (src/bcnn_net.c:452-462) minus the data loader and the update. */
double ref_time_fwd_bwd(bcnn_net *net, int warmup, int iters, double *fwd_s, double *bwd_s) {
    double best = 1e30, bf = 0, bb = 0;
    for (int i = 0; i < warmup + iters; ++i) {
        double t0 = now_s();
        bcnn_forward(net);
        double t1 = now_s();
        bcnn_backward(net);
        double t2 = now_s();
        if (i >= warmup && (t2 - t0) < best) { best = t2 - t0; bf = t1 - t0; bb = t2 - t1; }
    }
    if (fwd_s) *fwd_s = bf;
    if (bwd_s) *bwd_s = bb;
    return best;
}
