/* bcnn_weights_io.c -- model files: bcnn_save_weights / bcnn_load_weights.
 *
 * Byte-compatible with the reference (src/bcnn_net.c:595-681 writer, :1219-1558 reader):
 *   "BCNN" | u32 major | u32 minor | u32 patch | per node, in node order:
 *     conv / depthwise / full-connected : biases, weights [, conv with batch-norm: means, variances, scales]
 *     PReLU activation node             : slopes
 *     batch-norm node                   : means, variances, scales, biases (dst channels each)
 * The reader additionally expects a fused convolution's PReLU slopes after its batch-norm block although the
 * writer never stores them (reference asymmetry, kept: such a file fails with BCNN_INVALID_MODEL in both
 * implementations). Files named *.weights are read in the Darknet layout (int header, `seen` counter, scales
 * before means, weights last, no batch-norm biases); *.onnx is rejected like the reference does.
 * Host buffers are the staging area, the device mirrors are refreshed / read back around the file access
 * (the reference's CUDA build does the same, bcnn_net.c:620-668, 1294-1297). PREDICT-mode nets get the
 * batch-norm statistics folded into scales and biases at load time, as the reference's CPU build does
 * (:1281-1290, :1390-1399): the PREDICT forward here is that build's `x*scale + bias`. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "bcnn_internal.h"
#include "bcnn_hip.h"

static const char kMagic[4] = {0x42, 0x43, 0x4E, 0x4E};

static void pull(bcnn_tensor *t, int count) {
    if (t->data && t->data_gpu) bcnn_hip_memcpy_d2h(t->data, t->data_gpu, (size_t)count * sizeof(float));
}
static void push(bcnn_tensor *t, int count) {
    if (t->data && t->data_gpu) bcnn_hip_memcpy_h2d(t->data_gpu, t->data, (size_t)count * sizeof(float));
}
static int put(FILE *fp, bcnn_tensor *t, int count) {
    pull(t, count);
    return fwrite(t->data, sizeof(float), (size_t)count, fp) == (size_t)count;
}

bcnn_status bcnn_save_weights(bcnn_net *net, const char *filename) {
    FILE *fp = filename ? fopen(filename, "wb") : NULL;
    BCNN_CHECK_AND_LOG(net->log_ctx, fp, BCNN_INVALID_PARAMETER, "Could not open model file %s\n",
                       filename ? filename : "(null)");
    bcnn_hip_sync();
    const uint32_t ver[3] = {BCNN_VERSION_MAJOR, BCNN_VERSION_MINOR, BCNN_VERSION_PATCH};
    int ok = fwrite(kMagic, 1, 4, fp) == 4 && fwrite(ver, sizeof(uint32_t), 3, fp) == 3;
    for (int i = 0; ok && i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->type == BCNN_LAYER_CONV2D || nd->type == BCNN_LAYER_TRANSPOSE_CONV2D ||
            nd->type == BCNN_LAYER_DEPTHWISE_CONV2D || nd->type == BCNN_LAYER_FULL_CONNECTED) {
            bcnn_tensor *w = &net->tensors[nd->src[1]], *b = &net->tensors[nd->src[2]];
            ok = put(fp, b, bcnn_tensor_size(b)) && put(fp, w, bcnn_tensor_size(w));
            if (ok && nd->type == BCNN_LAYER_CONV2D && ((bcnn_conv_param *)nd->param)->batch_norm == 1)
                for (int k = 3; ok && k <= 5; ++k) {
                    bcnn_tensor *t = &net->tensors[nd->src[k]];
                    ok = put(fp, t, bcnn_tensor_size(t));
                }
        } else if (nd->type == BCNN_LAYER_ACTIVATION) {
            if (((bcnn_activation_param *)nd->param)->activation == BCNN_ACT_PRELU) {
                bcnn_tensor *w = &net->tensors[nd->src[1]];
                ok = put(fp, w, bcnn_tensor_size(w));
            }
        } else if (nd->type == BCNN_LAYER_BATCHNORM) {
            const int c = net->tensors[nd->dst[0]].c;
            for (int k = 1; ok && k <= 4; ++k) ok = put(fp, &net->tensors[nd->src[k]], c);
        }
    }
    fclose(fp);
    BCNN_CHECK_AND_LOG(net->log_ctx, ok, BCNN_INVALID_DATA, "Short write on model file %s\n", filename);
    return BCNN_SUCCESS;
}

/* ---- reader ------------------------------------------------------------------------------------- */
static int model_format(const char *filename) { /* by extension, like bcnn_model_find_format (:1467-1483) */
    const char *dot = strrchr(filename, '.');
    const char *ext = dot ? dot + 1 : filename;
    if (strcmp(ext, "weights") == 0) return 1;
    if (strcmp(ext, "onnx") == 0) return 2;
    return 0;
}

#define READ_OR_FAIL(net, fp, t, count, what)                                                                  \
    do {                                                                                                       \
        size_t nr_ = fread((t)->data, sizeof(float), (size_t)(count), (fp));                                   \
        BCNN_CHECK_AND_LOG((net)->log_ctx, nr_ == (size_t)(count), BCNN_INVALID_MODEL,                         \
                           "Inconsistent " what " size %s: expected %d but found %lu\n",                       \
                           (t)->name ? (t)->name : "", (int)(count), (unsigned long)nr_);                      \
    } while (0)

static void fold_batchnorm(bcnn_tensor *m, bcnn_tensor *v, bcnn_tensor *s, bcnn_tensor *b, int n) {
    for (int i = 0; i < n; ++i) {
        b->data[i] = b->data[i] - (s->data[i] * m->data[i]) / (sqrtf(v->data[i] + 0.000001f));
        s->data[i] = s->data[i] / (sqrtf(v->data[i] + 0.000001f));
    }
}

static bcnn_status load_conv(bcnn_net *net, bcnn_node *nd, FILE *fp, int format) {
    bcnn_tensor *w = &net->tensors[nd->src[1]], *b = &net->tensors[nd->src[2]];
    const int w_sz = bcnn_tensor_size(w), b_sz = bcnn_tensor_size(b);
    READ_OR_FAIL(net, fp, b, b_sz, "biases");
    if (format == 0) READ_OR_FAIL(net, fp, w, w_sz, "weights");
    if (nd->type == BCNN_LAYER_CONV2D) {
        bcnn_conv_param *p = (bcnn_conv_param *)nd->param;
        if (p->batch_norm == 1) {
            bcnn_tensor *m = &net->tensors[nd->src[3]], *v = &net->tensors[nd->src[4]], *s = &net->tensors[nd->src[5]];
            const int n = bcnn_tensor_size(s);
            if (format == 1) READ_OR_FAIL(net, fp, s, n, "batchnorm scales");
            READ_OR_FAIL(net, fp, m, bcnn_tensor_size(m), "batchnorm means");
            READ_OR_FAIL(net, fp, v, bcnn_tensor_size(v), "batchnorm variances");
            if (format == 0) READ_OR_FAIL(net, fp, s, n, "batchnorm scales");
            if (net->mode == BCNN_MODE_PREDICT) fold_batchnorm(m, v, s, b, n);
            push(m, bcnn_tensor_size(m)); push(v, bcnn_tensor_size(v)); push(s, n);
        }
    }
    if (format == 1) READ_OR_FAIL(net, fp, w, w_sz, "weights");
    if (nd->type == BCNN_LAYER_CONV2D) {
        bcnn_conv_param *p = (bcnn_conv_param *)nd->param;
        if (p->activation == BCNN_ACT_PRELU) {
            bcnn_tensor *sl = &net->tensors[nd->src[3 + 3 * p->batch_norm]];
            READ_OR_FAIL(net, fp, sl, bcnn_tensor_size(sl), "prelu slopes");
            push(sl, bcnn_tensor_size(sl));
        }
    }
    push(w, w_sz); push(b, b_sz);
    return BCNN_SUCCESS;
}

static bcnn_status load_batchnorm(bcnn_net *net, bcnn_node *nd, FILE *fp, int format) {
    bcnn_tensor *m = &net->tensors[nd->src[1]], *v = &net->tensors[nd->src[2]], *s = &net->tensors[nd->src[3]],
                *b = &net->tensors[nd->src[4]];
    const int c = net->tensors[nd->dst[0]].c;
    if (format == 1) READ_OR_FAIL(net, fp, s, c, "scales");
    READ_OR_FAIL(net, fp, m, c, "means");
    READ_OR_FAIL(net, fp, v, c, "variances");
    if (format == 0) {
        READ_OR_FAIL(net, fp, s, c, "scales");
        READ_OR_FAIL(net, fp, b, c, "biases");
    }
    if (net->mode == BCNN_MODE_PREDICT) fold_batchnorm(m, v, s, b, c);
    push(m, c); push(v, c); push(s, c); push(b, c);
    return BCNN_SUCCESS;
}

static void transpose_inplace(float *a, int rows, int cols) {
    float *t = (float *)calloc((size_t)rows * cols, sizeof(float));
    if (!t) return;
    for (int x = 0; x < rows; ++x)
        for (int y = 0; y < cols; ++y) t[(size_t)y * rows + x] = a[(size_t)x * cols + y];
    memcpy(a, t, (size_t)rows * cols * sizeof(float));
    free(t);
}

static bcnn_status load_fullc(bcnn_net *net, bcnn_node *nd, FILE *fp, int need_transpose) {
    bcnn_tensor *w = &net->tensors[nd->src[1]], *b = &net->tensors[nd->src[2]];
    const int w_sz = bcnn_tensor_size(w), b_sz = bcnn_tensor_size(b);
    READ_OR_FAIL(net, fp, b, b_sz, "biases");
    READ_OR_FAIL(net, fp, w, w_sz, "weights");
    if (need_transpose) {
        const bcnn_tensor *x = &net->tensors[nd->src[0]], *y = &net->tensors[nd->dst[0]];
        transpose_inplace(w->data, x->c * x->h * x->w, y->c * y->h * y->w);
    }
    push(w, w_sz); push(b, b_sz);
    return BCNN_SUCCESS;
}

bcnn_status bcnn_load_weights(bcnn_net *net, const char *filename) {
    BCNN_CHECK_AND_LOG(net->log_ctx, filename, BCNN_INVALID_PARAMETER, "Can not open file %s\n", "(null)");
    const int format = model_format(filename);
    FILE *fp = fopen(filename, "rb");
    BCNN_CHECK_AND_LOG(net->log_ctx, fp, BCNN_INVALID_PARAMETER, "Can not open file %s\n", filename);
    int need_transpose = 0;
    if (format == 0) {
        char magic[4] = {0, 0, 0, 0};
        uint32_t ver[3] = {0, 0, 0};
        size_t nr = fread(magic, 1, 4, fp);
        nr += fread(ver, sizeof(uint32_t), 3, fp);
        if (nr != 7 || memcmp(magic, kMagic, 4) != 0) {
            bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Invalid format for model file %s\n", filename);
            fclose(fp);
            return BCNN_INVALID_MODEL;
        }
        BCNN_INFO(net->log_ctx, "BCNN version %d.%d.%d used for model %s\n", (int)ver[0], (int)ver[1], (int)ver[2],
                  filename);
    } else if (format == 1) {
        int hdr[3] = {0, 0, 0};
        size_t nr = fread(hdr, sizeof(int), 3, fp);
        uint64_t seen = 0;
        if ((hdr[0] * 10 + hdr[1]) >= 2 && hdr[0] < 1000 && hdr[1] < 1000) {
            nr += fread(&seen, sizeof(uint64_t), 1, fp);
        } else {
            int iseen = 0;
            nr += fread(&iseen, sizeof(int), 1, fp);
            seen = (uint64_t)iseen;
        }
        (void)nr;
        BCNN_INFO(net->log_ctx, "Darknet version %d.%d seen %lu\n", hdr[0], hdr[1], (unsigned long)seen);
        need_transpose = (hdr[0] > 1000) || (hdr[1] > 1000);
    } else {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Model file %s format is not yet supported\n", filename);
        fclose(fp);
        return BCNN_INVALID_MODEL;
    }
    bcnn_status st = BCNN_SUCCESS;
    for (int i = 0; st == BCNN_SUCCESS && i < net->num_nodes; ++i) {
        bcnn_node *nd = &net->nodes[i];
        if (nd->type == BCNN_LAYER_CONV2D || nd->type == BCNN_LAYER_TRANSPOSE_CONV2D ||
            nd->type == BCNN_LAYER_DEPTHWISE_CONV2D) {
            st = load_conv(net, nd, fp, format);
        } else if (nd->type == BCNN_LAYER_ACTIVATION) {
            if (((bcnn_activation_param *)nd->param)->activation == BCNN_ACT_PRELU && format == 0) {
                bcnn_tensor *w = &net->tensors[nd->src[1]];
                size_t nr = fread(w->data, sizeof(float), (size_t)bcnn_tensor_size(w), fp);
                if (nr != (size_t)bcnn_tensor_size(w)) {
                    bcnn_log(net->log_ctx, BCNN_LOG_ERROR,
                             "Inconsistent prelu weights size: expected %d but found %lu\n", bcnn_tensor_size(w),
                             (unsigned long)nr);
                    st = BCNN_INVALID_MODEL;
                } else {
                    push(w, bcnn_tensor_size(w));
                }
            }
        } else if (nd->type == BCNN_LAYER_BATCHNORM) {
            st = load_batchnorm(net, nd, fp, format);
        } else if (nd->type == BCNN_LAYER_FULL_CONNECTED) {
            st = load_fullc(net, nd, fp, need_transpose);
        }
    }
    fclose(fp);
    bcnn_hip_sync();
    if (st == BCNN_SUCCESS) BCNN_INFO(net->log_ctx, "Model %s loaded succesfully\n", filename);
    return st;
}
