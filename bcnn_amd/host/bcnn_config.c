/* bcnn_config.c -- bcnn_load_net: build a net from an INI-style config file.
 *
 * Mirrors the reference loader (src/bcnn_net.c:504-593 net parameters, :716-966 layer parameters, :968-1112
 * section -> builder mapping, :1114-1216 driver; INI reader src/bh/inc/bh/bh_ini.h):
 *   - all blanks and tabs inside a line are removed before parsing; lines starting with '#', ';', '!' or empty
 *     are skipped; "[name]" opens a section; every other line must split into exactly "key=value";
 *   - the first section must be [net] or [network] and non-empty; unknown keys there are ignored (they belong
 *     to the command-line tool: data sources, output_model, ...);
 *   - every further section is one layer; `src` may list several tensors separated by ',';
 *   - when the model file is a Darknet *.weights file, sections may omit src/dst (implicit "lid<i-1>" ->
 *     "lid<i>" chaining), `pad=1` means size/2 and `layers=` / `from=` name earlier sections.
 * One deliberate difference: the reference discards the status of the layer builders; here a builder that
 * fails (an out-of-scope layer of this build, a bad tensor name) aborts the load with its status instead of
 * leaving a half-built graph. */
#include <stdio.h>
#include <string.h>

#include "bcnn_internal.h"

/* ---- tiny INI reader ------------------------------------------------------------------------------ */
typedef struct { char *name, *val; } ini_key;
typedef struct { char *name; ini_key *keys; int num_keys; } ini_section;
typedef struct { ini_section *sections; int num_sections; } ini_file;

static char *dup_str(const char *s) {
    size_t n = strlen(s) + 1;
    char *d = (char *)malloc(n);
    if (d) memcpy(d, s, n);
    return d;
}

static void ini_free(ini_file *f) {
    for (int i = 0; i < f->num_sections; ++i) {
        for (int j = 0; j < f->sections[i].num_keys; ++j) {
            free(f->sections[i].keys[j].name);
            free(f->sections[i].keys[j].val);
        }
        free(f->sections[i].keys);
        free(f->sections[i].name);
    }
    free(f->sections);
    f->sections = NULL;
    f->num_sections = 0;
}

static char *read_line(FILE *fp) { /* whole line, any length; NULL at end of file */
    size_t cap = 256, len = 0;
    char *buf = (char *)malloc(cap);
    int ch;
    if (!buf) return NULL;
    while ((ch = fgetc(fp)) != EOF) {
        if (len + 2 > cap) {
            cap *= 2;
            char *nb = (char *)realloc(buf, cap);
            if (!nb) { free(buf); return NULL; }
            buf = nb;
        }
        if (ch == '\n') { buf[len] = '\0'; return buf; }
        buf[len++] = (char)ch;
    }
    if (len == 0) { free(buf); return NULL; }
    buf[len] = '\0';
    return buf;
}

static int ini_read(const char *path, ini_file *out) {
    FILE *fp = fopen(path, "r");
    out->sections = NULL;
    out->num_sections = 0;
    if (!fp) {
        fprintf(stderr, "[ERROR] Could not open file: %s\n", path);
        return -1;
    }
    char *line;
    int rc = 0;
    while (rc == 0 && (line = read_line(fp)) != NULL) {
        size_t k = 0;
        for (size_t i = 0; line[i]; ++i) /* bh_strstrip: drop blanks everywhere, not only at the ends */
            if (line[i] != ' ' && line[i] != '\t' && line[i] != '\n' && line[i] != '\r') line[k++] = line[i];
        line[k] = '\0';
        if (line[0] == '[') {
            ini_section *ns = (ini_section *)realloc(out->sections, (size_t)(out->num_sections + 1) * sizeof(ini_section));
            if (!ns) rc = -1;
            else {
                out->sections = ns;
                ini_section *s = &out->sections[out->num_sections++];
                s->name = dup_str(line); s->keys = NULL; s->num_keys = 0;
            }
        } else if (line[0] != '\0' && line[0] != '#' && line[0] != ';' && line[0] != '!') {
            char *eq = strchr(line, '=');
            if (out->num_sections == 0 || !eq || eq == line || eq[1] == '\0' || strchr(eq + 1, '=')) {
                fprintf(stderr, "[ERROR] Invalid key section %s\n", line);
                rc = -1;
            } else {
                ini_section *s = &out->sections[out->num_sections - 1];
                ini_key *nk = (ini_key *)realloc(s->keys, (size_t)(s->num_keys + 1) * sizeof(ini_key));
                if (!nk) rc = -1;
                else {
                    s->keys = nk;
                    *eq = '\0';
                    s->keys[s->num_keys].name = dup_str(line);
                    s->keys[s->num_keys].val = dup_str(eq + 1);
                    s->num_keys++;
                }
            }
        }
        free(line);
    }
    fclose(fp);
    if (rc != 0) {
        fprintf(stderr, "[ERROR] Failed to parse config file %s\n", path);
        ini_free(out);
    }
    return rc;
}

/* ---- [net] parameters (reference bcnn_net_set_param, :504-593) ------------------------------------- */
static void net_set_param(bcnn_net *net, const char *name, const char *val) {
    bcnn_learner *ln = net->learner;
    bcnn_data_augmenter *da = net->data_aug;
    if (!strcmp(name, "input_width") || !strcmp(name, "width")) net->tensors[0].w = atoi(val);
    else if (!strcmp(name, "input_height") || !strcmp(name, "height")) net->tensors[0].h = atoi(val);
    else if (!strcmp(name, "input_channels") || !strcmp(name, "channels")) net->tensors[0].c = atoi(val);
    else if (!strcmp(name, "batch_size") || !strcmp(name, "batch")) { net->batch_size = atoi(val); net->tensors[0].n = atoi(val); }
    else if (ln && !strcmp(name, "max_batches")) ln->max_batches = atoi(val);
    else if (ln && (!strcmp(name, "learning_policy") || !strcmp(name, "decay_type"))) {
        if (!strcmp(val, "sigmoid")) ln->decay_type = BCNN_LR_DECAY_SIGMOID;
        else if (!strcmp(val, "exp")) ln->decay_type = BCNN_LR_DECAY_EXP;
        else if (!strcmp(val, "inv")) ln->decay_type = BCNN_LR_DECAY_INV;
        else if (!strcmp(val, "step")) ln->decay_type = BCNN_LR_DECAY_STEP;
        else if (!strcmp(val, "poly")) ln->decay_type = BCNN_LR_DECAY_POLY;
        else ln->decay_type = BCNN_LR_DECAY_CONSTANT;
    } else if (ln && !strcmp(name, "optimizer")) {
        if (!strcmp(val, "sgd")) ln->optimizer = BCNN_OPTIM_SGD;
        else if (!strcmp(val, "adam")) ln->optimizer = BCNN_OPTIM_ADAM;
    } else if (ln && !strcmp(name, "step")) ln->step = atoi(val);
    else if (ln && !strcmp(name, "learning_rate")) ln->base_learning_rate = ln->learning_rate = (float)atof(val);
    else if (ln && !strcmp(name, "beta1")) ln->beta1 = (float)atof(val);
    else if (ln && !strcmp(name, "beta2")) ln->beta2 = (float)atof(val);
    else if (ln && !strcmp(name, "decay")) ln->decay = (float)atof(val);
    else if (ln && !strcmp(name, "momentum")) ln->momentum = (float)atof(val);
    else if (ln && !strcmp(name, "gamma")) ln->gamma = (float)atof(val);
    else if (da && !strcmp(name, "range_shift_x")) da->range_shift_x = atoi(val);
    else if (da && !strcmp(name, "range_shift_y")) da->range_shift_y = atoi(val);
    else if (da && !strcmp(name, "min_scale")) da->min_scale = (float)atof(val);
    else if (da && !strcmp(name, "max_scale")) da->max_scale = (float)atof(val);
    else if (da && !strcmp(name, "rotation_range")) da->rotation_range = (float)atof(val);
    else if (da && !strcmp(name, "min_contrast")) da->min_contrast = (float)atof(val);
    else if (da && !strcmp(name, "max_contrast")) da->max_contrast = (float)atof(val);
    else if (da && !strcmp(name, "min_brightness")) da->min_brightness = atoi(val);
    else if (da && !strcmp(name, "max_brightness")) da->max_brightness = atoi(val);
    else if (da && !strcmp(name, "max_distortion")) da->max_distortion = (float)atof(val);
    else if (da && !strcmp(name, "max_spots")) da->max_random_spots = (int)atof(val);
    else if (da && !strcmp(name, "flip_h")) da->random_fliph = 1;
    else if (da && !strcmp(name, "mean_r")) da->mean_r = (float)atof(val) / 255.0f;
    else if (da && !strcmp(name, "mean_g")) da->mean_g = (float)atof(val) / 255.0f;
    else if (da && !strcmp(name, "mean_b")) da->mean_b = (float)atof(val) / 255.0f;
    else if (da && !strcmp(name, "swap_to_bgr")) da->swap_to_bgr = atoi(val);
    else if (da && !strcmp(name, "no_input_norm")) da->no_input_norm = atoi(val);
}

/* ---- layer parameters (reference bcnn_layer_param, :683-966) ------------------------------------- */
#define MAX_SRCS 16
typedef struct {
    int stride, pad, n_filts, size, outputs, num_groups, batchnorm, in_w, in_h, in_c;
    float alpha, beta, k, rate;
    bcnn_padding padding_type;
    bcnn_activation a;
    bcnn_filler_type init;
    bcnn_loss_metric cost;
    bcnn_loss loss;
    int num_srcs;
    char *src_id[MAX_SRCS];
    char *dst_id;
} layer_param;

static void lp_reset(layer_param *lp) {
    for (int i = 0; i < lp->num_srcs; ++i) free(lp->src_id[i]);
    free(lp->dst_id);
    memset(lp, 0, sizeof(*lp));
    lp->stride = 1; lp->n_filts = 1; lp->size = 3; lp->num_groups = 1; lp->rate = 1.0f;
    lp->padding_type = BCNN_PADDING_SAME; lp->a = BCNN_ACT_NONE; lp->init = BCNN_FILLER_XAVIER;
    lp->cost = BCNN_METRIC_SSE; lp->loss = BCNN_LOSS_EUCLIDEAN;
}

static void lp_set_srcs(layer_param *lp, const char *list) {
    for (int i = 0; i < lp->num_srcs; ++i) free(lp->src_id[i]);
    lp->num_srcs = 0;
    const char *p = list;
    while (*p && lp->num_srcs < MAX_SRCS) {
        const char *e = strchr(p, ',');
        size_t n = e ? (size_t)(e - p) : strlen(p);
        if (n > 0) {
            char *s = (char *)malloc(n + 1);
            memcpy(s, p, n);
            s[n] = '\0';
            lp->src_id[lp->num_srcs++] = s;
        }
        if (!e) break;
        p = e + 1;
    }
}

static void lp_set_lid(char **dst, int id) {
    char lid[32];
    snprintf(lid, sizeof(lid), "lid%d", id);
    free(*dst);
    *dst = dup_str(lid);
}

static void lp_set(bcnn_net *net, int section_idx, layer_param *lp, const char *name, const char *val, int format) {
    if (!strcmp(name, "dropout_rate") || !strcmp(name, "rate")) lp->rate = (float)atof(val);
    else if (!strcmp(name, "filters")) lp->n_filts = atoi(val);
    else if (!strcmp(name, "size")) lp->size = atoi(val);
    else if (!strcmp(name, "stride")) lp->stride = atoi(val);
    else if (!strcmp(name, "padding")) {
        if (format == 1) { lp->pad = atoi(val); lp->padding_type = lp->pad ? BCNN_PADDING_SAME : BCNN_PADDING_VALID; }
    } else if (!strcmp(name, "pad")) {
        if (format == 0) lp->pad = atoi(val);
        else lp->pad = atoi(val) ? lp->size / 2 : 0; /* Darknet: boolean, needs `size` to come first */
    } else if (!strcmp(name, "num_groups") || !strcmp(name, "groups")) lp->num_groups = atoi(val);
    else if (!strcmp(name, "alpha")) lp->alpha = (float)atoi(val); /* atoi: as the reference parses these three */
    else if (!strcmp(name, "beta")) lp->beta = (float)atoi(val);
    else if (!strcmp(name, "k")) lp->k = (float)atoi(val);
    else if (!strcmp(name, "w")) lp->in_w = atoi(val);
    else if (!strcmp(name, "h")) lp->in_h = atoi(val);
    else if (!strcmp(name, "c")) lp->in_c = atoi(val);
    else if (!strcmp(name, "bn") || !strcmp(name, "batchnorm") || !strcmp(name, "batch_normalize")) lp->batchnorm = atoi(val);
    else if (!strcmp(name, "src")) lp_set_srcs(lp, val);
    else if (!strcmp(name, "dst")) { free(lp->dst_id); lp->dst_id = dup_str(val); }
    else if (!strcmp(name, "output")) lp->outputs = atoi(val);
    else if (!strcmp(name, "padding_type")) {
        if (!strcmp(val, "same")) lp->padding_type = BCNN_PADDING_SAME;
        else if (!strcmp(val, "valid")) lp->padding_type = BCNN_PADDING_VALID;
        else if (!strcmp(val, "caffe")) lp->padding_type = BCNN_PADDING_CAFFE;
    } else if (!strcmp(name, "function") || !strcmp(name, "activation")) {
        if (!strcmp(val, "relu")) lp->a = BCNN_ACT_RELU;
        else if (!strcmp(val, "tanh")) lp->a = BCNN_ACT_TANH;
        else if (!strcmp(val, "ramp")) lp->a = BCNN_ACT_RAMP;
        else if (!strcmp(val, "clamp")) lp->a = BCNN_ACT_CLAMP;
        else if (!strcmp(val, "softplus")) lp->a = BCNN_ACT_SOFTPLUS;
        else if (!strcmp(val, "leaky_relu") || !strcmp(val, "lrelu") || !strcmp(val, "leaky")) lp->a = BCNN_ACT_LRELU;
        else if (!strcmp(val, "prelu")) lp->a = BCNN_ACT_PRELU;
        else if (!strcmp(val, "abs")) lp->a = BCNN_ACT_ABS;
        else if (!strcmp(val, "none") || !strcmp(val, "linear")) lp->a = BCNN_ACT_NONE;
        else {
            bcnn_log(net->log_ctx, BCNN_LOG_WARNING, "Unknown activation type %s, going with ReLU\n", val);
            lp->a = BCNN_ACT_RELU;
        }
    } else if (!strcmp(name, "init")) {
        if (!strcmp(val, "xavier")) lp->init = BCNN_FILLER_XAVIER;
        else if (!strcmp(val, "msra")) lp->init = BCNN_FILLER_MSRA;
        else {
            bcnn_log(net->log_ctx, BCNN_LOG_WARNING, "Unknown init type %s, going with xavier init\n", val);
            lp->init = BCNN_FILLER_XAVIER;
        }
    } else if (!strcmp(name, "metric")) {
        if (!strcmp(val, "error")) lp->cost = BCNN_METRIC_ERROR_RATE;
        else if (!strcmp(val, "logloss")) lp->cost = BCNN_METRIC_LOGLOSS;
        else if (!strcmp(val, "sse")) lp->cost = BCNN_METRIC_SSE;
        else if (!strcmp(val, "mse")) lp->cost = BCNN_METRIC_MSE;
        else if (!strcmp(val, "crps")) lp->cost = BCNN_METRIC_CRPS;
        else if (!strcmp(val, "dice")) lp->cost = BCNN_METRIC_DICE;
        else {
            bcnn_log(net->log_ctx, BCNN_LOG_WARNING, "Unknown cost metric %s, going with sse\n", val);
            lp->cost = BCNN_METRIC_SSE;
        }
    } else if (!strcmp(name, "loss")) {
        if (!strcmp(val, "l2") || !strcmp(val, "euclidean")) lp->loss = BCNN_LOSS_EUCLIDEAN;
        else if (!strcmp(val, "lifted_struct_similarity")) lp->loss = BCNN_LOSS_LIFTED_STRUCT;
        else {
            bcnn_log(net->log_ctx, BCNN_LOG_WARNING, "Unknown loss %s, going with euclidean loss\n", val);
            lp->loss = BCNN_LOSS_EUCLIDEAN;
        }
    } else if (!strcmp(name, "layers")) { /* Darknet [route]: absolute (0-based) or relative section numbers */
        lp_set_srcs(lp, val);
        for (int i = 0; i < lp->num_srcs; ++i) {
            const int l = atoi(lp->src_id[i]);
            lp_set_lid(&lp->src_id[i], l >= 0 ? l + 1 : section_idx + l);
        }
    } else if (!strcmp(name, "from")) { /* Darknet [shortcut]: previous section + the named one */
        const int l = atoi(val);
        lp_set_srcs(lp, "a,b");
        lp_set_lid(&lp->src_id[0], section_idx - 1);
        lp_set_lid(&lp->src_id[1], l >= 0 ? l + 1 : section_idx + l);
    }
    /* anchors / masks / classes of the YOLO head: that layer is not built here, the keys are ignored */
}

static int is_any(const char *name, const char *a, const char *b, const char *c, const char *d) {
    return !strcmp(name, a) || (b && !strcmp(name, b)) || (c && !strcmp(name, c)) || (d && !strcmp(name, d));
}

static bcnn_status add_layer(bcnn_net *net, const char *name, const layer_param *lp) {
    if (net->num_nodes == 0) {
        BCNN_CHECK_AND_LOG(net->log_ctx, net->tensors[0].w > 0 && net->tensors[0].h > 0 && net->tensors[0].c > 0,
                           BCNN_INVALID_PARAMETER, "Input's width, height and channels must be > 0\n");
        BCNN_CHECK_AND_LOG(net->log_ctx, net->tensors[0].n > 0, BCNN_INVALID_PARAMETER, "Batch size must be > 0\n");
    }
    BCNN_CHECK_AND_LOG(net->log_ctx, lp->num_srcs > 0 && lp->src_id[0], BCNN_INVALID_PARAMETER,
                       "Invalid input node name. Hint: Are you sure that 'src' field is correctly setup?\n");
    const char *src = lp->src_id[0], *dst = lp->dst_id;
    const int needs_dst = !is_any(name, "[input]", "[activation]", "[nl]", "[dropout]");
    BCNN_CHECK_AND_LOG(net->log_ctx, !needs_dst || dst, BCNN_INVALID_PARAMETER,
                       "Invalid output node name. Hint: Are you sure that 'dst' field is correctly setup?\n");
    if (!strcmp(name, "[input]")) return bcnn_add_input(net, lp->in_w, lp->in_h, lp->in_c, src);
    if (is_any(name, "[conv]", "[convolutional]", NULL, NULL))
        return bcnn_add_convolutional_layer(net, lp->n_filts, lp->size, lp->stride, lp->pad, lp->num_groups,
                                            lp->batchnorm, lp->init, lp->a, 0, src, dst);
    if (is_any(name, "[deconv]", "[deconvolutional]", NULL, NULL))
        return bcnn_add_deconvolutional_layer(net, lp->n_filts, lp->size, lp->stride, lp->pad, lp->init, lp->a, src, dst);
    if (is_any(name, "[depthwise-conv]", "[dw-conv]", NULL, NULL))
        return bcnn_add_depthwise_conv_layer(net, lp->size, lp->stride, lp->pad, 0, lp->init, lp->a, src, dst);
    if (is_any(name, "[activation]", "[nl]", NULL, NULL)) return bcnn_add_activation_layer(net, lp->a, src);
    if (is_any(name, "[batchnorm]", "[bn]", NULL, NULL)) return bcnn_add_batchnorm_layer(net, src, dst);
    if (!strcmp(name, "[lrn]")) return bcnn_add_lrn_layer(net, lp->size, lp->alpha, lp->beta, lp->k, src, dst);
    if (is_any(name, "[connected]", "[fullconnected]", "[fc]", "[ip]"))
        return bcnn_add_fullc_layer(net, lp->outputs, lp->init, lp->a, 0, src, dst);
    if (!strcmp(name, "[softmax]")) return bcnn_add_softmax_layer(net, src, dst);
    if (is_any(name, "[max]", "[maxpool]", NULL, NULL))
        return bcnn_add_maxpool_layer(net, lp->size, lp->stride, lp->padding_type, src, dst);
    if (!strcmp(name, "[avgpool]")) return bcnn_add_avgpool_layer(net, src, dst);
    if (!strcmp(name, "[upsample]")) return bcnn_add_upsample_layer(net, lp->stride, src, dst);
    if (!strcmp(name, "[dropout]")) return bcnn_add_dropout_layer(net, lp->rate, src);
    if (is_any(name, "[concat]", "[route]", NULL, NULL))
        return bcnn_add_concat_layer(net, lp->num_srcs, (char *const *)lp->src_id, dst);
    if (is_any(name, "[eltwise]", "[shortcut]", NULL, NULL)) {
        BCNN_CHECK_AND_LOG(net->log_ctx, lp->num_srcs >= 2, BCNN_INVALID_PARAMETER,
                           "Eltwise layer needs two sources (src=a,b)\n");
        return bcnn_add_eltwise_layer(net, lp->a, lp->src_id[0], lp->src_id[1], dst);
    }
    if (!strcmp(name, "[yolo]")) return bcnn_add_yolo_layer(net, 0, 0, 4, 0, NULL, NULL, src, dst);
    if (!strcmp(name, "[cost]")) return bcnn_add_cost_layer(net, lp->loss, lp->cost, 1.0f, src, "label", dst);
    bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Unknown Layer %s\n", name);
    return BCNN_INVALID_PARAMETER;
}

bcnn_status bcnn_load_net(bcnn_net *net, const char *config_path, const char *model_path) {
    int format = 0;
    if (model_path != NULL) { /* the model file's extension selects the dialect of the config file too */
        const char *dot = strrchr(model_path, '.');
        BCNN_CHECK_AND_LOG(net->log_ctx, dot && dot != model_path, BCNN_INVALID_DATA,
                           "File %s needs to have an extension (.bcnnmodel OR .onnx OR .weights)\n", model_path);
        if (!strcmp(dot + 1, "weights")) format = 1;
        else if (!strcmp(dot + 1, "onnx")) format = 2;
    }
    BCNN_CHECK_AND_LOG(net->log_ctx, format != 2, BCNN_INVALID_MODEL, "ONNX models are not supported by this build\n");
    BCNN_CHECK_AND_LOG(net->log_ctx, config_path, BCNN_INVALID_PARAMETER, "No config file given\n");
    ini_file cfg;
    if (ini_read(config_path, &cfg) != 0) return BCNN_INVALID_PARAMETER;
    bcnn_status st = BCNN_SUCCESS;
    if (cfg.num_sections == 0) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Empty config file %s\n", config_path);
        st = BCNN_INVALID_PARAMETER;
    } else if (strcmp(cfg.sections[0].name, "[net]") != 0 && strcmp(cfg.sections[0].name, "[network]") != 0) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Invalid config file %s: First section must be [net] or [network]\n",
                 config_path);
        st = BCNN_INVALID_PARAMETER;
    } else if (cfg.sections[0].num_keys == 0) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Invalid config file %s: empty section [net]\n", config_path);
        st = BCNN_INVALID_PARAMETER;
    }
    if (st == BCNN_SUCCESS) {
        for (int i = 0; i < cfg.sections[0].num_keys; ++i)
            net_set_param(net, cfg.sections[0].keys[i].name, cfg.sections[0].keys[i].val);
        layer_param lp;
        memset(&lp, 0, sizeof(lp));
        lp_reset(&lp);
        for (int i = 1; st == BCNN_SUCCESS && i < cfg.num_sections; ++i) {
            for (int j = 0; j < cfg.sections[i].num_keys; ++j)
                lp_set(net, i, &lp, cfg.sections[i].keys[j].name, cfg.sections[i].keys[j].val, format);
            if (format == 1) { /* Darknet: implicit chaining of unnamed tensors */
                if (lp.num_srcs == 0) { /* "lid0" for the first layer: its builder takes the net input regardless */
                    lp_set_srcs(&lp, "x");
                    lp_set_lid(&lp.src_id[0], i - 1);
                }
                if (lp.dst_id == NULL) lp_set_lid(&lp.dst_id, i);
            }
            st = add_layer(net, cfg.sections[i].name, &lp);
            lp_reset(&lp);
        }
        lp_reset(&lp);
    }
    ini_free(&cfg);
    if (st != BCNN_SUCCESS) return st;
    if (model_path != NULL) {
        BCNN_INFO(net->log_ctx, "Loading pre-trained model %s\n", model_path);
        BCNN_CHECK_STATUS(bcnn_load_weights(net, model_path));
    }
    return BCNN_SUCCESS;
}
