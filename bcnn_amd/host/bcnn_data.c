/*
 * bcnn_data.c -- dataset readers and online augmentation: the caller side of the hot path (what feeds tensors[0] / tensors[1]
 * before bcnn_forward). Reference: src/bcnn_data.c and src/data_loader/bcnn_{mnist,cifar10,classif,regression}_loader.c.
 *
 *   bcnn_set_data_loader      opens the train / test streams of one of the formats below and sizes the sample buffers
 *   bcnn_loader_next          fills one batch on the host, sample by sample, then uploads inputs (+ labels) to the device
 *                             (the reference's H2D hook, bcnn_data.c:413-425)
 *   bcnn_augment_data_with_*  augmentation ranges; bcnn_apply_data_augmentation draws the parameters of one sample from
 *                             libc rand() in the reference's order (flip, shift, scale, rotation, contrast, brightness), so
 *                             that a run seeded like a reference run sees the same samples byte for byte
 *   formats: BCNN_LOAD_MNIST (idx3 images + idx1 labels, big-endian headers), BCNN_LOAD_CIFAR10 (1 + 3072 byte records,
 *            planar RGB), BCNN_LOAD_CLASSIFICATION_LIST ("path label" lines), BCNN_LOAD_REGRESSION_LIST ("path v0 v1 ..").
 *            BCNN_LOAD_DETECTION_LIST belongs to the YOLO head, which is outside this build.
 *
 * Behaviours of the reference that are kept on purpose (each is visible to a consumer that compares runs):
 *   - bcnn_augment_data_with_flip stores its flag in `apply_fliph`, the flip only happens when the INI key `flip_h` has set
 *     `random_fliph` as well, and it is then applied to EVERY sample (no draw);
 *   - bcnn_augment_data_with_distortion stores `distortion`, not `max_distortion`: it enables nothing;
 *   - a shifted or rotated sample is composed over a buffer filled with 128 / 0 respectively;
 *   - the readers wrap around at end of file, and switching to VALID / PREDICT mode rewinds the test streams.
 * Not built: Perlin distortion and random spotlights (INI keys max_distortion / max_spots). Their draws still consume
 * rand() call for call like the reference (parameters, the distortion's own seed, four values per spot), so that the other
 * augmentations stay aligned; the image is left untouched and a warning is printed once.
 */
#include <string.h>

#include <bh/bh_string.h>
#include <bcnn_hip.h>
#include <bip/bip.h>

#include "bcnn_internal.h"

/* ---- augmentation ranges (reference bcnn_data.c:144-209) ---------------------------------------------------------- */
void bcnn_augment_data_with_shift(bcnn_net *net, int width_shift_range, int height_shift_range) {
    if (!net->data_aug) return;
    net->data_aug->range_shift_x = width_shift_range;
    net->data_aug->range_shift_y = height_shift_range;
}
void bcnn_augment_data_with_scale(bcnn_net *net, float min_scale, float max_scale) {
    if (!net->data_aug) return;
    net->data_aug->min_scale = min_scale;
    net->data_aug->max_scale = max_scale;
}
void bcnn_augment_data_with_rotation(bcnn_net *net, float rotation_range) {
    if (net->data_aug) net->data_aug->rotation_range = rotation_range;
}
void bcnn_augment_data_with_flip(bcnn_net *net, int horizontal_flip, int vertical_flip) {
    (void)vertical_flip;
    if (net->data_aug) net->data_aug->apply_fliph = horizontal_flip; /* sic: see the file header */
}
void bcnn_augment_data_with_color_adjustment(bcnn_net *net, int min_brightness, int max_brightness, float min_contrast,
                                             float max_contrast) {
    if (!net->data_aug) return;
    net->data_aug->min_brightness = min_brightness;
    net->data_aug->max_brightness = max_brightness;
    net->data_aug->min_contrast = min_contrast;
    net->data_aug->max_contrast = max_contrast;
}
void bcnn_augment_data_with_blobs(bcnn_net *net, int max_blobs) {
    if (net->data_aug) net->data_aug->max_random_spots = max_blobs;
}
void bcnn_augment_data_with_distortion(bcnn_net *net, float distortion) {
    if (net->data_aug) net->data_aug->distortion = distortion; /* sic */
}

/* uniform integer in [lo, hi] from one rand() draw, rounded to nearest (reference bcnn_utils.h:119-124) */
static int rand_between(int lo, int hi) {
    if (lo > hi) return 0;
    return (int)(((float)rand() / RAND_MAX * (hi - lo)) + lo + 0.5f);
}
/* one draw mapped to [-1/2, 1/2) * range and [0, 1] * (b - a) + a, in float like the reference */
static float rand_centred(float range) { return (float)(rand() - RAND_MAX / 2) / RAND_MAX * range; }
static float rand_span(float a, float b) { return ((float)rand() / RAND_MAX) * (b - a) + a; }

/* One sample, in place. `scratch` (same size as the image) is needed by flip / shift / rotation. */
bcnn_status bcnn_apply_data_augmentation(unsigned char *img, int width, int height, int depth, bcnn_data_augmenter *p,
                                         unsigned char *scratch) {
    const size_t stride = (size_t)width * depth, bytes = stride * height;
    int x_ul = 0, y_ul = 0;
    if (p->random_fliph && p->apply_fliph) {
        bip_fliph_image(img, width, height, depth, stride, scratch, stride);
        memcpy(img, scratch, bytes);
    }
    if (p->range_shift_x || p->range_shift_y) {
        if (p->use_precomputed) {
            x_ul = p->shift_x;
            y_ul = p->shift_y;
        } else {
            x_ul = p->shift_x = (int)rand_centred((float)p->range_shift_x);
            y_ul = p->shift_y = (int)rand_centred((float)p->range_shift_y);
        }
        memset(scratch, 128, bytes);
        bip_crop_image(img, width, height, stride, x_ul, y_ul, scratch, width, height, stride, depth);
        memcpy(img, scratch, bytes);
    }
    if (p->max_scale > 0.0f || p->min_scale > 0.0f) {
        const float scale = p->use_precomputed ? p->scale : (p->scale = rand_span(p->min_scale, p->max_scale));
        const int ws = (int)(width * scale), hs = (int)(height * scale);
        unsigned char *scaled = (unsigned char *)calloc((size_t)ws * hs * depth, 1);
        if (!scaled) return BCNN_FAILED_ALLOC;
        bip_resize_bilinear(img, width, height, stride, scaled, ws, hs, (size_t)ws * depth, depth);
        /* cropped back at the shift's origin, over the unscaled image */
        bip_crop_image(scaled, ws, hs, (size_t)ws * depth, x_ul, y_ul, img, width, height, stride, depth);
        free(scaled);
    }
    if (p->rotation_range > 0.0f) {
        const float theta =
            p->use_precomputed ? p->rotation : (p->rotation = bip_deg2rad(rand_centred(p->rotation_range)));
        memset(scratch, 128, bytes);
        bip_rotate_image(img, width, height, stride, scratch, width, height, stride, depth, theta, width / 2, height / 2,
                         BILINEAR);
        memcpy(img, scratch, bytes);
    }
    if (p->min_contrast > 0.0f || p->max_contrast > 0.0f) {
        const float c = p->use_precomputed ? p->contrast : (p->contrast = rand_span(p->min_contrast, p->max_contrast));
        bip_contrast_stretch(img, stride, width, height, depth, img, stride, c);
    }
    if (p->min_brightness != 0 || p->max_brightness != 0) {
        const int b = p->use_precomputed
                          ? p->brightness
                          : (p->brightness = (int)rand_span((float)p->min_brightness, (float)p->max_brightness));
        bip_image_brightness(img, stride, width, height, depth, img, stride, b);
    }
    /* Not built: Perlin distortion and random spotlights. Their draws are consumed exactly as the reference consumes them,
     * so that the samples behind this one see the generator in the reference's state: three parameters (when not
     * precomputed) plus the seed bip_image_perlin_distortion draws for itself on every call (bip.c:212); the spot count
     * plus four values per spot (mu_x, mu_y, sigma_x, sigma_y: bip.c:294-298). */
    if (p->max_distortion > 0.0f) {
        if (!p->use_precomputed) {
            p->distortion_kx = ((float)rand() - RAND_MAX / 2) / RAND_MAX;
            p->distortion_ky = ((float)rand() - RAND_MAX / 2) / RAND_MAX;
            p->distortion = ((float)rand() / RAND_MAX) * p->max_distortion;
        }
        (void)rand();
    }
    if (p->max_random_spots > 0) {
        const int spots = rand_between(0, p->max_random_spots);
        for (int i = 0; i < 4 * spots; ++i) (void)rand();
    }
    static int warned = 0;
    if ((p->max_distortion > 0.0f || p->max_random_spots > 0) && !warned) {
        warned = 1;
        fprintf(stderr, "[bcnn] max_distortion / max_spots: Perlin distortion and random spotlights are not built; samples keep "
                        "their pixels (the random stream stays aligned with the reference)\n");
    }
    return BCNN_SUCCESS;
}

/* ---- streams ------------------------------------------------------------------------------------------------------ */
static FILE *open_stream(bcnn_net *net, const char *path, bcnn_status *st) {
    if (!path) return NULL;
    FILE *f = fopen(path, "rb");
    if (!f) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Could not open file %s\n", path);
        *st = BCNN_INVALID_PARAMETER;
    }
    return f;
}

/* TRAIN mode reads the train streams, every other mode the test streams (reference bcnn_data.c:493-543) */
static bcnn_status select_streams(bcnn_net *net, bcnn_loader *it, int rewind_test) {
    const int train = net->mode == BCNN_MODE_TRAIN;
    if (!train && rewind_test) { /* every evaluation run sees the same samples */
        if (!it->f_test || fseek(it->f_test, 0L, SEEK_SET) != 0) {
            bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Could not rewind test dataset file\n");
            return BCNN_INVALID_DATA;
        }
        if (it->has_extra_data && (!it->f_test_extra || fseek(it->f_test_extra, 0L, SEEK_SET) != 0)) {
            bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Could not rewind extra test dataset file\n");
            return BCNN_INVALID_DATA;
        }
    }
    it->f_current = train ? it->f_train : it->f_test;
    it->f_current_extra = it->has_extra_data ? (train ? it->f_train_extra : it->f_test_extra) : NULL;
    if (!it->f_current || (it->has_extra_data && !it->f_current_extra)) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "A %s dataset must be provided\n", train ? "training" : "testing");
        return BCNN_INVALID_DATA;
    }
    return BCNN_SUCCESS;
}

bcnn_status bcnn_open_dataset(bcnn_loader *it, bcnn_net *net, const char *train_path, const char *train_path_extra,
                              const char *test_path, const char *test_path_extra, bool has_extra) {
    bcnn_status st = BCNN_SUCCESS;
    it->f_train = open_stream(net, train_path, &st);
    it->f_test = open_stream(net, test_path, &st);
    if (has_extra) {
        it->f_train_extra = open_stream(net, train_path_extra, &st);
        it->f_test_extra = open_stream(net, test_path_extra, &st);
    }
    if (st != BCNN_SUCCESS) return st;
    it->has_extra_data = has_extra;
    return select_streams(net, it, 0);
}

bcnn_status bcnn_switch_data_handles(bcnn_net *net, bcnn_loader *it) { return select_streams(net, it, 1); }

/* at end of file start over; otherwise stay where we are (the readers peek one byte) */
static void wrap_at_eof(FILE *f) {
    unsigned char probe;
    if (fread(&probe, 1, 1, f) == 0) rewind(f);
    else fseek(f, -1, SEEK_CUR);
}

/* ---- sample -> tensors ---------------------------------------------------------------------------------------------- */
static int needs_scratch(const bcnn_data_augmenter *a, int list_loader) {
    if (a->range_shift_x != 0 || a->range_shift_y != 0 || a->random_fliph != 0) return 1;
    if (list_loader) return a->rotation_range > 0.0f || a->max_random_spots > 0 || a->max_distortion > 0.0f;
    return a->rotation_range != 0;
}

static bcnn_status augment_if_training(bcnn_net *net, unsigned char *img, int w, int h, int c, int list_loader) {
    if (net->mode != BCNN_MODE_TRAIN || !net->data_aug) return BCNN_SUCCESS;
    unsigned char *scratch = NULL;
    if (needs_scratch(net->data_aug, list_loader)) {
        scratch = (unsigned char *)calloc((size_t)w * h * c, 1);
        if (!scratch) return BCNN_FAILED_ALLOC;
    }
    const bcnn_status st = bcnn_apply_data_augmentation(img, w, h, c, net->data_aug, scratch);
    free(scratch);
    return st;
}

/* the stored-size sample (MNIST / CIFAR) into input slot idx: centre crop when the net input is smaller, [-1, 1] floats */
static void sample_to_input(bcnn_net *net, bcnn_loader *it, int idx) {
    bcnn_tensor *in = &net->tensors[0];
    float *x = in->data + (size_t)idx * bcnn_tensor_size3d(in);
    const unsigned char *img = it->input_uchar;
    if (in->w < it->input_width || in->h < it->input_height) {
        bip_crop_image(it->input_uchar, it->input_width, it->input_height, (size_t)it->input_width * it->input_depth,
                       (it->input_width - in->w) / 2, (it->input_height - in->h) / 2, it->input_net, in->w, in->h,
                       (size_t)in->w * in->c, in->c);
        img = it->input_net;
    }
    bcnn_convert_img_to_float(img, in->w, in->h, in->c, 1 / 127.5f, 0, 127.5f, 127.5f, 127.5f, x);
}

static float *label_slot(bcnn_net *net, int idx, int *label_sz) {
    bcnn_tensor *lab = &net->tensors[1];
    *label_sz = bcnn_tensor_size3d(lab);
    float *y = lab->data + (size_t)idx * *label_sz;
    memset(y, 0, (size_t)*label_sz * sizeof(float));
    return y;
}

static bcnn_status check_input_shape(bcnn_net *net) {
    if (net->tensors[0].w > 0 && net->tensors[0].h > 0 && net->tensors[0].c > 0) return BCNN_SUCCESS;
    bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Input's width, height and channels must be > 0\n");
    return BCNN_INVALID_PARAMETER;
}

/* ---- MNIST: idx3-ubyte images + idx1-ubyte labels (bcnn_mnist_loader.c) ------------------------------------------- */
static uint32_t be32(const unsigned char *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static bcnn_status mnist_header(bcnn_net *net, bcnn_loader *it) {
    unsigned char h[16];
    if (fread(h, 1, 16, it->f_current) != 16) goto corrupt;
    const uint32_t images = be32(h + 4);
    it->input_height = (int)be32(h + 8);
    it->input_width = (int)be32(h + 12);
    /* the sample buffer is sized from these two and the network input is cut out of it: neither can be empty or absurd, and
     * the input cannot be larger than the stored sample (the reference reads past the sample there) */
    if (it->input_height < 1 || it->input_width < 1 || it->input_height > 4096 || it->input_width > 4096 ||
        net->tensors[0].h > it->input_height || net->tensors[0].w > it->input_width) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Mnist header: %d x %d samples do not fit the %d x %d network input\n",
                 it->input_width, it->input_height, net->tensors[0].w, net->tensors[0].h);
        return BCNN_INVALID_DATA;
    }
    if (fread(h, 1, 8, it->f_current_extra) != 8) goto corrupt;
    if (images != be32(h + 4)) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR,
                 "Inconsistent MNIST data: number of images and labels must be the same\n");
        return BCNN_INVALID_DATA;
    }
    return BCNN_SUCCESS;
corrupt:
    bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Corrupted Mnist data\n");
    return BCNN_INVALID_DATA;
}

static bcnn_status mnist_init(bcnn_loader *it, bcnn_net *net, const char *a, const char *b, const char *c, const char *d) {
    BCNN_CHECK_STATUS(bcnn_open_dataset(it, net, a, b, c, d, true));
    BCNN_CHECK_STATUS(check_input_shape(net));
    BCNN_CHECK_STATUS(mnist_header(net, it));
    it->input_depth = 1;
    it->input_uchar = (uint8_t *)calloc((size_t)it->input_width * it->input_height, 1);
    it->input_net = (uint8_t *)calloc((size_t)bcnn_tensor_size3d(&net->tensors[0]), 1);
    rewind(it->f_current);
    rewind(it->f_current_extra);
    return (it->input_uchar && it->input_net) ? BCNN_SUCCESS : BCNN_FAILED_ALLOC;
}

static bcnn_status mnist_next(bcnn_loader *it, bcnn_net *net, int idx) {
    wrap_at_eof(it->f_current);
    wrap_at_eof(it->f_current_extra);
    if (ftell(it->f_current) == 0 && ftell(it->f_current_extra) == 0) BCNN_CHECK_STATUS(mnist_header(net, it));
    unsigned char label;
    const size_t sz = (size_t)it->input_width * it->input_height;
    if (fread(&label, 1, 1, it->f_current_extra) != 1 || fread(it->input_uchar, 1, sz, it->f_current) != sz) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Corrupted Mnist data\n");
        return BCNN_INVALID_DATA;
    }
    BCNN_CHECK_STATUS(augment_if_training(net, it->input_uchar, it->input_width, it->input_height, it->input_depth, 0));
    sample_to_input(net, it, idx);
    if (net->mode != BCNN_MODE_PREDICT) {
        int n;
        float *y = label_slot(net, idx, &n);
        if ((int)label < n) y[label] = 1;
    }
    return BCNN_SUCCESS;
}

/* ---- CIFAR-10 binary batches: {label, 1024 R, 1024 G, 1024 B} records (bcnn_cifar10_loader.c) --------------------- */
static bcnn_status cifar10_init(bcnn_loader *it, bcnn_net *net, const char *a, const char *b, const char *c,
                                const char *d) {
    BCNN_CHECK_STATUS(bcnn_open_dataset(it, net, a, b, c, d, false));
    it->input_width = it->input_height = 32;
    it->input_depth = 3;
    it->input_uchar = (uint8_t *)calloc(32 * 32 * 3, 1);
    BCNN_CHECK_STATUS(check_input_shape(net));
    it->input_net = (uint8_t *)calloc((size_t)bcnn_tensor_size3d(&net->tensors[0]), 1);
    return (it->input_uchar && it->input_net) ? BCNN_SUCCESS : BCNN_FAILED_ALLOC;
}

static bcnn_status cifar10_next(bcnn_loader *it, bcnn_net *net, int idx) {
    unsigned char rec[1 + 3072];
    wrap_at_eof(it->f_current);
    if (fread(rec, 1, sizeof(rec), it->f_current) != sizeof(rec)) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Corrupted Cifar data\n");
        return BCNN_INVALID_DATA;
    }
    for (int k = 0; k < 3; ++k) /* planar -> interleaved */
        for (int p = 0; p < 1024; ++p) it->input_uchar[p * 3 + k] = rec[1 + k * 1024 + p];
    BCNN_CHECK_STATUS(augment_if_training(net, it->input_uchar, 32, 32, 3, 0));
    sample_to_input(net, it, idx);
    if (net->mode != BCNN_MODE_PREDICT) {
        int n;
        float *y = label_slot(net, idx, &n);
        if ((int)rec[0] < n) y[rec[0]] = 1;
    }
    return BCNN_SUCCESS;
}

/* ---- list files: one "image-path label..." line per sample (bcnn_classif_loader.c, bcnn_regression_loader.c) ----- */
static bcnn_status list_init(bcnn_loader *it, bcnn_net *net, const char *a, const char *b, const char *c, const char *d) {
    BCNN_CHECK_STATUS(bcnn_open_dataset(it, net, a, b, c, d, false));
    BCNN_CHECK_STATUS(check_input_shape(net));
    it->input_uchar = (uint8_t *)calloc((size_t)bcnn_tensor_size3d(&net->tensors[0]), 1);
    return it->input_uchar ? BCNN_SUCCESS : BCNN_FAILED_ALLOC;
}

/* decodes the file, crops it to the net input (centred for evaluation, at a random origin for training) */
static bcnn_status load_image(bcnn_net *net, char *path, int w, int h, int c, unsigned char *img, int *x_shift,
                              int *y_shift) {
    int wi = 0, hi = 0, ci = 0, x_ul = 0, y_ul = 0;
    unsigned char *buf = NULL;
    bip_load_image(path, &buf, &wi, &hi, &ci);
    if (!(wi > 0 && hi > 0 && buf)) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Invalid image %s\n", path);
        free(buf);
        return BCNN_INVALID_DATA;
    }
    if (c != ci) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Unexpected number of channels of image %s\n", path);
        free(buf);
        return BCNN_INVALID_DATA;
    }
    if (wi != w || hi != h) {
        if (net->mode == BCNN_MODE_TRAIN) {
            x_ul = rand_between(0, wi - w);
            y_ul = rand_between(0, hi - h);
        } else {
            x_ul = (wi - w) / 2;
            y_ul = (hi - h) / 2;
        }
        unsigned char *crop = (unsigned char *)calloc((size_t)w * h * c, 1);
        if (!crop) { free(buf); return BCNN_FAILED_ALLOC; }
        bip_crop_image(buf, wi, hi, (size_t)wi * ci, x_ul, y_ul, crop, w, h, (size_t)w * c, c);
        memcpy(img, crop, (size_t)w * h * c);
        free(crop);
    } else {
        memcpy(img, buf, (size_t)w * h * c);
    }
    free(buf);
    if (x_shift && y_shift) { *x_shift = x_ul; *y_shift = y_ul; }
    return BCNN_SUCCESS;
}

/* reference bcnn_fill_input_tensor (bcnn_data.c:336-377): a sample that fails to decode leaves the buffer as it was */
void bcnn_fill_input_tensor(bcnn_net *net, bcnn_loader *it, char *path_img, int idx) {
    bcnn_tensor *in = &net->tensors[0];
    load_image(net, path_img, in->w, in->h, in->c, it->input_uchar, net->data_aug ? &net->data_aug->shift_x : NULL,
               net->data_aug ? &net->data_aug->shift_y : NULL);
    if (net->data_aug) augment_if_training(net, it->input_uchar, in->w, in->h, in->c, 1);
    bcnn_convert_img_to_float(it->input_uchar, in->w, in->h, in->c, 1 / 127.5f, net->data_aug ? net->data_aug->swap_to_bgr : 0,
                              127.5f, 127.5f, 127.5f, in->data + (size_t)idx * bcnn_tensor_size3d(in));
}

static void free_tokens(char **tok, int n) {
    for (int i = 0; i < n; ++i) free(tok[i]);
    free(tok);
}

/* next line split at blanks, wrapping around once at end of file (reference bh_fsplitline) */
static int next_line_tokens(FILE *f, char ***tok) {
    char *line = bh_fgetline(f);
    if (!line) {
        rewind(f);
        line = bh_fgetline(f);
        if (!line) return 0;
    }
    char **t = NULL;
    const int n = bh_strsplit(line, ' ', &t);
    free(line);
    if (!t || n == 0) { free(t); return 0; }
    *tok = t;
    return n;
}

static bcnn_status list_classif_next(bcnn_loader *it, bcnn_net *net, int idx) {
    char **tok = NULL;
    const int n = next_line_tokens(it->f_current, &tok);
    if (n <= 0) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Invalid regression format\n"); /* the reference's wording */
        return BCNN_INVALID_DATA;
    }
    if (net->mode != BCNN_MODE_PREDICT && n != 2) {
        bcnn_log(net->log_ctx, BCNN_LOG_WARNING, "Unexpected classif format. Found label size of %d, expected %d.\n", n - 1, 1);
        free_tokens(tok, n);
        return BCNN_INVALID_DATA;
    }
    bcnn_fill_input_tensor(net, it, tok[0], idx);
    if (net->mode != BCNN_MODE_PREDICT) {
        int sz;
        float *y = label_slot(net, idx, &sz);
        const int cls = atoi(tok[1]);
        if (cls >= 0 && cls < sz) y[cls] = 1;
    }
    free_tokens(tok, n);
    return BCNN_SUCCESS;
}

static bcnn_status list_reg_next(bcnn_loader *it, bcnn_net *net, int idx) {
    char **tok = NULL;
    const int n = next_line_tokens(it->f_current, &tok);
    if (n <= 0) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "Invalid regression format\n");
        return BCNN_INVALID_DATA;
    }
    bcnn_fill_input_tensor(net, it, tok[0], idx);
    if (net->mode != BCNN_MODE_PREDICT) {
        int sz;
        float *y = label_slot(net, idx, &sz);
        if (n - 1 != sz)
            bcnn_log(net->log_ctx, BCNN_LOG_WARNING, "Unexpected label format. Found label size of %d, expected %d.\n", n - 1, sz);
        for (int i = 0; i < n - 1 && i < sz; ++i) y[i] = (float)atof(tok[i + 1]);
    }
    free_tokens(tok, n);
    return BCNN_SUCCESS;
}

/* ---- public entry points ------------------------------------------------------------------------------------------ */
static void loader_close(bcnn_loader *it) {
    FILE **fs[4] = {&it->f_train, &it->f_train_extra, &it->f_test, &it->f_test_extra};
    for (int i = 0; i < 4; ++i)
        if (*fs[i]) { fclose(*fs[i]); *fs[i] = NULL; }
    free(it->input_uchar);
    free(it->input_net);
    it->input_uchar = it->input_net = NULL;
}

void bcnn_destroy_data_loader(bcnn_net *net) {
    if (!net->data_loader) return;
    loader_close(net->data_loader);
    free(net->data_loader);
    net->data_loader = NULL;
}

bcnn_status bcnn_set_data_loader(bcnn_net *net, bcnn_loader_type type, const char *train_path_data,
                                 const char *train_path_extra, const char *test_path_data, const char *test_path_extra) {
    bcnn_destroy_data_loader(net);
    if (type == BCNN_LOAD_DETECTION_LIST || (int)type < 0 || (int)type >= BCNN_NUM_LOADERS) {
        bcnn_log(net->log_ctx, BCNN_LOG_ERROR,
                 "bcnn_set_data_loader: the detection-list format belongs to the YOLO head, which is outside the MI355X "
                 "hot-path build (see INTEGRATION.md)\n");
        return BCNN_INVALID_PARAMETER;
    }
    bcnn_loader *it = (bcnn_loader *)calloc(1, sizeof(bcnn_loader));
    if (!it) return BCNN_FAILED_ALLOC;
    it->type = type;
    net->data_loader = it;
    bcnn_status st;
    switch (type) {
        case BCNN_LOAD_MNIST: st = mnist_init(it, net, train_path_data, train_path_extra, test_path_data, test_path_extra); break;
        case BCNN_LOAD_CIFAR10: st = cifar10_init(it, net, train_path_data, train_path_extra, test_path_data, test_path_extra); break;
        default: st = list_init(it, net, train_path_data, train_path_extra, test_path_data, test_path_extra); break;
    }
    if (st != BCNN_SUCCESS) bcnn_destroy_data_loader(net); /* no half-opened loader for a later bcnn_loader_next to trip over */
    return st;
}

/* One batch: samples on the host, then the reference's host -> device hook (bcnn_data.c:398-427). A sample that cannot be
 * read is skipped and the next one takes its slot like in the reference -- but after 1000 failures in a row the batch is
 * given up (the reference retries forever, e.g. on a truncated file). Without a loader the caller has filled the host
 * tensors itself and only the upload happens. */
bcnn_status bcnn_loader_next(bcnn_net *net) {
    bcnn_loader *it = net->data_loader;
    if (it) {
        int failures = 0;
        for (int i = 0; i < net->batch_size; ++i) {
            bcnn_status st;
            switch (it->type) {
                case BCNN_LOAD_MNIST: st = mnist_next(it, net, i); break;
                case BCNN_LOAD_CIFAR10: st = cifar10_next(it, net, i); break;
                case BCNN_LOAD_CLASSIFICATION_LIST: st = list_classif_next(it, net, i); break;
                default: st = list_reg_next(it, net, i); break;
            }
            if (st != BCNN_SUCCESS) {
                if (++failures >= 1000) {
                    bcnn_log(net->log_ctx, BCNN_LOG_ERROR, "bcnn_loader_next: 1000 samples in a row could not be read\n");
                    return BCNN_INVALID_DATA;
                }
                --i;
                continue;
            }
            failures = 0;
        }
    }
    for (int i = 0; i < net->num_inputs; ++i) {
        bcnn_tensor *t = &net->tensors[net->inputs[i]];
        if (t->data && t->data_gpu) bcnn_hip_memcpy_h2d(t->data_gpu, t->data, (size_t)bcnn_tensor_size(t) * sizeof(float));
    }
    bcnn_tensor *lab = &net->tensors[1];
    if (net->mode != BCNN_MODE_PREDICT && lab->data && lab->data_gpu)
        bcnn_hip_memcpy_h2d(lab->data_gpu, lab->data, (size_t)bcnn_tensor_size(lab) * sizeof(float));
    return BCNN_SUCCESS;
}
