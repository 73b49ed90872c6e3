/*
 * bcnn_unsupported.c -- entry points of the public API that lie outside the hot path (SURVEY.md
 * section 8: control plane, detection head, rarely used layers; the dataset readers are in bcnn_data.c). They exist so
 * that every consumer of the reference links; each returns BCNN_INVALID_PARAMETER (or does nothing)
 * and says so in the log. INTEGRATION.md lists them.
 */
#include <string.h>

#include "bcnn_internal.h"

#define NOT_BUILT(net, what)                                                                                  \
    do {                                                                                                      \
        bcnn_log((net)->log_ctx, BCNN_LOG_ERROR, "%s is outside the MI355X hot-path build (see INTEGRATION.md)\n", \
                 (what));                                                                                     \
        return BCNN_INVALID_PARAMETER;                                                                        \
    } while (0)

/* uint8 HWC image -> float CHW, reference bcnn_data.c:70-100 */
void bcnn_convert_img_to_float(const uint8_t *src, int w, int h, int c, float norm_coeff, int swap_to_bgr,
                               float mean_r, float mean_g, float mean_b, float *dst) {
    const float m[3] = {mean_r, mean_g, mean_b};
    for (int k = 0; k < c; ++k) {
        const int ks = (swap_to_bgr && c == 3) ? 2 - k : k;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                dst[(size_t)k * w * h + (size_t)y * w + x] =
                    ((float)src[((size_t)y * w + x) * c + ks] - (c == 3 ? m[ks] : m[0])) * norm_coeff;
    }
}

bcnn_status bcnn_fill_tensor_with_image(bcnn_net *net, const uint8_t *src, int w, int h, int c, float norm_coeff,
                                        int swap_to_bgr, float mean_r, float mean_g, float mean_b, int tensor_index,
                                        int batch_index) {
    if (tensor_index < 0 || tensor_index >= net->num_tensors) return BCNN_INVALID_PARAMETER;
    bcnn_tensor *t = &net->tensors[tensor_index];
    if (t->w != w || t->h != h || t->c != c || batch_index < 0 || batch_index >= t->n) return BCNN_INVALID_PARAMETER;
    bcnn_convert_img_to_float(src, w, h, c, norm_coeff, swap_to_bgr, mean_r, mean_g, mean_b,
                              t->data + (size_t)batch_index * w * h * c);
    return bcnn_upload_tensor(net, tensor_index, 0);
}

void bcnn_draw_color_box(unsigned char *img, int w_img, int h_img, float cx, float cy, float w, float h,
                         unsigned char color[3]) {
    (void)img; (void)w_img; (void)h_img; (void)cx; (void)cy; (void)w; (void)h; (void)color;
}

bcnn_output_detection *bcnn_yolo_get_detections(bcnn_net *net, int batch, int width, int height, int netw, int neth,
                                                float thresh, int relative, int *num_dets) {
    (void)net; (void)batch; (void)width; (void)height; (void)netw; (void)neth; (void)thresh; (void)relative;
    if (num_dets) *num_dets = 0;
    return NULL;
}

bcnn_status bcnn_add_deconvolutional_layer(bcnn_net *net, int n, int size, int stride, int pad, bcnn_filler_type init,
                                           bcnn_activation act, const char *s, const char *d) {
    (void)n; (void)size; (void)stride; (void)pad; (void)init; (void)act; (void)s; (void)d;
    NOT_BUILT(net, "deconvolution layer");
}
bcnn_status bcnn_add_lrn_layer(bcnn_net *net, int ls, float a, float b, float k, const char *s, const char *d) {
    (void)ls; (void)a; (void)b; (void)k; (void)s; (void)d;
    NOT_BUILT(net, "LRN layer");
}
bcnn_status bcnn_add_concat_layer(bcnn_net *net, int n, char *const *ids, const char *d) {
    (void)n; (void)ids; (void)d;
    NOT_BUILT(net, "concat layer");
}
bcnn_status bcnn_add_dropout_layer(bcnn_net *net, float rate, const char *id) { (void)rate; (void)id; NOT_BUILT(net, "dropout layer"); }
bcnn_status bcnn_add_upsample_layer(bcnn_net *net, int size, const char *s, const char *d) {
    (void)size; (void)s; (void)d;
    NOT_BUILT(net, "upsample layer");
}
bcnn_status bcnn_add_yolo_layer(bcnn_net *net, int nb, int nc, int coords, int total, int *mask, float *anchors,
                                const char *s, const char *d) {
    (void)nb; (void)nc; (void)coords; (void)total; (void)mask; (void)anchors; (void)s; (void)d;
    NOT_BUILT(net, "YOLOv3 head");
}
