/* bip_augment.c -- libbip.so: the image operations the online data augmenter applies to a sample (reference
 * src/bcnn_data.c:211-334 calls them in this order: horizontal flip, shift = crop with a negative origin, scale, rotation,
 * contrast, brightness). Each one reproduces the reference's integer / float arithmetic step by step so that an augmented
 * sample is the same byte for byte (tests/test_data_loader.py compares against the compiled reference):
 *   bip_crop_image        src/bip/src/bip.c:319-346   (row copies between two rectangles, negative origins allowed)
 *   bip_fliph_image       :1309-1325
 *   bip_rotate_image      :1202-1291  (16.16 fixed-point inverse map, float bilinear blend, 0 outside the source)
 *   bip_contrast_stretch  :85-129     (around the per-channel integer mean, 20.12 fixed-point gain)
 *   bip_image_brightness  :131-150 */
#include <math.h>
#include <string.h>

#include "bip/bip.h"

static int32_t clamp_u8(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

bip_status bip_crop_image(uint8_t *src, size_t src_width, size_t src_height, size_t src_stride, int32_t x_ul, int32_t y_ul,
                          uint8_t *dst, size_t dst_width, size_t dst_height, size_t dst_stride, size_t depth) {
    if (!src || !dst) return BIP_INVALID_PTR;
    /* the copied rectangle starts at (sx, sy) in the source and lands at (dx, dy) in the destination */
    const uint32_t dx = x_ul < 0 ? (uint32_t)(-x_ul) : 0u, dy = y_ul < 0 ? (uint32_t)(-y_ul) : 0u;
    const uint32_t sx = x_ul > 0 ? (uint32_t)x_ul : 0u, sy = y_ul > 0 ? (uint32_t)y_ul : 0u;
    /* bytes per row and rows: the reference computes both in size_t (an origin beyond the row end wraps to a huge value
     * and the destination room wins), then narrows to 32 bits */
    const size_t src_room = src_stride - (size_t)sx * depth, dst_room = dst_stride - (size_t)dx * depth;
    const uint32_t row_bytes = (uint32_t)(src_room > dst_room ? dst_room : src_room);
    const size_t rows_d = dst_height - dy, rows_s = src_height - sy;
    const uint32_t rows = (uint32_t)(rows_d < rows_s ? rows_d : rows_s);
    if (dy > dst_height || dx > dst_width || sx > src_width || sy > src_height) return BIP_SUCCESS; /* nothing overlaps */
    const uint8_t *s = src + (size_t)sy * src_stride + (size_t)sx * depth;
    uint8_t *d = dst + (size_t)dy * dst_stride + (size_t)dx * depth;
    for (uint32_t y = 0; y < rows; ++y, s += src_stride, d += dst_stride) memcpy(d, s, row_bytes);
    return BIP_SUCCESS;
}

bip_status bip_fliph_image(uint8_t *src, size_t width, size_t height, size_t depth, size_t src_stride, uint8_t *dst,
                           size_t dst_stride) {
    for (size_t y = 0; y < height; ++y) {
        const uint8_t *s = src + y * src_stride;
        uint8_t *d = dst + y * dst_stride;
        for (size_t x = 0; x < width; ++x) memcpy(d + x * depth, s + (width - 1 - x) * depth, depth);
    }
    return BIP_SUCCESS;
}

bip_status bip_rotate_image(uint8_t *src, size_t src_width, size_t src_height, size_t src_stride, uint8_t *dst,
                            size_t dst_width, size_t dst_height, size_t dst_stride, size_t depth, float angle,
                            int32_t center_x, int32_t center_y, bip_interpolation interpolation) {
    (void)src_stride; /* like the reference, the source is addressed as a dense src_width x depth image */
    if (src_width == 0 || src_height == 0 || dst_width == 0 || dst_height == 0) return BIP_INVALID_SIZE;
    if (!src || !dst) return BIP_INVALID_PTR;
    if (interpolation != NEAREST_NEIGHBOR && interpolation != BILINEAR) return BIP_SUCCESS;
    const int32_t ca = (int32_t)(cos(angle) * 65536), sa = (int32_t)(sin(angle) * 65536);
    const int32_t cx16 = center_x << 16, cy16 = center_y << 16;
    const int32_t sw = (int32_t)src_width, sh = (int32_t)src_height;
    for (size_t y = 0; y < dst_height; ++y) {
        const int32_t v = (int32_t)y - center_y;
        uint8_t *row = dst + y * dst_stride;
        for (size_t x = 0; x < dst_width; ++x) {
            const int32_t u = (int32_t)x - center_x;
            /* 16.16 source position of this destination pixel (32-bit wrap-around like the reference) */
            const int32_t px = (int32_t)((uint32_t)ca * (uint32_t)u - (uint32_t)sa * (uint32_t)v + (uint32_t)cx16);
            const int32_t py = (int32_t)((uint32_t)sa * (uint32_t)u + (uint32_t)ca * (uint32_t)v + (uint32_t)cy16);
            uint8_t *out = row + x * depth;
            if (interpolation == NEAREST_NEIGHBOR) {
                const int32_t mx = (int32_t)((uint32_t)px + 32768u) >> 16, my = (int32_t)((uint32_t)py + 32768u) >> 16;
                if (mx >= 0 && mx < sw - 1 && my >= 0 && my < sh - 1)
                    memcpy(out, src + ((size_t)my * src_width + (size_t)mx) * depth, depth);
                else
                    memset(out, 0, depth);
                continue;
            }
            const int32_t mx = px >> 16, my = py >> 16;
            if (!(mx >= 0 && mx < sw - 1 && my >= 0 && my < sh - 1)) {
                memset(out, 0, depth);
                continue;
            }
            const float fx = (float)(px - (mx << 16)) / 65536, fy = (float)(py - (my << 16)) / 65536;
            const uint8_t *p00 = src + ((size_t)my * src_width + (size_t)mx) * depth;
            const uint8_t *p01 = p00 + depth, *p10 = p00 + src_width * depth, *p11 = p10 + depth;
            for (size_t k = 0; k < depth; ++k) {
                /* four products of three factors each, summed left to right in float; truncated to 8 bits */
                const float level = (float)p00[k] * (1 - fx) * (1 - fy) + (float)p01[k] * (fx) * (1 - fy) +
                                    (float)p10[k] * (1 - fx) * (fy) + (float)p11[k] * (fx) * (fy);
                out[k] = (uint8_t)level;
            }
        }
    }
    return BIP_SUCCESS;
}

bip_status bip_contrast_stretch(uint8_t *src, size_t src_stride, size_t width, size_t height, size_t depth, uint8_t *dst,
                                size_t dst_stride, float contrast) {
    if (width == 0 || height == 0) return BIP_INVALID_SIZE;
    if (!src || !dst) return BIP_INVALID_PTR;
    const int32_t gain = (int32_t)(contrast * (1 << 12) + 0.5);
    uint32_t mean[8] = {0};
    if (depth > 8) return BIP_INVALID_PARAMETER;
    for (size_t y = 0; y < height; ++y)
        for (size_t x = 0; x < width; ++x)
            for (size_t d = 0; d < depth; ++d) mean[d] += src[y * src_stride + depth * x + d];
    for (size_t d = 0; d < depth; ++d) mean[d] = (uint32_t)(mean[d] / (width * height));
    for (size_t y = 0; y < height; ++y)
        for (size_t x = 0; x < width; ++x)
            for (size_t d = 0; d < depth; ++d) {
                const int32_t centred = (int32_t)src[y * src_stride + depth * x + d] - (int32_t)mean[d];
                const int32_t pix = ((centred * gain + (1 << 11)) >> 12) + (int32_t)mean[d];
                dst[y * dst_stride + depth * x + d] = (uint8_t)clamp_u8(pix);
            }
    return BIP_SUCCESS;
}

bip_status bip_image_brightness(uint8_t *src, size_t src_stride, size_t width, size_t height, size_t depth, uint8_t *dst,
                                size_t dst_stride, int32_t brightness) {
    if (width == 0 || height == 0) return BIP_INVALID_SIZE;
    if (!src || !dst) return BIP_INVALID_PTR;
    for (size_t y = 0; y < height; ++y)
        for (size_t x = 0; x < width * depth; ++x)
            dst[y * dst_stride + x] = (uint8_t)clamp_u8((int32_t)src[y * src_stride + x] + brightness);
    return BIP_SUCCESS;
}
