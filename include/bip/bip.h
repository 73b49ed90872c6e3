/* bip/bip.h -- the slice of the reference's image library (src/bip/inc/bip/bip.h) that unchanged consumers of
 * the public API call: src/cli/bcnn_cl.c dumps detection overlays with bip_write_image (:1872);
 * examples/inference_benchmark loads its test image with bip_load_image (:530) and fits it to the net input
 * with bip_resize_bilinear (:338). Own dependency-free implementations in bcnn_amd/host/bip_min.c and
 * bip_decode.c / bip_jpeg.c (libbip.so): PNG writer (stored deflate blocks), JPEG / PNG / PNM / BMP reader (JPEG with the
 * reference's -- stb_image's -- exact pixels, PNG with a full inflate),
 * and the reference's fixed-point bilinear resize (bit-identical, tests/test_bip.py); bip_augment.c holds the operations
 * of the online data augmenter (crop / shift, horizontal flip, rotation, contrast, brightness: bcnn_data.c:211-334),
 * byte for byte like the reference (tests/test_data_loader.py). Perlin distortion and random spotlights are not built. */
#ifndef BIP_H
#define BIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { BIP_SUCCESS, BIP_INVALID_PTR, BIP_INVALID_SIZE, BIP_INVALID_PARAMETER, BIP_UNKNOWN_ERROR } bip_status;
typedef enum { NEAREST_NEIGHBOR, BILINEAR } bip_interpolation;
#define bip_deg2rad(x) (x) * 0.01745329252f

/* writes `src` (src_height rows of src_stride bytes, src_depth = 1, 3 or 4 interleaved channels) as a PNG file */
bip_status bip_write_image(char *filename, uint8_t *src, int32_t src_width, int32_t src_height, int32_t src_depth,
                           int32_t src_stride);
/* decodes an image file / buffer into a malloc'ed interleaved 8-bit image with the file's own channel count
 * (reference: stbi_load(..., 0)); the caller frees *src */
bip_status bip_load_image(char *filename, uint8_t **src, int32_t *src_width, int32_t *src_height, int32_t *src_depth);
bip_status bip_load_image_from_memory(unsigned char *buffer, int buffer_size, uint8_t **src, int32_t *src_width,
                                      int32_t *src_height, int32_t *src_depth);
/* bilinear resize with half-pixel centres and 4-bit fixed-point weights per axis, depth 1..4 interleaved
 * channels, strides in bytes (reference src/bip/src/bip.c:1077-1200) */
bip_status bip_resize_bilinear(uint8_t *src, size_t src_width, size_t src_height, size_t src_stride, uint8_t *dst,
                               size_t dst_width, size_t dst_height, size_t dst_stride, size_t depth);
/* ---- the data augmenter's operations (interleaved 8-bit images, strides in bytes) ---- */
/* copies the overlap of the source rectangle at (x_ul, y_ul) into dst; negative origins shift the image right / down and
 * leave the uncovered part of dst as it was (reference bip.h:219) */
bip_status bip_crop_image(uint8_t *src, size_t src_width, size_t src_height, size_t src_stride, int32_t x_ul, int32_t y_ul,
                          uint8_t *dst, size_t dst_width, size_t dst_height, size_t dst_stride, size_t depth);
bip_status bip_fliph_image(uint8_t *src, size_t width, size_t height, size_t depth, size_t src_stride, uint8_t *dst,
                           size_t dst_stride);
/* rotation by `angle` radians around (center_x, center_y); pixels that map outside the source become 0 (bip.h:361) */
bip_status bip_rotate_image(uint8_t *src, size_t src_width, size_t src_height, size_t src_stride, uint8_t *dst,
                            size_t dst_width, size_t dst_height, size_t dst_stride, size_t depth, float angle,
                            int32_t center_x, int32_t center_y, bip_interpolation interpolation);
bip_status bip_contrast_stretch(uint8_t *src, size_t src_stride, size_t width, size_t height, size_t depth, uint8_t *dst,
                                size_t dst_stride, float contrast);
bip_status bip_image_brightness(uint8_t *src, size_t src_stride, size_t width, size_t height, size_t depth, uint8_t *dst,
                                size_t dst_stride, int32_t brightness);
#ifdef __cplusplus
}
#endif
#endif
