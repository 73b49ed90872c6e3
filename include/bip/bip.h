/* bip/bip.h -- the slice of the reference's image library (src/bip/inc/bip/bip.h) that unchanged consumers of
 * the public API call: src/cli/bcnn_cl.c dumps detection overlays with bip_write_image (:1872);
 * examples/inference_benchmark loads its test image with bip_load_image (:530) and fits it to the net input
 * with bip_resize_bilinear (:338). Own dependency-free implementations in bcnn_amd/host/bip_min.c and
 * bip_decode.c (libbip.so): PNG writer (stored deflate blocks), PNG / PNM / BMP reader with a full inflate,
 * and the reference's fixed-point bilinear resize (bit-identical, tests/test_bip.py). The rest of the image
 * processing (rotation, colour augmentation, ... used by the data augmenter) is out of scope. */
#ifndef BIP_H
#define BIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { BIP_SUCCESS, BIP_INVALID_PTR, BIP_INVALID_SIZE, BIP_INVALID_PARAMETER, BIP_UNKNOWN_ERROR } bip_status;

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
#ifdef __cplusplus
}
#endif
#endif
