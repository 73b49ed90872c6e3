/* bip/bip.h -- the slice of the reference's image library that unchanged consumers of the public API call
 * (src/cli/bcnn_cl.c dumps detection overlays with bip_write_image). Own minimal implementation in
 * bcnn_amd/host/bip_min.c (libbip.so): 8-bit grey / RGB(A) PNG writer with stored (uncompressed) deflate
 * blocks. Image processing (resize, rotation, ... used by the data augmenter) is out of scope. */
#ifndef BIP_H
#define BIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { BIP_SUCCESS, BIP_INVALID_PTR, BIP_INVALID_SIZE, BIP_INVALID_PARAMETER, BIP_UNKNOWN_ERROR } bip_status;

/* writes `src` (src_height rows of src_stride bytes, src_depth = 1, 3 or 4 interleaved channels) as a PNG file */
bip_status bip_write_image(char *filename, uint8_t *src, int32_t src_width, int32_t src_height, int32_t src_depth,
                           int32_t src_stride);
#ifdef __cplusplus
}
#endif
#endif
