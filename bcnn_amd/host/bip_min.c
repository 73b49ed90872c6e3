/* bip_min.c -- libbip.so: bip_write_image as a dependency-free PNG writer (stored deflate blocks, CRC-32,
 * Adler-32) and bip_resize_bilinear. Exists so that unchanged consumers of the reference (src/cli/bcnn_cl.c,
 * examples/inference_benchmark) link and run; see include/bip/bip.h. The decoders live in bip_decode.c. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bip/bip.h"

static uint32_t crc_table[256];
static void crc_init(void) {
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
}
static uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c;
}
static void put32(uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static int chunk(FILE *fp, const char *tag, const uint8_t *data, uint32_t len) {
    uint8_t hdr[8], crc[4];
    put32(hdr, len);
    memcpy(hdr + 4, tag, 4);
    uint32_t c = crc_update(0xffffffffu, hdr + 4, 4);
    c = crc_update(c, data, len) ^ 0xffffffffu;
    put32(crc, c);
    return fwrite(hdr, 1, 8, fp) == 8 && fwrite(data, 1, len, fp) == len && fwrite(crc, 1, 4, fp) == 4;
}

bip_status bip_write_image(char *filename, uint8_t *src, int32_t w, int32_t h, int32_t depth, int32_t stride) {
    if (!filename || !src) return BIP_INVALID_PTR;
    if (w <= 0 || h <= 0 || stride < w * depth) return BIP_INVALID_SIZE;
    if (depth != 1 && depth != 3 && depth != 4) return BIP_INVALID_PARAMETER;
    if (!crc_table[1]) crc_init();
    const size_t row = (size_t)w * depth + 1, raw = row * h;     /* filter byte 0 + pixels per scanline */
    const size_t nblocks = (raw + 65534) / 65535;
    const size_t zlen = 2 + raw + 5 * nblocks + 4;
    uint8_t *z = (uint8_t *)malloc(zlen), *scan = (uint8_t *)malloc(raw);
    if (!z || !scan) { free(z); free(scan); return BIP_UNKNOWN_ERROR; }
    for (int32_t y = 0; y < h; ++y) {
        scan[y * row] = 0;
        memcpy(scan + y * row + 1, src + (size_t)y * stride, (size_t)w * depth);
    }
    size_t o = 0, pos = 0;
    z[o++] = 0x78; z[o++] = 0x01;
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw; ++i) { a = (a + scan[i]) % 65521u; b = (b + a) % 65521u; }
    while (pos < raw) {
        const size_t n = raw - pos < 65535 ? raw - pos : 65535;
        z[o++] = (pos + n == raw) ? 1 : 0;
        z[o++] = n & 0xff; z[o++] = n >> 8; z[o++] = ~n & 0xff; z[o++] = (~n >> 8) & 0xff;
        memcpy(z + o, scan + pos, n);
        o += n; pos += n;
    }
    put32(z + o, (b << 16) | a);
    o += 4;
    FILE *fp = fopen(filename, "wb");
    if (!fp) { free(z); free(scan); return BIP_INVALID_PARAMETER; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)w); put32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8; ihdr[9] = depth == 1 ? 0 : (depth == 3 ? 2 : 6); ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    int ok = fwrite(sig, 1, 8, fp) == 8 && chunk(fp, "IHDR", ihdr, 13) && chunk(fp, "IDAT", z, (uint32_t)o) &&
             chunk(fp, "IEND", (const uint8_t *)"", 0);
    fclose(fp);
    free(z); free(scan);
    return ok ? BIP_SUCCESS : BIP_UNKNOWN_ERROR;
}

/* Source position of destination sample i: half-pixel centres, then clamped so that (index, index + 1) stays
 * inside the image; the fraction is quantised to 1/16 (reference bip.c:1118-1156). */
static void resize_tap(size_t i, float scale, size_t src_extent, int32_t *index, int32_t *frac) {
    float alpha = (float)((i + 0.5) * scale - 0.5);
    long idx = (long)floor(alpha);
    alpha -= idx;
    if (idx < 0) { idx = 0; alpha = 0; }
    if (idx > (long)src_extent - 2) { idx = (long)src_extent - 2; alpha = 1; }
    if (idx < 0) { idx = 0; alpha = 0; } /* one-sample axis: replicate (the reference reads out of bounds here) */
    *index = (int32_t)idx;
    *frac = (int32_t)(alpha * 16 + 0.5);
}

bip_status bip_resize_bilinear(uint8_t *src, size_t src_width, size_t src_height, size_t src_stride, uint8_t *dst,
                               size_t dst_width, size_t dst_height, size_t dst_stride, size_t depth) {
    if (!src || !dst) return BIP_INVALID_PTR;
    if (!src_width || !src_height || !dst_width || !dst_height) return BIP_INVALID_SIZE;
    if (depth < 1 || depth > 4) {
        fprintf(stderr, "resize_bilinear: 'depth' value must be >= 1 and <= 4\n");
        return BIP_INVALID_PARAMETER;
    }
    int32_t *ix = (int32_t *)malloc(2 * dst_width * sizeof(int32_t)), *ax = ix ? ix + dst_width : NULL;
    if (!ix) return BIP_UNKNOWN_ERROR;
    const float x_scale = (float)src_width / dst_width, y_scale = (float)src_height / dst_height;
    for (size_t x = 0; x < dst_width; ++x) resize_tap(x, x_scale, src_width, &ix[x], &ax[x]);
    const size_t xstep = src_width > 1 ? depth : 0, ystep = src_height > 1 ? src_stride : 0;
    for (size_t y = 0; y < dst_height; ++y) {
        int32_t iy, ay;
        resize_tap(y, y_scale, src_height, &iy, &ay);
        const uint8_t *r0 = src + (size_t)iy * src_stride, *r1 = r0 + ystep;
        uint8_t *out = dst + y * dst_stride;
        for (size_t x = 0; x < dst_width; ++x)
            for (size_t c = 0; c < depth; ++c) {
                const size_t o = (size_t)ix[x] * depth + c;
                /* horizontal pass in 1/16 units on both rows, vertical pass in 1/256 units, round to nearest */
                const int32_t h0 = (r0[o] << 4) + (r0[o + xstep] - r0[o]) * ax[x];
                const int32_t h1 = (r1[o] << 4) + (r1[o + xstep] - r1[o]) * ax[x];
                out[x * depth + c] = (uint8_t)(((h0 << 4) + (h1 - h0) * ay + 128) >> 8);
            }
    }
    free(ix);
    return BIP_SUCCESS;
}
