/* bip_decode.c -- libbip.so: bip_load_image / bip_load_image_from_memory without third-party code.
 *
 * The reference forwards to stb_image (src/bip/src/bip.c:1837-1870: stbi_load(filename, &w, &h, &c, 0), i.e. the
 * file's own channel count, 8 bits per channel, rows top-down, interleaved). This build decodes the formats the
 * hot-path consumers meet -- what examples/inference_benchmark feeds a net and what bip_write_image (ours and the
 * reference's stb writer) emits:
 *   PNG  8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced, any deflate block type (RFC 1950/1951/2083)
 *   PNM  P5 / P6 (binary), P2 / P3 (ASCII), maxval <= 255
 *   BMP  uncompressed 24 / 32 bit (bottom-up or top-down)
 *   JPEG baseline / extended / progressive Huffman, 8-bit, 1 or 3 components (bip_jpeg.c: the reference's pixels exactly)
 * Anything else (interlaced / 16-bit PNG, palette BMP, GIF, ...) fails with BIP_UNKNOWN_ERROR and a message on stderr, the
 * reference's error convention for an undecodable file. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bip/bip.h"

/* ---------------------------------------------------------------------------------------------- inflate */
typedef struct {
    const uint8_t *in;
    size_t in_len, in_pos;
    uint32_t bitbuf;
    int bitcnt;
    uint8_t *out;
    size_t out_len, out_pos;
    int err;
} inflate_state;

static uint32_t take_bits(inflate_state *s, int n) {
    while (s->bitcnt < n) {
        if (s->in_pos >= s->in_len) { s->err = 1; return 0; }
        s->bitbuf |= (uint32_t)s->in[s->in_pos++] << s->bitcnt;
        s->bitcnt += 8;
    }
    const uint32_t v = s->bitbuf & ((n == 32) ? 0xffffffffu : ((1u << n) - 1u));
    s->bitbuf >>= n;
    s->bitcnt -= n;
    return v;
}

/* canonical Huffman code described by the number of codes of each length and the symbols in code order */
typedef struct {
    uint16_t count[16];
    uint16_t symbol[288];
} huff_table;

static int huff_build(huff_table *h, const uint8_t *lengths, int n) {
    uint16_t offs[16];
    memset(h->count, 0, sizeof(h->count));
    for (int i = 0; i < n; ++i) h->count[lengths[i]]++;
    if (h->count[0] == n) return 0; /* no codes: legal for an unused distance tree */
    int left = 1;
    for (int len = 1; len < 16; ++len) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0) return -1; /* over-subscribed */
    }
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + h->count[len]);
    for (int i = 0; i < n; ++i)
        if (lengths[i]) h->symbol[offs[lengths[i]]++] = (uint16_t)i;
    return left; /* > 0: incomplete code (only acceptable for single-code trees) */
}

static int huff_decode(inflate_state *s, const huff_table *h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
        code |= (int)take_bits(s, 1);
        if (s->err) return -1;
        const int cnt = h->count[len];
        if (code - cnt < first) return h->symbol[index + (code - first)];
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    s->err = 1;
    return -1;
}

static const uint16_t k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
                                        67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                         1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11,
                                         12, 12, 13, 13};

static void inflate_codes(inflate_state *s, const huff_table *lit, const huff_table *dist) {
    for (;;) {
        int sym = huff_decode(s, lit);
        if (s->err) return;
        if (sym < 256) {
            if (s->out_pos >= s->out_len) { s->err = 1; return; }
            s->out[s->out_pos++] = (uint8_t)sym;
        } else if (sym == 256) {
            return;
        } else {
            sym -= 257;
            if (sym >= 29) { s->err = 1; return; }
            const int len = k_len_base[sym] + (int)take_bits(s, k_len_extra[sym]);
            const int ds = huff_decode(s, dist);
            if (s->err || ds < 0 || ds >= 30) { s->err = 1; return; }
            const size_t d = (size_t)k_dist_base[ds] + take_bits(s, k_dist_extra[ds]);
            if (s->err || d > s->out_pos || s->out_pos + (size_t)len > s->out_len) { s->err = 1; return; }
            for (int i = 0; i < len; ++i, ++s->out_pos) s->out[s->out_pos] = s->out[s->out_pos - d];
        }
    }
}

/* zlib stream (2-byte header, deflate blocks, Adler-32 trailer) -> exactly out_len bytes; 0 on success */
static int zlib_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    if (in_len < 6 || (in[0] & 0x0f) != 8 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return -1;
    inflate_state s = {0};
    s.in = in; s.in_len = in_len; s.in_pos = 2; s.out = out; s.out_len = out_len;
    int last;
    do {
        last = (int)take_bits(&s, 1);
        const int type = (int)take_bits(&s, 2);
        if (s.err) return -1;
        if (type == 0) { /* stored */
            s.bitbuf = 0; s.bitcnt = 0;
            if (s.in_pos + 4 > s.in_len) return -1;
            const unsigned len = s.in[s.in_pos] | (s.in[s.in_pos + 1] << 8);
            const unsigned nlen = s.in[s.in_pos + 2] | (s.in[s.in_pos + 3] << 8);
            s.in_pos += 4;
            if ((len ^ 0xffffu) != nlen || s.in_pos + len > s.in_len || s.out_pos + len > s.out_len) return -1;
            memcpy(s.out + s.out_pos, s.in + s.in_pos, len);
            s.in_pos += len; s.out_pos += len;
        } else if (type == 1) { /* fixed codes */
            uint8_t lengths[320];
            huff_table lit, dist;
            int i = 0;
            for (; i < 144; ++i) lengths[i] = 8;
            for (; i < 256; ++i) lengths[i] = 9;
            for (; i < 280; ++i) lengths[i] = 7;
            for (; i < 288; ++i) lengths[i] = 8;
            huff_build(&lit, lengths, 288);
            for (i = 0; i < 30; ++i) lengths[i] = 5;
            huff_build(&dist, lengths, 30);
            inflate_codes(&s, &lit, &dist);
        } else if (type == 2) { /* dynamic codes */
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t lengths[320];
            huff_table lencode, lit, dist;
            const int nlen = (int)take_bits(&s, 5) + 257, ndist = (int)take_bits(&s, 5) + 1;
            const int ncode = (int)take_bits(&s, 4) + 4;
            if (s.err || nlen > 286 || ndist > 30) return -1;
            memset(lengths, 0, sizeof(lengths));
            for (int i = 0; i < ncode; ++i) lengths[order[i]] = (uint8_t)take_bits(&s, 3);
            if (huff_build(&lencode, lengths, 19) != 0) return -1;
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = huff_decode(&s, &lencode);
                if (s.err) return -1;
                if (sym < 16) {
                    lengths[idx++] = (uint8_t)sym;
                } else {
                    int prev = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) return -1;
                        prev = lengths[idx - 1];
                        rep = 3 + (int)take_bits(&s, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)take_bits(&s, 3);
                    } else {
                        rep = 11 + (int)take_bits(&s, 7);
                    }
                    if (s.err || idx + rep > nlen + ndist) return -1;
                    while (rep--) lengths[idx++] = (uint8_t)prev;
                }
            }
            if (lengths[256] == 0) return -1;
            int r = huff_build(&lit, lengths, nlen);
            if (r < 0 || (r > 0 && nlen - lit.count[0] != 1)) return -1;
            r = huff_build(&dist, lengths + nlen, ndist);
            if (r < 0 || (r > 0 && ndist - dist.count[0] != 1)) return -1;
            inflate_codes(&s, &lit, &dist);
        } else {
            return -1;
        }
        if (s.err) return -1;
    } while (!last);
    if (s.out_pos != out_len) return -1;
    /* Adler-32 of the decoded bytes (big-endian trailer), when present */
    s.bitbuf = 0; s.bitcnt = 0;
    if (s.in_pos + 4 <= s.in_len) {
        uint32_t a = 1, b = 0;
        for (size_t i = 0; i < out_len; ++i) { a = (a + out[i]) % 65521u; b = (b + a) % 65521u; }
        const uint8_t *t = s.in + s.in_pos;
        const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
        if (want != ((b << 16) | a)) return -1;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------- PNG */
static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static bip_status decode_png(const uint8_t *buf, size_t len, uint8_t **out, int32_t *w, int32_t *h, int32_t *depth) {
    size_t pos = 8, zcap = 0, zlen = 0;
    uint8_t *z = NULL, palette[256 * 3];
    uint32_t width = 0, height = 0;
    int bits = 0, ctype = -1, interlace = 0, have_plte = 0, done = 0;
    memset(palette, 0, sizeof(palette));
    while (!done && pos + 12 <= len) {
        const uint32_t clen = be32(buf + pos);
        const uint8_t *tag = buf + pos + 4, *data = buf + pos + 8;
        if (clen > len || pos + 12 + clen > len) break;
        if (!memcmp(tag, "IHDR", 4) && clen >= 13) {
            width = be32(data); height = be32(data + 4);
            bits = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(tag, "PLTE", 4) && clen <= 768) {
            memcpy(palette, data, clen);
            have_plte = 1;
        } else if (!memcmp(tag, "IDAT", 4)) {
            if (zlen + clen > zcap) {
                zcap = (zlen + clen) * 2;
                uint8_t *nz = (uint8_t *)realloc(z, zcap);
                if (!nz) { free(z); return BIP_UNKNOWN_ERROR; }
                z = nz;
            }
            memcpy(z + zlen, data, clen);
            zlen += clen;
        } else if (!memcmp(tag, "IEND", 4)) {
            done = 1;
        }
        pos += 12 + (size_t)clen;
    }
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: ch = 0;
    }
    if (!z || !ch || bits != 8 || interlace || !width || !height || width > (1u << 24) || height > (1u << 24) ||
        (ctype == 3 && !have_plte)) {
        fprintf(stderr, "[ERROR] bip_load_image: unsupported PNG variant (bit depth %d, colour type %d, interlace %d)\n",
                bits, ctype, interlace);
        free(z);
        return BIP_UNKNOWN_ERROR;
    }
    const size_t row = (size_t)width * ch, raw_len = (row + 1) * height;
    uint8_t *raw = (uint8_t *)malloc(raw_len);
    if (!raw || zlib_inflate(z, zlen, raw, raw_len) != 0) {
        fprintf(stderr, "[ERROR] bip_load_image: corrupt PNG data stream\n");
        free(z); free(raw);
        return BIP_UNKNOWN_ERROR;
    }
    free(z);
    const int out_ch = (ctype == 3) ? 3 : ch;
    uint8_t *img = (uint8_t *)malloc((size_t)width * height * out_ch);
    uint8_t *lines = (uint8_t *)calloc(2, row); /* previous / current reconstructed scanline */
    if (!img || !lines) { free(raw); free(img); free(lines); return BIP_UNKNOWN_ERROR; }
    uint8_t *prev = lines, *cur = lines + row;
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t *src = raw + (size_t)y * (row + 1);
        const int filter = src[0];
        ++src;
        for (size_t x = 0; x < row; ++x) {
            const int a = x >= (size_t)ch ? cur[x - ch] : 0, b = prev[x], c = x >= (size_t)ch ? prev[x - ch] : 0;
            int v = src[x];
            switch (filter) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            cur[x] = (uint8_t)v;
        }
        uint8_t *dst = img + (size_t)y * width * out_ch;
        if (ctype == 3) {
            for (uint32_t x = 0; x < width; ++x) memcpy(dst + 3 * x, palette + 3 * cur[x], 3);
        } else {
            memcpy(dst, cur, row);
        }
        uint8_t *t = prev; prev = cur; cur = t;
    }
    free(raw); free(lines);
    *out = img; *w = (int32_t)width; *h = (int32_t)height; *depth = out_ch;
    return BIP_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------- PNM */
static int pnm_token(const uint8_t *buf, size_t len, size_t *pos, long *value) {
    size_t p = *pos;
    for (;;) {
        while (p < len && (buf[p] == ' ' || buf[p] == '\t' || buf[p] == '\r' || buf[p] == '\n')) ++p;
        if (p < len && buf[p] == '#') { while (p < len && buf[p] != '\n') ++p; continue; }
        break;
    }
    if (p >= len || buf[p] < '0' || buf[p] > '9') return -1;
    long v = 0;
    while (p < len && buf[p] >= '0' && buf[p] <= '9') { v = v * 10 + (buf[p] - '0'); if (v > (1L << 30)) return -1; ++p; }
    *pos = p; *value = v;
    return 0;
}

static bip_status decode_pnm(const uint8_t *buf, size_t len, uint8_t **out, int32_t *w, int32_t *h, int32_t *depth) {
    const int kind = buf[1] - '0';
    const int ch = (kind == 3 || kind == 6) ? 3 : 1, ascii = kind < 4;
    size_t pos = 2;
    long width, height, maxval;
    if (pnm_token(buf, len, &pos, &width) || pnm_token(buf, len, &pos, &height) || pnm_token(buf, len, &pos, &maxval) ||
        width <= 0 || height <= 0 || maxval <= 0 || maxval > 255) {
        fprintf(stderr, "[ERROR] bip_load_image: unsupported PNM header\n");
        return BIP_UNKNOWN_ERROR;
    }
    const size_t count = (size_t)width * height * ch;
    uint8_t *img = (uint8_t *)malloc(count);
    if (!img) return BIP_UNKNOWN_ERROR;
    if (ascii) {
        for (size_t i = 0; i < count; ++i) {
            long v;
            if (pnm_token(buf, len, &pos, &v) || v > maxval) { free(img); return BIP_UNKNOWN_ERROR; }
            img[i] = (uint8_t)v;
        }
    } else {
        ++pos; /* exactly one whitespace byte after maxval */
        if (pos + count > len) { free(img); return BIP_UNKNOWN_ERROR; }
        memcpy(img, buf + pos, count);
    }
    *out = img; *w = (int32_t)width; *h = (int32_t)height; *depth = ch;
    return BIP_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------- BMP */
static uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

static bip_status decode_bmp(const uint8_t *buf, size_t len, uint8_t **out, int32_t *w, int32_t *h, int32_t *depth) {
    if (len < 54) return BIP_UNKNOWN_ERROR;
    const uint32_t offset = le32(buf + 10), hdr = le32(buf + 14);
    const int32_t width = (int32_t)le32(buf + 18), sheight = (int32_t)le32(buf + 22);
    const int bpp = buf[28] | (buf[29] << 8);
    const uint32_t compression = le32(buf + 30);
    const int32_t height = sheight < 0 ? -sheight : sheight;
    if (hdr < 40 || width <= 0 || height <= 0 || (bpp != 24 && bpp != 32) || (compression != 0 && compression != 3)) {
        fprintf(stderr, "[ERROR] bip_load_image: unsupported BMP variant (%d bpp, compression %u)\n", bpp, compression);
        return BIP_UNKNOWN_ERROR;
    }
    const int sch = bpp / 8;
    const size_t stride = ((size_t)width * sch + 3) & ~(size_t)3;
    if ((size_t)offset + stride * height > len) return BIP_UNKNOWN_ERROR;
    uint8_t *img = (uint8_t *)malloc((size_t)width * height * 3);
    if (!img) return BIP_UNKNOWN_ERROR;
    for (int32_t y = 0; y < height; ++y) {
        const uint8_t *src = buf + offset + stride * (size_t)(sheight < 0 ? y : height - 1 - y);
        uint8_t *dst = img + (size_t)y * width * 3;
        for (int32_t x = 0; x < width; ++x) { /* stored B, G, R */
            dst[3 * x + 0] = src[sch * x + 2];
            dst[3 * x + 1] = src[sch * x + 1];
            dst[3 * x + 2] = src[sch * x + 0];
        }
    }
    *out = img; *w = width; *h = height; *depth = 3;
    return BIP_SUCCESS;
}

uint8_t *bip_decode_jpeg(const uint8_t *buf, size_t len, int32_t *w, int32_t *h, int32_t *depth); /* bip_jpeg.c */

/* ---------------------------------------------------------------------------------------------- entry points */
bip_status bip_load_image_from_memory(unsigned char *buffer, int buffer_size, uint8_t **src, int32_t *src_width,
                                      int32_t *src_height, int32_t *src_depth) {
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!buffer || !src || !src_width || !src_height || !src_depth) return BIP_INVALID_PTR;
    const size_t len = buffer_size > 0 ? (size_t)buffer_size : 0;
    if (len >= 8 && !memcmp(buffer, png_sig, 8)) return decode_png(buffer, len, src, src_width, src_height, src_depth);
    if (len >= 3 && buffer[0] == 'P' && ((buffer[1] >= '2' && buffer[1] <= '3') || (buffer[1] >= '5' && buffer[1] <= '6')))
        return decode_pnm(buffer, len, src, src_width, src_height, src_depth);
    if (len >= 2 && buffer[0] == 'B' && buffer[1] == 'M') return decode_bmp(buffer, len, src, src_width, src_height, src_depth);
    if (len >= 4 && buffer[0] == 0xff && buffer[1] == 0xd8) {
        uint8_t *img = bip_decode_jpeg(buffer, len, src_width, src_height, src_depth);
        if (img) { *src = img; return BIP_SUCCESS; }
        fprintf(stderr, "[ERROR] bip_load_image: corrupt or unsupported JPEG stream (8-bit Huffman, 1 or 3 components are supported)\n");
        return BIP_UNKNOWN_ERROR;
    }
    fprintf(stderr, "[ERROR] Cannot load image from buffer: format not supported by this build (PNG, JPEG, PNM and BMP are)\n");
    return BIP_UNKNOWN_ERROR;
}

bip_status bip_load_image(char *filename, uint8_t **src, int32_t *src_width, int32_t *src_height, int32_t *src_depth) {
    if (!filename || !src) return BIP_INVALID_PTR;
    FILE *fp = fopen(filename, "rb");
    if (!fp) {
        fprintf(stderr, "[ERROR] Cannot load file image %s\n", filename);
        return BIP_UNKNOWN_ERROR;
    }
    fseek(fp, 0, SEEK_END);
    const long size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    uint8_t *buf = (size > 0 && size < (1L << 30)) ? (uint8_t *)malloc((size_t)size) : NULL;
    const int ok = buf && fread(buf, 1, (size_t)size, fp) == (size_t)size;
    fclose(fp);
    bip_status st = BIP_UNKNOWN_ERROR;
    if (ok) st = bip_load_image_from_memory(buf, (int)size, src, src_width, src_height, src_depth);
    if (st != BIP_SUCCESS) fprintf(stderr, "[ERROR] Cannot load file image %s\n", filename);
    free(buf);
    return st;
}
