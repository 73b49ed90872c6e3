/* bip_jpeg.c -- libbip.so: JPEG decoding for bip_load_image / bip_load_image_from_memory.
 *
 * The reference decodes images through stb_image 2.08 (src/bip/thirdparty/stb_image/stb_image.h, called from
 * src/bip/src/bip.c:1837-1870 with req_comp = 0), and its list-file dataset readers and examples/inference_benchmark are
 * normally fed .jpg files. A JPEG decoder is only a drop-in here if it returns the SAME PIXELS, because the decoded bytes
 * go straight into the augmenter and the input tensor: the lossy part of JPEG -- inverse DCT, chroma upsampling,
 * YCbCr -> RGB -- is implementation-defined, so this file follows that library's published arithmetic step by step:
 *   - 8-bit Huffman JPEG, baseline / extended sequential (SOF0, SOF1) and progressive (SOF2), 1 or 3 components,
 *     sampling factors 1..4, restart intervals, 8-bit quantisation tables; everything else is refused;
 *   - coefficients are dequantised in 16-bit arithmetic; the inverse DCT is the Loeffler-Ligtenberg-Moschytz integer
 *     transform with 12-bit constants, 2 extra bits kept between the column and the row pass, rounding constants
 *     512 / 65536 + (128 << 17), shifts 10 / 17;
 *   - chroma planes are brought to full resolution row by row with the "triangle" filters (3 near + 1 far) / 4
 *     (one axis) and (9, 3, 3, 1) / 16 (both axes), nearest neighbour for the other ratios;
 *   - Y, Cb, Cr -> R, G, B in 20-bit fixed point with the constants 1.402, 0.71414, 0.34414, 1.772 rounded to 12 bits
 *     and the Cb term of green truncated to its upper 16 bits; a one-component file yields one channel.
 * tests/test_bip.py compares every supported variant (subsampling 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 / 4:1:1, grey,
 * progressive, restart markers, odd sizes, qualities 5..100) byte for byte with the reference's loader. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bip/bip.h"

/* zig-zag position -> row-major index of the 8 x 8 block */
static const uint8_t k_unzig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    /* canonical Huffman code: symbols in code order, first code / first symbol index per length, and a 9-bit prefix table */
    uint8_t symbols[256];
    int32_t first_code[18]; /* first code of length L, left-justified to 16 bits; [17] = sentinel */
    int32_t first_index[17];
    int32_t end_code[18]; /* one past the last code of length L, left-justified */
    uint16_t quick[512];  /* (length << 8) | symbol for codes of <= 9 bits, 0 = longer */
    int defined;
} jhuff;

typedef struct {
    int id, h, v, tq, dc_table, ac_table;
    int dc_pred;
    int width, height;     /* samples that carry image content */
    int pitch, rows;       /* allocated plane: whole MCUs */
    int blocks_w, blocks_h;
    uint8_t *plane;
    int16_t *coeff; /* progressive: all coefficient blocks of the component */
} jcomp;

typedef struct {
    const uint8_t *p, *end;
    uint32_t bits;  /* left-justified bit reservoir */
    int nbits;
    int marker;     /* marker met while refilling (0 = none); after it the reservoir is fed zeros */
    uint8_t qt[4][64]; /* row-major */
    jhuff dc[4], ac[4];
    jcomp comp[3];
    int ncomp, width, height, hmax, vmax, mcus_x, mcus_y;
    int progressive, restart_interval;
    /* current scan */
    int scan_n, scan_comp[3], ss, se, ah, al, eob_run, todo;
} jdec;

/* ---- byte / bit input ------------------------------------------------------------------------------------------------ */
static int get8(jdec *d) { return d->p < d->end ? *d->p++ : 0; }
static int get16(jdec *d) { const int a = get8(d); return (a << 8) | get8(d); }

static void refill(jdec *d) {
    while (d->nbits <= 24) {
        int b = 0;
        if (!d->marker) {
            b = get8(d);
            if (b == 0xff) {
                const int c = get8(d);
                if (c != 0) { /* a marker ends the entropy-coded segment: zeros from here on */
                    d->marker = c;
                    b = 0;
                }
            }
        }
        d->bits |= (uint32_t)b << (24 - d->nbits);
        d->nbits += 8;
    }
}

static int take_bits(jdec *d, int n) { /* n in 1..16 */
    if (d->nbits < n) refill(d);
    const uint32_t v = d->bits >> (32 - n);
    d->bits <<= n;
    d->nbits -= n;
    return (int)v;
}
static int take_bit(jdec *d) { return take_bits(d, 1); }

/* n magnitude bits -> signed value (the "extend" procedure of the standard, F.2.2.1) */
static int take_signed(jdec *d, int n) {
    const int v = take_bits(d, n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
}

static int huff_build(jhuff *h, const int counts[16], const uint8_t *symbols, int nsym) {
    int code = 0, k = 0;
    memset(h->quick, 0, sizeof(h->quick));
    memcpy(h->symbols, symbols, (size_t)nsym);
    for (int len = 1; len <= 16; ++len) {
        h->first_index[len] = k;
        h->first_code[len] = code << (16 - len);
        for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
            if (code >= (1 << len) || k >= nsym) return 0; /* more codes than the length can hold / than symbols given */
            if (len <= 9)
                for (int fill = 0; fill < (1 << (9 - len)); ++fill)
                    h->quick[(code << (9 - len)) + fill] = (uint16_t)((len << 8) | symbols[k]);
        }
        h->end_code[len] = code << (16 - len);
        code <<= 1;
    }
    h->end_code[17] = 0x7fffffff;
    h->defined = 1;
    return 1;
}

static int huff_symbol(jdec *d, const jhuff *h) {
    if (d->nbits < 16) refill(d);
    const uint16_t q = h->quick[d->bits >> 23];
    if (q) {
        d->bits <<= q >> 8;
        d->nbits -= q >> 8;
        return q & 0xff;
    }
    const int32_t top = (int32_t)(d->bits >> 16);
    for (int len = 10; len <= 16; ++len)
        if (top < h->end_code[len]) {
            const int idx = h->first_index[len] + ((top - h->first_code[len]) >> (16 - len));
            d->bits <<= len;
            d->nbits -= len;
            return h->symbols[idx & 255];
        }
    return -1;
}

/* ---- inverse DCT ------------------------------------------------------------------------------------------------------ */
#define FIX(x) ((int)((x) * 4096 + 0.5))
static uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* one 8-point pass: even part in e[0..3], odd part in o[0..3] (both scaled by 4096); the caller combines e[i] +- o[3-i].
 * All arithmetic is modulo 2^32 (unsigned), which is what the reference's int arithmetic amounts to on every target it
 * runs on and keeps a corrupt stream's oversized coefficients from being undefined behaviour here. */
typedef uint32_t u32;
#define UMUL(a, k) ((u32)(a) * (u32)(int32_t)(k))
static void idct8(int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t s4, int32_t s5, int32_t s6, int32_t s7, u32 e[4],
                  u32 o[4]) {
    const u32 z = UMUL((u32)s2 + (u32)s6, FIX(0.5411961f));
    const u32 a = z + UMUL(s6, FIX(-1.847759065f)), b = z + UMUL(s2, FIX(0.765366865f));
    const u32 c = ((u32)s0 + (u32)s4) * 4096u, dd = ((u32)s0 - (u32)s4) * 4096u;
    e[0] = c + b; e[3] = c - b; e[1] = dd + a; e[2] = dd - a;
    const u32 p3 = (u32)s7 + (u32)s3, p4 = (u32)s5 + (u32)s1, p1 = (u32)s7 + (u32)s1, p2 = (u32)s5 + (u32)s3;
    const u32 p5 = UMUL(p3 + p4, FIX(1.175875602f));
    const u32 q1 = p5 + UMUL(p1, FIX(-0.899976223f)), q2 = p5 + UMUL(p2, FIX(-2.562915447f));
    const u32 q3 = UMUL(p3, FIX(-1.961570560f)), q4 = UMUL(p4, FIX(-0.390180644f));
    o[3] = UMUL(s1, FIX(1.501321110f)) + q1 + q4;
    o[2] = UMUL(s3, FIX(3.072711026f)) + q2 + q3;
    o[1] = UMUL(s5, FIX(2.053119869f)) + q2 + q4;
    o[0] = UMUL(s7, FIX(0.298631336f)) + q1 + q3;
}
static int32_t sar(u32 v, int n) { /* arithmetic shift right of the two's-complement value */
    return (int32_t)(v >> n) | ((v & 0x80000000u) ? (int32_t)(~0u << (32 - n)) : 0);
}

static void idct_block(uint8_t *out, int pitch, const int16_t c[64]) {
    int32_t mid[64];
    u32 e[4], o[4];
    for (int x = 0; x < 8; ++x) { /* columns; 2 extra bits of precision are kept */
        if (!(c[x + 8] | c[x + 16] | c[x + 24] | c[x + 32] | c[x + 40] | c[x + 48] | c[x + 56])) {
            const int32_t dc = c[x] * 4;
            for (int y = 0; y < 8; ++y) mid[8 * y + x] = dc;
            continue;
        }
        idct8(c[x], c[x + 8], c[x + 16], c[x + 24], c[x + 32], c[x + 40], c[x + 48], c[x + 56], e, o);
        for (int i = 0; i < 4; ++i) {
            mid[8 * i + x] = sar(e[i] + 512u + o[3 - i], 10);
            mid[8 * (7 - i) + x] = sar(e[i] + 512u - o[3 - i], 10);
        }
    }
    for (int y = 0; y < 8; ++y) { /* rows: remove 12 + 2 + 3 bits, re-centre on 128 */
        const int32_t *m = mid + 8 * y;
        idct8(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], e, o);
        uint8_t *row = out + (size_t)y * pitch;
        for (int i = 0; i < 4; ++i) {
            const u32 base = e[i] + 65536u + (128u << 17);
            row[i] = clamp255(sar(base + o[3 - i], 17));
            row[7 - i] = clamp255(sar(base - o[3 - i], 17));
        }
    }
}

/* ---- coefficient decoding --------------------------------------------------------------------------------------------- */
static int block_sequential(jdec *d, jcomp *c, int16_t blk[64]) {
    const jhuff *hd = &d->dc[c->dc_table], *ha = &d->ac[c->ac_table];
    const uint8_t *q = d->qt[c->tq];
    memset(blk, 0, 64 * sizeof(int16_t));
    const int t = huff_symbol(d, hd);
    if (t < 0 || t > 15) return 0;
    c->dc_pred += t ? take_signed(d, t) : 0;
    blk[0] = (int16_t)(c->dc_pred * q[0]);
    for (int k = 1; k < 64;) {
        const int rs = huff_symbol(d, ha);
        if (rs < 0) return 0;
        const int run = rs >> 4, size = rs & 15;
        if (size == 0) {
            if (rs != 0xf0) break; /* end of block */
            k += 16;
            continue;
        }
        k += run;
        if (k > 63) return 0;
        const int at = k_unzig[k++];
        blk[at] = (int16_t)(take_signed(d, size) * q[at]);
    }
    return 1;
}

static int block_prog_dc(jdec *d, jcomp *c, int16_t blk[64]) {
    if (d->se != 0) return 0;
    if (d->ah == 0) { /* first pass of the DC coefficient */
        memset(blk, 0, 64 * sizeof(int16_t));
        const int t = huff_symbol(d, &d->dc[c->dc_table]);
        if (t < 0 || t > 15) return 0;
        c->dc_pred += t ? take_signed(d, t) : 0;
        blk[0] = (int16_t)(c->dc_pred * (1 << d->al));
    } else if (take_bit(d)) { /* refinement: one more bit */
        blk[0] += (int16_t)(1 << d->al);
    }
    return 1;
}

/* correction bit of an already non-zero coefficient during a refinement pass */
static void refine_nonzero(jdec *d, int16_t *p, int16_t bit) {
    if (take_bit(d) && (*p & bit) == 0) *p = (int16_t)(*p > 0 ? *p + bit : *p - bit);
}

static int block_prog_ac(jdec *d, jcomp *c, int16_t blk[64]) {
    const jhuff *ha = &d->ac[c->ac_table];
    if (d->ss == 0) return 0;
    if (d->ah == 0) { /* first pass over the band ss..se */
        if (d->eob_run) { --d->eob_run; return 1; }
        for (int k = d->ss; k <= d->se;) {
            const int rs = huff_symbol(d, ha);
            if (rs < 0) return 0;
            const int run = rs >> 4, size = rs & 15;
            if (size == 0) {
                if (run < 15) { /* end of band for 2^run (+ extra bits) blocks, this one included */
                    d->eob_run = (1 << run) + (run ? take_bits(d, run) : 0) - 1;
                    break;
                }
                k += 16;
                continue;
            }
            k += run;
            if (k > 63) return 0;
            blk[k_unzig[k++]] = (int16_t)(take_signed(d, size) * (1 << d->al));
        }
        return 1;
    }
    /* refinement pass: every non-zero coefficient met gets a correction bit, new coefficients enter as +-(1 << al) */
    const int16_t bit = (int16_t)(1 << d->al);
    if (d->eob_run) {
        --d->eob_run;
        for (int k = d->ss; k <= d->se; ++k) {
            int16_t *p = &blk[k_unzig[k]];
            if (*p) refine_nonzero(d, p, bit);
        }
        return 1;
    }
    for (int k = d->ss; k <= d->se;) {
        const int rs = huff_symbol(d, ha);
        if (rs < 0) return 0;
        int run = rs >> 4, value = 0;
        const int size = rs & 15;
        if (size == 0) {
            if (run < 15) {
                d->eob_run = (1 << run) - 1 + (run ? take_bits(d, run) : 0);
                run = 64; /* the rest of the band only gets correction bits */
            }
        } else {
            if (size != 1) return 0;
            value = take_bit(d) ? bit : -bit;
        }
        while (k <= d->se) { /* skip `run` zero coefficients, correcting the non-zero ones on the way */
            int16_t *p = &blk[k_unzig[k++]];
            if (*p) {
                refine_nonzero(d, p, bit);
            } else {
                if (run == 0) { *p = (int16_t)value; break; }
                --run;
            }
        }
    }
    return 1;
}

/* ---- scans -------------------------------------------------------------------------------------------------------------- */
static void restart(jdec *d) {
    d->bits = 0; d->nbits = 0; d->marker = 0; d->eob_run = 0;
    for (int i = 0; i < 3; ++i) d->comp[i].dc_pred = 0;
    d->todo = d->restart_interval ? d->restart_interval : 0x7fffffff;
}

/* after each MCU: at the end of a restart interval the next marker must be RSTn; 0 = stop decoding this scan */
static int mcu_done(jdec *d) {
    if (--d->todo > 0) return 1;
    if (d->nbits < 24) refill(d);
    if (d->marker < 0xd0 || d->marker > 0xd7) return 0;
    restart(d);
    return 1;
}

static int one_block(jdec *d, jcomp *c, int bx, int by) {
    if (d->progressive) {
        int16_t *blk = c->coeff + 64 * ((size_t)by * c->blocks_w + bx);
        return d->ss == 0 ? block_prog_dc(d, c, blk) : block_prog_ac(d, c, blk);
    }
    int16_t blk[64];
    if (!block_sequential(d, c, blk)) return 0;
    idct_block(c->plane + (size_t)by * 8 * c->pitch + bx * 8, c->pitch, blk);
    return 1;
}

static int decode_scan(jdec *d) {
    restart(d);
    if (d->scan_n == 1) { /* one component: its blocks in raster order, only those that carry image content */
        jcomp *c = &d->comp[d->scan_comp[0]];
        const int bw = (c->width + 7) >> 3, bh = (c->height + 7) >> 3;
        for (int by = 0; by < bh; ++by)
            for (int bx = 0; bx < bw; ++bx) {
                if (!one_block(d, c, bx, by)) return 0;
                if (!mcu_done(d)) return 1;
            }
        return 1;
    }
    if (d->progressive && d->ss != 0) return 0; /* AC scans are never interleaved */
    for (int my = 0; my < d->mcus_y; ++my)
        for (int mx = 0; mx < d->mcus_x; ++mx) {
            for (int k = 0; k < d->scan_n; ++k) {
                jcomp *c = &d->comp[d->scan_comp[k]];
                for (int y = 0; y < c->v; ++y)
                    for (int x = 0; x < c->h; ++x)
                        if (!one_block(d, c, mx * c->h + x, my * c->v + y)) return 0;
            }
            if (!mcu_done(d)) return 1;
        }
    return 1;
}

/* ---- markers -------------------------------------------------------------------------------------------------------------- */
static int next_marker(jdec *d) { /* 0xff = none */
    if (d->marker) { const int m = d->marker; d->marker = 0; return m; }
    int x = get8(d);
    if (x != 0xff) return 0xff;
    while (x == 0xff && d->p < d->end) x = get8(d);
    return x;
}

static int read_tables(jdec *d, int m) {
    if (m == 0xdd) { /* DRI */
        if (get16(d) != 4) return 0;
        d->restart_interval = get16(d);
        return 1;
    }
    if (m == 0xdb) { /* DQT, 8-bit entries only */
        int left = get16(d) - 2;
        while (left > 0) {
            const int pt = get8(d);
            if ((pt >> 4) != 0 || (pt & 15) > 3) return 0;
            for (int i = 0; i < 64; ++i) d->qt[pt & 15][k_unzig[i]] = (uint8_t)get8(d);
            left -= 65;
        }
        return left == 0;
    }
    if (m == 0xc4) { /* DHT */
        int left = get16(d) - 2;
        while (left > 0) {
            const int tt = get8(d);
            int counts[16], n = 0;
            uint8_t symbols[256];
            if ((tt >> 4) > 1 || (tt & 15) > 3) return 0;
            for (int i = 0; i < 16; ++i) n += counts[i] = get8(d);
            if (n > 256) return 0;
            for (int i = 0; i < n; ++i) symbols[i] = (uint8_t)get8(d);
            if (!huff_build((tt >> 4) ? &d->ac[tt & 15] : &d->dc[tt & 15], counts, symbols, n)) return 0;
            left -= 17 + n;
        }
        return left == 0;
    }
    if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) { /* APPn, COM */
        const int len = get16(d) - 2;
        if (len < 0 || d->p + len > d->end) return 0;
        d->p += len;
        return 1;
    }
    return 0;
}

static int read_frame(jdec *d) {
    const int len = get16(d);
    if (len < 11 || get8(d) != 8) return 0; /* 8-bit samples only */
    d->height = get16(d);
    d->width = get16(d);
    d->ncomp = get8(d);
    if (!d->height || !d->width || (d->ncomp != 1 && d->ncomp != 3) || len != 8 + 3 * d->ncomp) return 0;
    if ((1 << 30) / d->width / d->ncomp < d->height) return 0;
    d->hmax = d->vmax = 1;
    for (int i = 0; i < d->ncomp; ++i) {
        jcomp *c = &d->comp[i];
        c->id = get8(d);
        if (c->id != i + 1 && c->id != i) return 0;
        const int hv = get8(d);
        c->h = hv >> 4; c->v = hv & 15; c->tq = get8(d);
        if (c->h < 1 || c->h > 4 || c->v < 1 || c->v > 4 || c->tq > 3) return 0;
        if (c->h > d->hmax) d->hmax = c->h;
        if (c->v > d->vmax) d->vmax = c->v;
    }
    /* every component must divide the largest sampling factor: the upsampler replicates by the integer ratio
     * hmax / h, and with a truncated ratio (h = 3 under hmax = 4) a plane row is shorter than the image row the colour
     * conversion reads from it (heap over-read). No encoder writes such frames. */
    for (int i = 0; i < d->ncomp; ++i)
        if (d->hmax % d->comp[i].h != 0 || d->vmax % d->comp[i].v != 0) return 0;
    d->mcus_x = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
    d->mcus_y = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
    for (int i = 0; i < d->ncomp; ++i) {
        jcomp *c = &d->comp[i];
        c->width = (d->width * c->h + d->hmax - 1) / d->hmax;
        c->height = (d->height * c->v + d->vmax - 1) / d->vmax;
        c->pitch = d->mcus_x * c->h * 8;
        c->rows = d->mcus_y * c->v * 8;
        c->blocks_w = c->pitch >> 3; c->blocks_h = c->rows >> 3;
        c->plane = (uint8_t *)calloc((size_t)c->pitch * c->rows + 16, 1);
        if (!c->plane) return 0;
        if (d->progressive) {
            c->coeff = (int16_t *)calloc((size_t)c->blocks_w * c->blocks_h * 64, sizeof(int16_t));
            if (!c->coeff) return 0;
        }
    }
    return 1;
}

static int read_scan_header(jdec *d) {
    const int len = get16(d);
    d->scan_n = get8(d);
    if (d->scan_n < 1 || d->scan_n > d->ncomp || len != 6 + 2 * d->scan_n) return 0;
    for (int i = 0; i < d->scan_n; ++i) {
        const int id = get8(d), tables = get8(d);
        int which = 0;
        while (which < d->ncomp && d->comp[which].id != id) ++which;
        if (which == d->ncomp || (tables >> 4) > 3 || (tables & 15) > 3) return 0;
        d->comp[which].dc_table = tables >> 4;
        d->comp[which].ac_table = tables & 15;
        d->scan_comp[i] = which;
    }
    d->ss = get8(d); d->se = get8(d);
    const int a = get8(d);
    d->ah = a >> 4; d->al = a & 15;
    if (d->progressive) {
        if (d->ss > 63 || d->se > 63 || d->ss > d->se || d->ah > 13 || d->al > 13) return 0;
    } else {
        if (d->ss != 0 || d->ah != 0 || d->al != 0) return 0;
        d->se = 63;
    }
    return 1;
}

static int decode_planes(jdec *d) {
    if (next_marker(d) != 0xd8) return 0; /* SOI */
    int m = next_marker(d);
    while (m != 0xc0 && m != 0xc1 && m != 0xc2) { /* tables and application segments up to the frame header */
        if (!read_tables(d, m)) return 0;
        m = next_marker(d);
        while (m == 0xff) {
            if (d->p >= d->end) return 0;
            m = next_marker(d);
        }
    }
    d->progressive = m == 0xc2;
    if (!read_frame(d)) return 0;
    for (m = next_marker(d); m != 0xd9; m = next_marker(d)) { /* until EOI */
        if (m == 0xda) {
            if (!read_scan_header(d) || !decode_scan(d)) return 0;
            if (!d->marker) { /* zero padding behind the entropy-coded data */
                while (d->p < d->end) {
                    const int x = get8(d);
                    if (x == 0xff) { d->marker = get8(d); break; }
                    if (x != 0) return 0;
                }
            }
        } else if (!read_tables(d, m)) {
            return 0;
        }
    }
    if (d->progressive) /* all scans seen: dequantise (16-bit arithmetic) and transform */
        for (int i = 0; i < d->ncomp; ++i) {
            jcomp *c = &d->comp[i];
            const int bw = (c->width + 7) >> 3, bh = (c->height + 7) >> 3;
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) {
                    int16_t *blk = c->coeff + 64 * ((size_t)by * c->blocks_w + bx);
                    for (int k = 0; k < 64; ++k) blk[k] = (int16_t)(blk[k] * d->qt[c->tq][k]);
                    idct_block(c->plane + (size_t)by * 8 * c->pitch + bx * 8, c->pitch, blk);
                }
        }
    return 1;
}

/* ---- chroma upsampling (one output row from the nearer and the farther source row) ---------------------------------- */
static const uint8_t *up_rows(uint8_t *out, const uint8_t *near_row, const uint8_t *far_row, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near_row;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near_row[i] + far_row[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near_row[0]; return out; }
        out[0] = near_row[0];
        out[1] = (uint8_t)((near_row[0] * 3 + near_row[1] + 2) >> 2);
        for (int i = 1; i < w - 1; ++i) {
            const int n = 3 * near_row[i] + 2;
            out[2 * i] = (uint8_t)((n + near_row[i - 1]) >> 2);
            out[2 * i + 1] = (uint8_t)((n + near_row[i + 1]) >> 2);
        }
        out[2 * w - 2] = (uint8_t)((near_row[w - 2] * 3 + near_row[w - 1] + 2) >> 2);
        out[2 * w - 1] = near_row[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        int prev, cur = 3 * near_row[0] + far_row[0]; /* vertical blend, 4x */
        if (w == 1) { out[0] = out[1] = (uint8_t)((cur + 2) >> 2); return out; }
        out[0] = (uint8_t)((cur + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            prev = cur;
            cur = 3 * near_row[i] + far_row[i];
            out[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((cur + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; ++i) /* other ratios: nearest neighbour along the row, the nearer row vertically */
        for (int j = 0; j < hs; ++j) out[i * hs + j] = near_row[i];
    return out;
}

#define CFIX(x) (((int)((x) * 4096.0f + 0.5f)) << 8)
static void ycc_to_rgb(uint8_t *out, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int count) {
    for (int i = 0; i < count; ++i, out += 3) {
        const int yf = (y[i] << 20) + (1 << 19), r_ = cr[i] - 128, b_ = cb[i] - 128;
        const int r = (yf + r_ * CFIX(1.40200f)) >> 20;
        const int g = (int)(yf + (r_ * -CFIX(0.71414f)) + (int)(((unsigned)(b_ * -CFIX(0.34414f))) & 0xffff0000u)) >> 20;
        const int b = (yf + b_ * CFIX(1.77200f)) >> 20;
        out[0] = clamp255(r); out[1] = clamp255(g); out[2] = clamp255(b);
    }
}

/* NULL unless the buffer holds a JPEG stream this decoder covers; the caller frees the image */
uint8_t *bip_decode_jpeg(const uint8_t *buf, size_t len, int32_t *w, int32_t *h, int32_t *depth) {
    jdec *d = (jdec *)calloc(1, sizeof(jdec));
    uint8_t *image = NULL, *line[3] = {NULL, NULL, NULL};
    if (!d) return NULL;
    d->p = buf; d->end = buf + len;
    if (!decode_planes(d)) goto done;
    const int n = d->ncomp;
    image = (uint8_t *)malloc((size_t)n * d->width * d->height + 1);
    if (!image) goto done;
    struct { int hs, vs, w_lores, ystep, ypos; const uint8_t *row0, *row1; } up[3];
    for (int k = 0; k < n; ++k) {
        line[k] = (uint8_t *)malloc((size_t)d->width + 8);
        if (!line[k]) { free(image); image = NULL; goto done; }
        up[k].hs = d->hmax / d->comp[k].h;
        up[k].vs = d->vmax / d->comp[k].v;
        up[k].ystep = up[k].vs >> 1;
        up[k].w_lores = (d->width + up[k].hs - 1) / up[k].hs;
        up[k].ypos = 0;
        up[k].row0 = up[k].row1 = d->comp[k].plane;
    }
    for (int y = 0; y < d->height; ++y) {
        const uint8_t *src[3] = {NULL, NULL, NULL};
        for (int k = 0; k < n; ++k) {
            /* in the lower half of a source row the next row is the nearer one */
            const int lower = up[k].ystep >= (up[k].vs >> 1);
            src[k] = up_rows(line[k], lower ? up[k].row1 : up[k].row0, lower ? up[k].row0 : up[k].row1, up[k].w_lores,
                             up[k].hs, up[k].vs);
            if (++up[k].ystep >= up[k].vs) {
                up[k].ystep = 0;
                up[k].row0 = up[k].row1;
                if (++up[k].ypos < d->comp[k].height) up[k].row1 += d->comp[k].pitch;
            }
        }
        uint8_t *out = image + (size_t)n * d->width * y;
        if (n == 3) ycc_to_rgb(out, src[0], src[1], src[2], d->width);
        else memcpy(out, src[0], (size_t)d->width);
    }
    *w = d->width; *h = d->height; *depth = n;
done:
    for (int k = 0; k < 3; ++k) {
        free(line[k]);
        free(d->comp[k].plane);
        free(d->comp[k].coeff);
    }
    free(d);
    return image;
}
