/* tools/fuzz_bip.c -- mutation fuzzer for libbip's image decoders (JPEG, PNG, PNM, BMP): every seed file given on the
 * command line is decoded once as it is and 400 times with 1..8 random byte / bit / truncation mutations. Built with
 * -fsanitize=address,undefined by tests/test_bip_fuzz.py (sanitizers run on the CPU build only). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "bip/bip.h"
int main(int argc, char **argv) {
    unsigned seed = 12345;
    long total = 0, ok = 0;
    for (int a = 1; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb"); if (!f) continue;
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        uint8_t *base = malloc(n); if (fread(base, 1, n, f) != (size_t)n) n = 0; fclose(f); if (n == 0) { free(base); continue; }
        for (int it = 0; it <= 400; ++it) {
            uint8_t *buf = malloc(n); memcpy(buf, base, n);
            int flips = it == 0 ? 0 : 1 + (rand_r(&seed) % 8); /* it 0: the seed itself (hand-made hostile seeds) */
            long len = n;
            for (int k = 0; k < flips; ++k) {
                long pos = rand_r(&seed) % n;
                int mode = rand_r(&seed) % 4;
                if (mode == 0) buf[pos] ^= 1 << (rand_r(&seed) % 8);
                else if (mode == 1) buf[pos] = rand_r(&seed);
                else if (mode == 2) buf[pos] = 0xff;
                else len = 1 + rand_r(&seed) % n;
            }
            uint8_t *img = NULL; int32_t w = 0, h = 0, d = 0;
            bip_status st = bip_load_image_from_memory(buf, (int)len, &img, &w, &h, &d);
            ++total;
            if (st == BIP_SUCCESS) { ++ok; volatile uint8_t x = img[(size_t)w * h * d - 1]; (void)x; free(img); }
            free(buf);
        }
        free(base);
    }
    printf("%ld mutated files, %ld decoded\n", total, ok);
    return 0;
}
