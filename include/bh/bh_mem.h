/* bh/bh_mem.h -- the one memory helper unchanged reference consumers use (bh_free: free and reset the pointer).
 * Own minimal implementation for link closure (src/cli/bcnn_cl.c, examples); not part of the hot path. */
#ifndef BH_MEM_H
#define BH_MEM_H
#include <stdint.h>
#include <stdlib.h>
#ifndef bh_free
#define bh_free(buf)      \
    do {                  \
        if (buf) {        \
            free(buf);    \
            (buf) = NULL; \
        }                 \
    } while (0)
#endif
#define bh_is_aligned16(x) (!(((uintptr_t)(x)) & 15))
#define bh_is_aligned32(x) (!(((uintptr_t)(x)) & 31))
#endif
