/* bh/bh_string.h -- string helpers with the names and behaviour unchanged reference consumers rely on
 * (bh_strfill, bh_strsplit, bh_strstrip, bh_fgetline). Own implementation for link closure. */
#ifndef BH_STRING_H
#define BH_STRING_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bh_mem.h"
#ifdef __cplusplus
extern "C" {
#endif

/* whole next line without its terminator, heap allocated; NULL at end of file */
static inline char *bh_fgetline(FILE *fp) {
    size_t cap = 256, len = 0;
    int ch;
    char *buf;
    if (feof(fp)) return NULL;
    buf = (char *)malloc(cap);
    if (!buf) return NULL;
    while ((ch = fgetc(fp)) != EOF && ch != '\n') {
        if (len + 2 > cap) {
            char *nb = (char *)realloc(buf, cap *= 2);
            if (!nb) { free(buf); return NULL; }
            buf = nb;
        }
        buf[len++] = (char)ch;
    }
    if (ch == EOF && len == 0) { free(buf); return NULL; }
    buf[len] = '\0';
    return buf;
}

/* split at every `c`; empty pieces are dropped; returns the number of heap-allocated pieces in *arr */
static inline int bh_strsplit(char *str, char c, char ***arr) {
    int n = 0;
    const char *p = str;
    *arr = NULL;
    while (*p) {
        const char *e = strchr(p, c);
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len > 0) {
            char **na = (char **)realloc(*arr, (size_t)(n + 1) * sizeof(char *));
            if (!na) return n;
            *arr = na;
            (*arr)[n] = (char *)malloc(len + 1);
            memcpy((*arr)[n], p, len);
            (*arr)[n][len] = '\0';
            ++n;
        }
        if (!e) break;
        p = e + 1;
    }
    return n;
}

/* removes every blank, tab and newline (anywhere in the string, like the reference) */
static inline int bh_strstrip(char *s) {
    size_t k = 0, i;
    for (i = 0; s[i]; ++i)
        if (s[i] != ' ' && s[i] != '\t' && s[i] != '\n' && s[i] != '\r') s[k++] = s[i];
    s[k] = '\0';
    return 0;
}

static inline void bh_strfill(char **dst, const char *src) {
    size_t n = strlen(src) + 1;
    bh_free(*dst);
    *dst = (char *)calloc(n, sizeof(char));
    memcpy(*dst, src, n);
}
#ifdef __cplusplus
}
#endif
#endif
