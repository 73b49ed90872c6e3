/* bh/bh_log.h -- minimal stderr logger with the reference's entry point names (bh_log, BH_LOG_*). Own
 * implementation for link closure of unchanged consumers; the library itself logs through bcnn_log. */
#ifndef BH_LOG_H
#define BH_LOG_H
#include <stdarg.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { BH_LOG_INFO = 0, BH_LOG_WARNING = 1, BH_LOG_ERROR = 2, BH_LOG_SILENT = 3 } bh_log_level;
static inline void bh_log(bh_log_level level, const char *fmt, ...) {
    static const char *tag[] = {"[INFO] ", "[WARNING] ", "[ERROR] ", ""};
    va_list ap;
    if (level >= BH_LOG_SILENT) return;
    fputs(tag[level], stderr);
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
}
#define bh_log_info(...) bh_log(BH_LOG_INFO, __VA_ARGS__)
#define bh_log_warning(...) bh_log(BH_LOG_WARNING, __VA_ARGS__)
#define bh_log_error(...) bh_log(BH_LOG_ERROR, __VA_ARGS__)
#ifdef __cplusplus
}
#endif
#endif
