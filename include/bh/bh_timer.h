/* bh/bh_timer.h -- minimal monotonic stopwatch with the interface the reference examples use
 * (bh_timer, bh_timer_start/stop/get_msec/get_usec). Own implementation for link closure of unchanged
 * consumers; not part of the hot path. */
#ifndef BH_TIMER_H
#define BH_TIMER_H
#include <time.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    struct timespec begin;
    struct timespec end;
} bh_timer;
static inline void bh_timer_start(bh_timer *t) { clock_gettime(CLOCK_MONOTONIC, &t->begin); }
static inline void bh_timer_stop(bh_timer *t) { clock_gettime(CLOCK_MONOTONIC, &t->end); }
static inline double bh_timer_get_usec(const bh_timer *t) {
    return (double)(t->end.tv_sec - t->begin.tv_sec) * 1e6 + (double)(t->end.tv_nsec - t->begin.tv_nsec) * 1e-3;
}
static inline double bh_timer_get_msec(const bh_timer *t) { return bh_timer_get_usec(t) * 1e-3; }
#ifdef __cplusplus
}
#endif
#endif
