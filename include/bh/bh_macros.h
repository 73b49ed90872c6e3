/* bh/bh_macros.h -- the handful of helper macros consumers of the reference expect (own implementation). */
#ifndef BH_MACROS_H
#define BH_MACROS_H
#define bh_min(a, b) (((a) < (b)) ? (a) : (b))
#define bh_max(a, b) (((a) > (b)) ? (a) : (b))
#define bh_clamp(x, lo, hi) (((x) < (lo)) ? (lo) : (((x) > (hi)) ? (hi) : (x)))
#define bh_abs(x) (((x) < 0) ? -(x) : (x))
#define bh_swap(a, b, T) do { T tmp_ = (a); (a) = (b); (b) = tmp_; } while (0)
#endif
