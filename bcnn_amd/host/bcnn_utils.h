/* bcnn_utils.h -- compatibility include: consumers of the reference include this name; everything lives in
 * bcnn_internal.h in this build. */
#ifndef BCNN_COMPAT_BCNN_UTILS_H
#define BCNN_COMPAT_BCNN_UTILS_H
#include "bcnn_internal.h"
#endif
