/* bcnn_eltwise_layer.h -- compatibility include: consumers of the reference include this name; everything lives in
 * bcnn_internal.h in this build. */
#ifndef BCNN_COMPAT_BCNN_ELTWISE_LAYER_H
#define BCNN_COMPAT_BCNN_ELTWISE_LAYER_H
#include "bcnn_internal.h"
#endif
