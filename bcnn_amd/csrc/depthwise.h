// depthwise.h -- shapes and entry points shared by the depthwise-convolution kernels (depthwise.hip: register-window
// kernels for any shape; depthwise_lds.hip: the LDS-staged 3x3 kernels).
#pragma once
#include "conv_common.h"

namespace bcnn_hip {

struct DwShape {
    int N, C, H, W, OH, OW, ksz, stride, pad;
};

// Batch-norm coefficients of the stand-alone batch-norm node that consumes a depthwise layer's output, for the backward
// kernel that applies bcnn_batchnorm_layer.c:292-296 to the incoming gradient on the fly (device pointers, [C] each).
struct DwBnBwd {
    const float* dz;      // gradient with respect to the batch-norm OUTPUT
    const float* mean;    // saved batch mean
    const float* var;     // saved batch variance
    const float* scale;
    const float* dmean;   // written by bn_bwd_finalize_kernel
    const float* dvar;
};

// The convolution node (batch-norm + cheap activation) whose output is this layer's input, when the executor let it stop
// after its batch statistics: `x` handed to the kernels below is then that node's PRE-NORMALISATION output and every loaded
// element goes through act(bn(.)) first (bn_one of bn_math.h: the values the producer's apply sweep would have written).
struct DwBnIn {
    const float* mean;   // saved batch mean [C]; NULL: x is the input itself
    const float* var;
    const float* scale;
    const float* bias;
    int act;
};

// depthwise_lds.hip. All return false (and launch nothing) when the shape is not theirs.
bool depthwise_lds_ok(const DwShape& s);
size_t depthwise_lds_stats_floats(const DwShape& s);    // capacity a ConvStats needs for depthwise_forward_lds
size_t depthwise_lds_partial_floats(const DwShape& s);  // scratch of depthwise_backward_lds
// y = act(dwconv(x) + bias); with `stats` also the per-channel sum / sum of squares partials of y
bool depthwise_forward_lds(const float* x, const float* w, const float* bias, float* y, const DwShape& s, int act,
                           ConvStats* stats, const DwBnIn* in = nullptr);
// g = dy * act'(y) (written back over dy when `write_back`), or with `bn` g = BNbackward(bn->dz) * act'(y) and dy is not
// touched; dbias += sum g; dw += sum x * g; dx = (overwrite ? 0 : dx) + w * g
// in_sums (with `in`, overwrite): the kernel also emits the backward sums of the producer's batch-norm over the dx it
// writes -- partials[(channel * splits + i) * 2 + {S1, S2}], the layout bn_bwd_finalize consumes; out: splits (0: not emitted)
size_t depthwise_lds_in_sums_floats(const DwShape& s);
bool depthwise_backward_lds(const float* x, const float* w, const float* y, float* dy, float* dx, float* dw, float* dbias,
                            const DwShape& s, int act, int overwrite, int write_back, const DwBnBwd* bn,
                            const DwBnIn* in = nullptr, ConvStats* in_sums = nullptr);

// depthwise_march.hip: the same contracts for rows of at most 64 column groups of 4 / 2 / 1 floats, tried first by the two
// entry points above; slots per channel of their statistics / sums / partials (0: not their shape)
bool depthwise_march_ok(const DwShape& s);
size_t depthwise_march_splits(const DwShape& s);
bool depthwise_forward_march(const float* x, const float* w, const float* bias, float* y, const DwShape& s, int act,
                             ConvStats* stats, const DwBnIn* in);
bool depthwise_backward_march_takes(const float* x, const float* y, const float* dy, const float* dx, const DwShape& s, int act,
                                    const DwBnBwd* bn, const DwBnIn* in);  // false: depthwise_backward_march would launch nothing
bool depthwise_backward_march(const float* x, const float* w, const float* y, float* dy, float* dx, float* dw, float* dbias,
                              const DwShape& s, int act, int overwrite, int write_back, const DwBnBwd* bn, const DwBnIn* in,
                              ConvStats* in_sums);

}  // namespace bcnn_hip
