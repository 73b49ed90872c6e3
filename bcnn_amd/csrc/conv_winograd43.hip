// conv_winograd43.hip -- Winograd F(4x4, 3x3) for 3x3 / s1 / p1 layers: the product rule (which layers), the weight-pack plan
// and the dispatch of forward / dX to the kernel. 36 instead of 64 multiplies per 4 x 4 outputs and channel pair: 1.78x fewer
// MFMAs than the F(2x2, 3x3) kernel of conv_winograd_fused.hip, in fp32 throughout (measured error against float64 on the
// ResNet shapes: 5e-6 ... 1.3e-5 of the tensor's largest magnitude, tools/exp/wino43_error.py; the 1e-4 bar and the
// element-wise bar of tests/_golden.py both hold).
//
//   V = B^T d B over 6 x 6 input patches (stride 4), U = G g G^T, M_xi = U_xi V_xi for the 36 positions xi, y = A^T M A.
//
// The kernel is conv_winograd43b.hip (round 6: 16 x 16 MFMA tiles, output transform in registers) for every plane of whole
// 4 x 4 tiles -- all the product rule admits (ResNet-18: the 56 x 56 and 28 x 28 stages). The round-5 kernel
// (wino43_first_form_exp.h) exists in the experiment build only: planes that are not whole tiles, and as the A/B partner.
// Only the raw form exists (no bias / activation epilogue): every eligible layer of the benchmarked graphs feeds a batch-norm
// (forward) or is a data gradient. dX of such a layer is the same convolution of dy with the rotated, transposed filter.
//
// Reference: bcnn_forward_conv_layer_cpu's Winograd branch (bcnn_conv_layer.c:388-436; F(2x2,3x3) on bcnn_mat.c:1403-2138)
// is the precedent for computing these layers in a transformed domain; the transform matrices are Lavin & Gray's F(4x4,3x3).
#include "conv_common.h"
#include "lds_dma.h"
#include "wino43_math.h"
#include "wino43_pack.h"

namespace bcnn_hip {

constexpr int W4_BT = 32;   // tiles per workgroup unit
constexpr int W4_BF = 32;   // output channels per unit
constexpr int W4_KC = 8;    // reduction channels per chunk
constexpr int W4_NP = 36;   // positions
#ifdef BCNN_HIP_EXPERIMENT
constexpr int W4_NW = 12;   // waves: three per SIMD, each owns W4_PW positions
constexpr int W4_PW = W4_NP / W4_NW;  // 3
constexpr int W4_OP = W4_NP * W4_KC * 32;  // floats of U (or V) per stage
constexpr int W4_STAGE = 2 * W4_OP;        // U then V: 73,728 bytes
#include "wino43_first_form_exp.h"  // the round-5 kernel: planes that are not whole tiles, A/B partner
#endif

// conv_winograd43b.hip: the second form of the kernel (16 x 16 MFMA tiles, output transform in registers), whole-tile planes
void wino43b_run(const float* src, const float* w, float* dst, const ConvShape& s, int dx_mode, ConvStats* stats);
int wino43b_stats_slots(const ConvShape& s);
void wino43b_pack_dims(int J, int M, int* Jpad, int* Mpad);
// which form takes a wanted layer: the second one wherever the planes are whole tiles (experiment build: BCNN_HIP_W43_FORM=1
// keeps everything on the first form)
static bool wino43_second_form(const ConvShape& s) {
    static const int forced_first = BCNN_EXP_ENV("BCNN_HIP_W43_FORM") ? (BCNN_EXP_ENV("BCNN_HIP_W43_FORM")[0] == '1') : 0;
    return !forced_first && (s.H & 3) == 0 && (s.W & 3) == 0;
}

static int g_w43_force = -1;  // experiment build: BCNN_HIP_WINOGRAD43=0/1 overrides the rule
bool wino43_wanted(const ConvShape& s, int J, int M) {
    if (s.ksz != 3 || s.stride != 1 || s.pad != 1 || s.groups != 1) return false;
#ifndef BCNN_HIP_EXPERIMENT
    if ((s.H & 3) != 0 || (s.W & 3) != 0) return false;  // the product library holds the whole-tile kernel only
#endif
    if (s.H < 3 || s.W < 3) return false;
    if (J < 16 || (J % W4_KC) != 0 || M < 16) return false;
    if ((size_t)s.N * J * s.HW * 4 >= 0x7ffffff0ull || (size_t)s.N * M * s.HW * 4 >= 0x7ffffff0ull) return false;
    if (g_w43_force < 0) {
        const char* e = BCNN_EXP_ENV("BCNN_HIP_WINOGRAD43");
        g_w43_force = e ? (e[0] == '0' ? 0 : 1) : 2;
    }
    if (g_w43_force != 2) return g_w43_force == 1;
    // enough units to fill the chip, and the channel counts the 32-wide blocks are made for
    // Measured against the F(2x2,3x3) kernel on the ResNet-18 stages (N = 128, forward with statistics / dX, ms, same box):
    // 64 ch 56 x 56 0.157 / 0.152 against 0.195 / 0.177; 128 ch 28 x 28 0.143 / 0.141 against 0.168 / 0.163; 512 ch 7 x 7
    // (four tiles cover 8 x 8, as do the sixteen 2 x 2 tiles) 0.179 / 0.172 against 0.184 / 0.181; 256 ch 14 x 14 (tiles cover
    // 16 x 16: 31 % overhang where F(2x2,3x3) has none) 0.169 / 0.168 against 0.159 / 0.157. So: planes this tiling pads no
    // more than the 2 x 2 tiling does, enough units for a full round of the chip, the channel counts the blocks are made for --
    // Inside the ResNet-18 step the 7 x 7 layers came out even or worse (forward class 1.94 against 1.91 ms, step 10.79
    // against 10.73 ms): whole-tile planes only. The RAG instantiations stay (experiment switch, tests/test_winograd43.py).
    const long long th4 = (s.H + 3) / 4, tw4 = (s.W + 3) / 4;
    if ((s.H & 3) != 0 || (s.W & 3) != 0) return false;
    const long long T = (long long)s.N * th4 * tw4;
    return J >= 64 && M >= 64 && ceil_div(T, W4_BT) * ((M + W4_BF - 1) / W4_BF) >= (long long)kCUs;
}

// what can still send a wanted layer to the F(2x2, 3x3) kernel: 16-byte rows need 16-byte aligned tensors, and the caller's
// statistics buffer has to hold one slot per channel and block of 32 tiles
static bool wino43_usable(const float* src, const float* dst, const ConvShape& s, int dx_mode, const ConvStats* stats) {
    const bool rag = (s.H & 3) != 0 || (s.W & 3) != 0;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & (rag ? 3 : 15)) != 0) return false;
    if (stats && stats->partials) {
        const long long T = (long long)s.N * ((s.H + 3) / 4) * ((s.W + 3) / 4);
        const int M = dx_mode ? s.C : s.F;
        const size_t slots = wino43_second_form(s) ? (size_t)wino43b_stats_slots(s) : (size_t)ceil_div(T, W4_BT);
        if ((size_t)M * slots * 2 > stats->capacity) return false;
    }
    return true;
}

// bcnn_hip_conv_prepack: the 36-position pack this layer's forward (dx_mode 0, raw form) / data-gradient (1) kernel will ask
// prepack_take for; false when the layer does not run here
bool wino43_pack_plan(const ConvShape& s, int dx_mode, WinoPackJob* job, size_t* floats) {
    const int J = dx_mode ? s.F : s.C, M = dx_mode ? s.C : s.F;
    if (!wino43_wanted(s, J, M)) return false;
    job->w = nullptr; job->u = nullptr;
    job->F = s.F; job->C = s.C; job->dx_mode = dx_mode;
    job->Jpad = (J + W4_KC - 1) / W4_KC * W4_KC;
    job->Mpad = (M + W4_BF - 1) / W4_BF * W4_BF;
    job->layout = 0;
    if (wino43_second_form(s)) {
        wino43b_pack_dims(J, M, &job->Jpad, &job->Mpad);
        job->layout = 1;
    }
    job->blocks = (int)ceil_div((long long)job->Jpad * job->Mpad, 256);
    job->npos = W4_NP;
    *floats = (size_t)W4_NP * job->Jpad * job->Mpad;
    return true;
}

static double w43_flops(const ConvShape& s) { return 2.0 * 36.0 * ((double)s.N * ((s.H + 3) / 4) * ((s.W + 3) / 4)) * s.C * s.F; }
// the same without the tiles' overhang on planes that are not whole tiles (7 x 7: four tiles cover 8 x 8)
static double w43_useful_flops(const ConvShape& s) { return 2.0 * 36.0 * ((double)s.N * s.H * s.W / 16.0) * s.C * s.F; }
static double w43_bytes(const ConvShape& s) {
    return 4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW);
}

// raw output only (a fused batch-norm behind it, or a caller that adds nothing): the other forms stay on F(2x2, 3x3)
bool conv_forward_winograd43(const float* x, const float* w, float* y, const ConvShape& s, int raw, ConvStats* stats) {
    if (!raw || !wino43_wanted(s, s.C, s.F) || !wino43_usable(x, y, s, 0, stats)) return false;
    KTimer kt(K_CONV_FWD_WINO43, w43_flops(s), w43_bytes(s), w43_useful_flops(s));
    if (wino43_second_form(s)) wino43b_run(x, w, y, s, 0, stats);
#ifdef BCNN_HIP_EXPERIMENT
    else wino43_run(x, w, y, s, 0, stats);
#endif
    return true;
}

bool conv_backward_data_winograd43(const float* w, const float* dy, float* dx, const ConvShape& s) {
    if (!wino43_wanted(s, s.F, s.C) || !wino43_usable(dy, dx, s, 1, nullptr)) return false;
    KTimer kt(K_CONV_DX_WINO43, w43_flops(s), w43_bytes(s), w43_useful_flops(s));
    if (wino43_second_form(s)) wino43b_run(dy, w, dx, s, 1, nullptr);
#ifdef BCNN_HIP_EXPERIMENT
    else wino43_run(dy, w, dx, s, 1, nullptr);
#endif
    return true;
}

}  // namespace bcnn_hip
