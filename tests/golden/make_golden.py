#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/libbcnn_ref.so).

Run in the build container (where /root/reference exists) after `make -C oracle ref`:
    python tests/golden/make_golden.py
Each fixture holds a case's shape parameters, its seeded inputs (in__*) and the reference's
outputs (out__*). The reference publishes no golden vectors of its own,
so these ARE the pins: oracle/bcnn_oracle.c and the HIP path are both checked against them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_bind as rb  # noqa: E402
from oracle import ref_cases as rc  # noqa: E402

A = rb
CASES = [
    # ---- conv: plain, strides, pads, groups, quirks --------------------------------------------
    rc.make_conv(1, 2, 3, 8, 8, 8, 3, 1, 1, name="conv_k3s1p1"),
    rc.make_conv(2, 2, 4, 9, 7, 8, 3, 2, 1, act=A.ACT_RELU, name="conv_k3s2p1_relu_ragged"),
    rc.make_conv(3, 2, 8, 6, 6, 4, 1, 1, 0, name="conv_k1s1_pointwise"),
    rc.make_conv(4, 2, 4, 8, 8, 8, 1, 2, 0, name="conv_k1s2_quirk1"),
    rc.make_conv(5, 2, 3, 16, 16, 4, 7, 2, 3, act=A.ACT_LRELU, name="conv_k7s2p3_lrelu"),
    rc.make_conv(6, 2, 4, 8, 8, 8, 3, 1, 1, g=2, name="conv_groups2"),
    rc.make_conv(7, 2, 3, 8, 8, 8, 3, 1, 1, bias_one=True, name="conv_bias_one_quirk2"),
    rc.make_conv(8, 2, 3, 8, 8, 8, 3, 1, 1, bn=1, act=A.ACT_RELU, name="conv_bn_relu"),
    rc.make_conv(9, 3, 8, 5, 5, 16, 1, 1, 0, bn=1, name="conv_bn_k1"),
    rc.make_conv(10, 2, 3, 8, 8, 8, 3, 1, 1, input_grad=False, name="conv_input_layer_no_dx"),
    rc.make_conv(11, 1, 2, 12, 9, 5, 5, 1, 0, act=A.ACT_TANH, name="conv_k5p0_tanh_nonsquare"),
    rc.make_conv(12, 2, 3, 8, 8, 8, 3, 1, 1, carry=True, name="conv_momentum_carry"),
    rc.make_conv(13, 2, 3, 8, 8, 8, 3, 1, 1, bn=1, act=A.ACT_RELU, carry=True, name="conv_bn_carry"),
    rc.make_conv(14, 2, 6, 10, 10, 6, 3, 2, 1, g=3, act=A.ACT_RELU, name="conv_groups3_s2"),
    rc.make_conv(15, 2, 3, 8, 8, 8, 3, 1, 1, bn=1, act=A.ACT_RELU, mode=A.MODE_VALID, input_grad=False,
                 name="conv_bn_valid_mode"),
    rc.make_conv(16, 2, 3, 9, 9, 8, 5, 2, 2, bn=1, mode=A.MODE_PREDICT, input_grad=False,
                 name="conv_bn_predict_mode"),
    rc.make_conv(17, 4, 16, 14, 14, 32, 3, 1, 1, act=A.ACT_RELU, name="conv_c16_f32_14x14"),
    rc.make_conv(18, 2, 32, 7, 7, 64, 3, 1, 1, bn=1, act=A.ACT_RELU, name="conv_c32_f64_7x7_bn"),
    rc.make_conv(19, 1, 3, 33, 35, 64, 3, 1, 1, name="conv_stem_like_33x35"),
    rc.make_conv(20, 2, 3, 8, 8, 4, 3, 1, 0, act=A.ACT_RAMP, name="conv_k3p0_ramp"),
    rc.make_conv(21, 2, 2, 6, 6, 3, 3, 1, 2, act=A.ACT_LOGISTIC, name="conv_pad2_logistic"),
    rc.make_conv(22, 2, 2, 6, 6, 3, 3, 1, 1, act=A.ACT_CLAMP, name="conv_clamp"),
    rc.make_conv(23, 2, 2, 6, 6, 3, 3, 1, 1, act=A.ACT_ABS, name="conv_abs"),
    rc.make_conv(24, 2, 2, 6, 6, 3, 3, 1, 1, act=A.ACT_SOFTPLUS, name="conv_softplus"),
    # fused PReLU (slopes = src slot 3 + 3 * batch_norm of the node), without and with the fused batch-norm; 8 and 64 filters
    # (the second one reaches the LDS-DMA GEMM and its epilogue)
    # forward only: the reference's backward through such a node dereferences a NULL slope-gradient buffer (oracle/ref_cases.py)
    rc.make_conv(25, 2, 3, 8, 8, 8, 3, 1, 1, act=A.ACT_PRELU, forward_only=True, name="conv_prelu_fwd"),
    rc.make_conv(26, 2, 3, 8, 8, 8, 3, 1, 1, bn=1, act=A.ACT_PRELU, forward_only=True, name="conv_bn_prelu_fwd"),
    rc.make_conv(27, 3, 32, 10, 10, 64, 3, 1, 1, bn=1, act=A.ACT_PRELU, forward_only=True, name="conv_dma_bn_prelu_c32_f64_fwd"),
    rc.make_conv(28, 2, 16, 9, 9, 40, 5, 1, 2, act=A.ACT_PRELU, mode=A.MODE_PREDICT, input_grad=False, name="conv_k5_prelu_predict"),
    # ---- conv at channel counts that reach the LDS-DMA GEMM kernels (M > 32, reduction majors >= 8/16):
    #      ragged M and J tiles, stride-parity classes, 1x1 raw views, groups, fused batch-norm statistics
    rc.make_conv(101, 3, 48, 12, 12, 80, 3, 1, 1, act=A.ACT_RELU, name="conv_dma_c48_f80_k3"),
    rc.make_conv(102, 2, 40, 13, 11, 72, 3, 2, 1, name="conv_dma_c40_f72_k3s2_odd"),
    rc.make_conv(103, 2, 64, 9, 9, 64, 1, 1, 0, act=A.ACT_LRELU, name="conv_dma_c64_f64_k1"),
    rc.make_conv(104, 2, 48, 10, 10, 64, 1, 2, 0, name="conv_dma_c48_f64_k1s2_quirk1"),
    rc.make_conv(105, 2, 64, 8, 8, 128, 3, 1, 1, g=2, name="conv_dma_groups2_c64_f128"),
    rc.make_conv(106, 2, 34, 10, 10, 66, 5, 1, 2, name="conv_dma_c34_f66_k5p2"),
    rc.make_conv(107, 4, 32, 14, 14, 64, 3, 1, 1, bn=1, act=A.ACT_RELU, name="conv_dma_bn_relu_c32_f64"),
    rc.make_conv(108, 2, 33, 7, 9, 65, 3, 1, 1, name="conv_dma_odd_c33_f65"),
    rc.make_conv(109, 2, 36, 11, 11, 64, 3, 3, 0, name="conv_dma_c36_f64_k3s3"),
    rc.make_conv(110, 2, 36, 10, 10, 40, 2, 3, 0, name="conv_dma_c36_f40_k2s3_empty_classes"),
    rc.make_conv(111, 6, 64, 8, 8, 64, 3, 1, 1, bn=1, act=A.ACT_RELU, carry=True, name="conv_dma_bn_carry_c64_f64"),
    rc.make_conv(112, 2, 64, 12, 12, 96, 3, 2, 1, bn=1, act=A.ACT_RELU, name="conv_dma_bn_c64_f96_s2"),
    # PREDICT mode, 3x3 / stride 1 / one group: the reference takes its Winograd F(2x2,3x3) path here
    # (bcnn_conv_layer.c:388-436, bcnn_mat.c:1403-2138); this build runs the general kernels and has to stay within
    # the conv tolerance of that result
    rc.make_conv(140, 2, 16, 10, 12, 32, 3, 1, 1, act=A.ACT_RELU, mode=A.MODE_PREDICT, input_grad=False,
                 via_model_file=True, name="conv_predict_winograd_ref_c16_f32"),
    rc.make_conv(141, 2, 64, 9, 11, 72, 3, 1, 1, act=A.ACT_LRELU, mode=A.MODE_PREDICT, input_grad=False,
                 via_model_file=True, name="conv_predict_winograd_ref_c64_f72"),
    # ---- stand-alone batchnorm -----------------------------------------------------------------
    rc.make_bn(30, 2, 3, 5, 5, name="bn_train"),
    rc.make_bn(31, 4, 8, 7, 9, carry=True, name="bn_train_carry"),
    rc.make_bn(32, 2, 3, 5, 5, mode=A.MODE_VALID, name="bn_valid"),
    rc.make_bn(33, 2, 3, 5, 5, mode=A.MODE_PREDICT, name="bn_predict"),
    rc.make_bn(34, 3, 4, 16, 16, shift=2.0, name="bn_train_shifted_mean"),
    rc.make_bn(35, 1, 5, 1, 1, name="bn_train_1x1_n1"),
    # ---- maxpool -------------------------------------------------------------------------------
    rc.make_maxpool(40, 2, 3, 8, 8, 2, 2, A.PADDING_SAME, name="maxpool_2x2s2"),
    rc.make_maxpool(41, 2, 3, 7, 9, 3, 2, A.PADDING_SAME, name="maxpool_3x3s2_same_ragged"),
    rc.make_maxpool(42, 2, 3, 7, 9, 3, 2, A.PADDING_VALID, name="maxpool_3x3s2_valid"),
    rc.make_maxpool(43, 2, 3, 7, 9, 3, 2, A.PADDING_CAFFE, name="maxpool_3x3s2_caffe"),
    rc.make_maxpool(44, 2, 4, 8, 8, 3, 1, A.PADDING_SAME, ties=True, name="maxpool_3x3s1_ties_nan"),
    rc.make_maxpool(45, 1, 2, 5, 5, 2, 1, A.PADDING_VALID, ties=True, name="maxpool_2x2s1_ties"),
    rc.make_maxpool(46, 2, 8, 14, 14, 3, 2, A.PADDING_SAME, name="maxpool_resnet_stem_like"),
    # rows of 4k columns: the two-outputs-per-thread stride-2 kernels (16-byte loads), ties / NaN / -FLT_MAX included
    rc.make_maxpool(130, 2, 3, 12, 16, 3, 2, A.PADDING_SAME, ties=True, name="maxpool_vec_3x3s2_same_12x16_ties"),
    rc.make_maxpool(131, 2, 3, 11, 16, 3, 2, A.PADDING_VALID, name="maxpool_vec_3x3s2_valid_11x16"),
    rc.make_maxpool(132, 2, 2, 9, 12, 3, 2, A.PADDING_CAFFE, ties=True, name="maxpool_vec_3x3s2_caffe_9x12_ties"),
    rc.make_maxpool(133, 3, 2, 10, 20, 2, 2, A.PADDING_SAME, ties=True, name="maxpool_vec_2x2s2_10x20_ties"),
    # ---- avgpool -------------------------------------------------------------------------------
    rc.make_avgpool(50, 2, 3, 7, 7, name="avgpool_7x7"),
    rc.make_avgpool(51, 3, 5, 1, 1, name="avgpool_1x1"),
    rc.make_avgpool(52, 2, 4, 13, 9, name="avgpool_13x9"),
    # ---- activation map ------------------------------------------------------------------------
] + [rc.make_act(60 + a, a, name="act_%d" % a) for a in range(10)] + [
    # ---- depthwise -----------------------------------------------------------------------------
    rc.make_dw(80, 2, 4, 8, 8, 3, 1, 1, name="dw_k3s1p1"),
    rc.make_dw(81, 2, 4, 9, 7, 3, 2, 1, act=A.ACT_RELU, name="dw_k3s2p1_relu"),
    rc.make_dw(82, 2, 3, 8, 8, 3, 1, 1, input_grad=False, name="dw_no_src_grad_skips_dw"),
    rc.make_dw(83, 2, 3, 8, 8, 5, 1, 2, bias_one=True, act=A.ACT_LRELU, name="dw_k5p2_bias_one"),
    rc.make_dw(84, 1, 2, 6, 6, 3, 1, 0, name="dw_k3p0"),
    # shapes for the 4-outputs-per-thread 3x3 kernels: aligned / ragged rows, tiny planes, both strides
    rc.make_dw(120, 2, 5, 13, 11, 3, 1, 1, act=A.ACT_RELU, name="dw_vec_13x11_s1"),
    rc.make_dw(121, 2, 3, 14, 14, 3, 2, 1, act=A.ACT_RELU, name="dw_vec_14x14_s2"),
    rc.make_dw(122, 3, 4, 7, 7, 3, 1, 1, name="dw_vec_7x7_s1"),
    rc.make_dw(123, 2, 6, 12, 16, 3, 1, 1, bias_one=True, act=A.ACT_LRELU, name="dw_vec_12x16_s1_bias_one"),
    rc.make_dw(124, 2, 3, 11, 9, 3, 2, 0, name="dw_vec_11x9_s2p0"),
    rc.make_dw(125, 1, 2, 16, 16, 3, 3, 1, name="dw_k3s3_fallback"),
    # ---- raw kernels ---------------------------------------------------------------------------
    rc.make_im2col(90, 3, 8, 8, 3, 1, 1, name="im2col_k3s1p1"),
    rc.make_im2col(91, 2, 9, 7, 3, 2, 1, name="im2col_k3s2p1"),
    rc.make_im2col(92, 2, 11, 11, 5, 2, 2, name="im2col_k5s2p2"),
    rc.make_im2col(93, 1, 6, 6, 3, 1, 0, name="im2col_k3p0"),
    rc.make_gemm(100, 0, 0, 8, 50, 27, name="gemm_nn"),
    rc.make_gemm(101, 0, 1, 8, 27, 50, name="gemm_nt"),
    rc.make_gemm(102, 1, 0, 27, 50, 8, beta=0.0, name="gemm_tn_beta0"),
    rc.make_gemm(103, 1, 1, 13, 17, 19, alpha=0.5, beta=2.0, name="gemm_tt"),
    rc.make_gemm(104, 0, 0, 70, 300, 200, name="gemm_nn_multi_panel"),
    # ---- optimizer steps (next row f-1): three updates with fresh gradients added in between ---------
    rc.make_optim(150, "sgd", 1000, 37, name="optim_sgd_momentum_decay"),
    rc.make_optim(151, "sgd", 77, 5, momentum=0.0, decay=0.0, name="optim_sgd_plain"),
    rc.make_optim(152, "adam", 1000, 37, name="optim_adam"),
    rc.make_optim(153, "adam", 77, 5, lr=0.001, decay=0.0, beta1=0.8, beta2=0.99, batch=16, name="optim_adam_b16"),
]


def main():
    assert rb.available(), "build oracle/_ref first: make -C oracle ref"
    out_dir = os.path.dirname(os.path.abspath(__file__))
    names = set()
    for case in CASES:
        name = case["name"]
        assert name not in names, name
        names.add(name)
        if os.path.exists(os.path.join(out_dir, name + ".npz")) and "--force" not in sys.argv:
            continue  # fixtures are pins: existing ones are only rewritten on request
        outs = rc.run_ref(case)
        blob = {}
        for k, v in case.items():
            if isinstance(v, np.ndarray):
                blob["in__" + k] = v
            elif k != "name":
                blob["p__" + k] = np.array(v)
        for k, v in outs.items():
            blob["out__" + k] = v
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **blob)
        print("%-34s %s" % (name, " ".join("%s%s" % (k, tuple(v.shape)) for k, v in outs.items())))
    print("%d fixtures written to %s" % (len(CASES), out_dir))


if __name__ == "__main__":
    main()
