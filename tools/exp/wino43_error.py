#!/usr/bin/env python3
"""How accurate would F(4x4, 3x3) be in fp32 on the ResNet-18 stage shapes, against the two parity bars of tests/_golden.py?
numpy only (runs anywhere). Per stage: one image, C input channels, a few output channels, random data like the parity tests;
direct convolution in float64 = truth; F(2x2,3x3) and F(4x4,3x3) with every operation rounded to fp32 (transforms as
float32 matrix products, the channel reduction as a float32 sum in channel order -- what an fp32 MFMA chain does).
Prints max|err| / max|ref| (the per-tensor norm, bar 1e-4) and the worst element against 1e-4 |ref| + 1e-5 max|ref|
(the element-wise bar, <= 1 passes)."""
import numpy as np

f32 = np.float32
# Lavin & Gray: F(2x2,3x3) and F(4x4,3x3) transform matrices
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], f32)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], f32)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], f32)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], f32)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], f32)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], f32)


def wino(x, w, BT, G, AT, m):
    """x [C][H][W] (H, W multiples of m), w [F][C][3][3]; all arithmetic in float32"""
    C, H, W = x.shape
    F = w.shape[0]
    a = m + 2
    xp = np.zeros((C, H + 2, W + 2), f32)
    xp[:, 1:-1, 1:-1] = x
    U = np.einsum("ij,fcjk,lk->fcil", G, w, G).astype(f32)            # [F][C][a][a]
    y = np.zeros((F, H, W), f32)
    for th in range(H // m):
        for tw in range(W // m):
            d = xp[:, th * m:th * m + a, tw * m:tw * m + a]
            V = np.einsum("ij,cjk,lk->cil", BT, d, BT).astype(f32)    # [C][a][a]
            M = np.zeros((F, a, a), f32)
            for c in range(C):                                         # fp32 accumulation in channel order
                M = (M + U[:, c] * V[c][None]).astype(f32)
            y[:, th * m:(th + 1) * m, tw * m:(tw + 1) * m] = np.einsum("ij,fjk,lk->fil", AT, M, AT).astype(f32)
    return y


def direct64(x, w):
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((w.shape[0], H, W))
    for kr in range(3):
        for kc in range(3):
            y += np.einsum("fc,chw->fhw", w[:, :, kr, kc].astype(np.float64), xp[:, kr:kr + H, kc:kc + W])
    return y


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    print("%-22s %-10s %-12s %12s %14s" % ("stage", "input", "algorithm", "rel (1e-4)", "element (<=1)"))
    for name, C, HW in (("64 ch 56x56", 64, 56), ("128 ch 28x28", 128, 28), ("256 ch 14x14 (pad 16)", 256, 16), ("512 ch 7x7 (pad 8)", 512, 8)):
        # "relu": the input of such a layer in the forward pass (a ReLU output: non-negative, half zeros); "zero-mean": what the data
        # gradient convolves (the gradient behind a batch-norm: white, zero mean per channel; round 6 asked whether the transforms'
        # cancellation is worse there: hardly, 4.7e-6 ... 9.2e-6 against 4.6e-6 ... 5.8e-6)
        for dist in ("relu", "zero-mean"):
            x = rs.uniform(-1, 1, (C, HW, HW)).astype(f32)
            if dist == "relu":
                x *= (x > 0)
            w = (rs.uniform(-1, 1, (8, C, 3, 3)) * (3.0 / (C * 9)) ** 0.5).astype(f32)
            ref = direct64(x, w)
            for alg, args in (("F(2x2,3x3)", (BT2, G2, AT2, 2)), ("F(4x4,3x3)", (BT4, G4, AT4, 4))):
                y = wino(x, w, *args).astype(np.float64)
                rel = np.abs(y - ref).max() / np.abs(ref).max()
                bound = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max()
                print("%-22s %-10s %-12s %12.2e %14.3f" % (name, dist, alg, rel, (np.abs(y - ref) / bound).max()), flush=True)
