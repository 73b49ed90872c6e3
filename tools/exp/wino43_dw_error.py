#!/usr/bin/env python3
"""Would the F(4x4,3x3) algorithm be accurate enough for the WEIGHT gradient in fp32?  dW = G^T [ sum_tiles (A dy A^T) . (B^T d B) ] G
(the transposed bilinear algorithm), the sum over all tiles of all images accumulated in fp32 in tile order, K-split over 64
workgroups like the fused F(2x2,3x3) kernel does (partials added in float64 like its finalize kernel). numpy only.
Prints max|err| / max|ref| (bar 1e-4) and the worst element against 1e-4 |ref| + 1e-5 max|ref| for F(2x2) and F(4x4)."""
import numpy as np
from wino43_error import BT2, G2, AT2, BT4, G4, AT4, f32

def dw_wino(x, dy, BT, G, AT, m, ksplit=64):
    """x [N][C][H][W], dy [N][F][H][W] (pad 1, stride 1) -> dW [F][C][3][3], every operation rounded to fp32"""
    N, C, H, W = x.shape
    F = dy.shape[1]
    a = m + 2
    xp = np.zeros((N, C, H + 2, W + 2), f32)
    xp[:, :, 1:-1, 1:-1] = x
    TH, TW = H // m, W // m
    # all tiles: V [T][C][a][a], dM [T][F][a][a]
    d = np.stack([xp[:, :, th * m:th * m + a, tw * m:tw * m + a] for th in range(TH) for tw in range(TW)], 1)   # [N][T'][C][a][a]
    g = np.stack([dy[:, :, th * m:(th + 1) * m, tw * m:(tw + 1) * m] for th in range(TH) for tw in range(TW)], 1)
    d = d.reshape(-1, C, a, a); g = g.reshape(-1, F, m, m)
    V = np.einsum("ij,tcjk,lk->tcil", BT, d, BT).astype(f32)
    A = AT.T.copy()
    dM = np.einsum("ij,tfjk,lk->tfil", A, g, A).astype(f32)
    T = V.shape[0]
    per = (T + ksplit - 1) // ksplit
    total = np.zeros((F, C, a, a), np.float64)
    for s in range(ksplit):
        acc = np.zeros((F, C, a, a), f32)
        for t in range(s * per, min(T, (s + 1) * per)):
            acc = (acc + dM[t][:, None] * V[t][None]).astype(f32)
        total += acc
    dU = total.astype(f32)
    return np.einsum("ji,fcjk,kl->fcil", G, dU, G).astype(f32)

def dw_direct64(x, dy):
    N, C, H, W = x.shape
    xp = np.zeros((N, C, H + 2, W + 2)); xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((dy.shape[1], C, 3, 3))
    for kr in range(3):
        for kc in range(3):
            out[:, :, kr, kc] = np.einsum("nfhw,nchw->fc", dy.astype(np.float64), xp[:, :, kr:kr + H, kc:kc + W])
    return out

rs = np.random.RandomState(1)
print("%-26s %-12s %12s %14s" % ("shape", "algorithm", "rel (1e-4)", "element (<=1)"))
for name, N, C, F, HW in (("N=32, 56x56", 32, 6, 6, 56), ("N=128, 28x28", 128, 6, 6, 28)):
    x = rs.uniform(-1, 1, (N, C, HW, HW)).astype(f32); x *= (x > 0)          # a ReLU output
    dy = (rs.standard_normal((N, F, HW, HW)) * 1e-3).astype(f32)               # a gradient behind a batch-norm
    dy -= dy.mean(axis=(0, 2, 3), keepdims=True)
    ref = dw_direct64(x, dy)
    for alg, args in (("F(2x2,3x3)", (BT2, G2, AT2, 2)), ("F(4x4,3x3)", (BT4, G4, AT4, 4))):
        y = dw_wino(x, dy, *args).astype(np.float64)
        rel = np.abs(y - ref).max() / np.abs(ref).max()
        bound = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max()
        print("%-26s %-12s %12.2e %14.3f" % (name, alg, rel, (np.abs(y - ref) / bound).max()), flush=True)
