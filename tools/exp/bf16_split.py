#!/usr/bin/env python3
"""numpy prototype of the split-bf16 Winograd F(2x2,3x3) product: error against float64 for 1 / 2 / 3 bf16 parts and
the choice of part products, on ResNet-like channel counts (what tools/exp/bf16_check.py measures on the GPU kernels)."""
import numpy as np
rs = np.random.RandomState(0)
def bf16_trunc(x):
    u = x.astype(np.float32).view(np.uint32) & np.uint32(0xffff0000)
    return u.view(np.float32)
def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)
def split(x, parts, rnd):
    out = []; r = x.astype(np.float32)
    for _ in range(parts):
        h = rnd(r); out.append(h); r = (r - h).astype(np.float32)
    return out
def wino_conv(x, w, parts, terms, rnd):
    # F(2x2,3x3), one image: x [C,H,W] (H,W even), w [F,C,3,3], pad 1; GEMM in split-bf16 with fp32 accumulation emulated in fp64->fp32 per term sum
    C, H, W = x.shape; F = w.shape[0]
    Bt = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], np.float32)
    G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], np.float32)
    At = np.array([[1,1,1,0],[0,1,-1,-1]], np.float32)
    xp = np.pad(x, ((0,0),(1,1),(1,1)))
    TH, TW = H//2, W//2
    d = np.empty((C, TH, TW, 4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            d[:, :, :, i, j] = xp[:, i:i+2*TH:2, j:j+2*TW:2]
    V = np.einsum('ij,cthjk,lk->cthil', Bt, d, Bt).astype(np.float32)           # [C,TH,TW,4,4]
    U = np.einsum('ij,fcjk,lk->fcil', G, w, G).astype(np.float32)               # [F,C,4,4]
    if parts == 0:
        M = np.einsum('fcil,cthil->fthil', U.astype(np.float64), V.astype(np.float64))
        M32 = np.einsum('fcil,cthil->fthil', U, V)  # fp32 reference accumulate (numpy pairwise-ish)
        M = M32.astype(np.float64)
    else:
        Us = split(U, parts, rnd); Vs = split(V, parts, rnd)
        M = np.zeros((F, TH, TW, 4, 4), np.float64)
        for a in range(parts):
            for b in range(parts):
                if a + b < terms:
                    M += np.einsum('fcil,cthil->fthil', Us[a].astype(np.float64), Vs[b].astype(np.float64))
        M = M.astype(np.float32).astype(np.float64)
    y = np.einsum('ij,fthjk,lk->fthil', At.astype(np.float64), M, At.astype(np.float64))   # [F,TH,TW,2,2]
    return y.transpose(0,1,3,2,4).reshape(F, H, W)
def direct64(x, w):
    C, H, W = x.shape; F = w.shape[0]
    xp = np.pad(x.astype(np.float64), ((0,0),(1,1),(1,1)))
    y = np.zeros((F, H, W))
    for i in range(3):
        for j in range(3):
            y += np.einsum('fc,chw->fhw', w[:, :, i, j].astype(np.float64), xp[:, i:i+H, j:j+W])
    return y
for C, F, H in [(64, 64, 16), (256, 64, 8), (512, 32, 8)]:
    x = rs.uniform(-1, 1, (C, H, H)).astype(np.float32)
    w = (rs.uniform(-1, 1, (F, C, 3, 3)) * (3.0 / (C * 9)) ** 0.5).astype(np.float32)
    ref = direct64(x, w)
    den = np.abs(ref).max()
    print("C=%d F=%d %dx%d" % (C, F, H, H))
    print("   fp32 winograd            : %.2e" % (np.abs(wino_conv(x, w, 0, 0, None) - ref).max() / den))
    for parts, terms, rnd, name in [(2, 2, bf16_trunc, "2 parts trunc, 3 products (hh,hl,lh)"), (2, 2, bf16_rne, "2 parts rne, 3 products"),
                                    (2, 3, bf16_rne, "2 parts rne, 4 products"), (3, 3, bf16_rne, "3 parts rne, 6 products"), (3, 5, bf16_rne, "3 parts rne, 9 products"),
                                    (1, 1, bf16_rne, "plain bf16")]:
        e = np.abs(wino_conv(x, w, parts, terms, rnd) - ref).max() / den
        print("   %-40s: %.2e" % (name, e))
