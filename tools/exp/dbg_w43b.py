"""debug: wino43b raw forward / dX on a small forced shape against torch float64; prints where the errors sit"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from bcnn_amd import ops
DEV = "cuda:0"
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(2, 16, 32, 8, 8)]
for (n, c, f, h, w) in shapes:
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand((n, c, h, w), device=DEV, generator=gen) * 2 - 1
    wt = (torch.rand((f, c, 3, 3), device=DEV, generator=gen) * 2 - 1) * (3.0 / (c * 9)) ** 0.5
    b = torch.zeros(f, device=DEV)
    Z = lambda: torch.zeros(f, device=DEV)
    bn = dict(run_mean=Z(), run_var=Z() + 1, scales=Z() + 1, saved_mean=Z(), saved_var=Z(),
              workspace=torch.full((n, f, h, w), float("nan"), device=DEV))
    y = torch.empty((n, f, h, w), device=DEV)
    ops.conv_forward(x, wt, b, y, 3, 1, 1, 1, 0, bn=bn)
    torch.cuda.synchronize()
    raw = F.conv2d(x.double().cpu(), wt.double().cpu(), None, padding=1)
    got = bn["workspace"].double().cpu()
    err = (got - raw).abs()
    bad = err > 1e-4 * raw.abs().max()
    print("shape", (n, c, f, h, w), "fwd rel", float(err.max() / raw.abs().max()), "bad", int(bad.sum()), "of", bad.numel(),
          "nan", int(torch.isnan(got).sum()))
    if bad.any():
        idx = bad.nonzero()
        print(" bad n:", sorted(set(idx[:, 0].tolist()))[:20])
        print(" bad f:", sorted(set(idx[:, 1].tolist()))[:70])
        print(" bad h:", sorted(set(idx[:, 2].tolist()))[:60])
        print(" bad w:", sorted(set(idx[:, 3].tolist()))[:60])
        i = idx[0].tolist()
        print(" first", i, float(got[tuple(i)]), float(raw[tuple(i)]))
    mean = raw.mean(dim=(0, 2, 3))
    print(" mean rel", float((bn["saved_mean"].double().cpu() - mean).abs().max() / mean.abs().max()))
