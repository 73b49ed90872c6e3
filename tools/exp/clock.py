import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
n, c, h, w, f, k, s, p = (int(v) for v in sys.argv[1:9])
dev = "cuda:0"
oh, ow = ops.conv_out_hw(h, w, k, s, p)
x = torch.rand((n, c, h, w), device=dev) * 2 - 1
wt = (torch.rand((f, c, k, k), device=dev) * 2 - 1) * (3.0 / (c * k * k)) ** 0.5
bias = torch.rand(f, device=dev) * 0.1
y = torch.empty((n, f, oh, ow), device=dev)
for _ in range(5): ops.conv_forward(x, wt, bias, y, k, s, p, 1, 2)
raw = C.CDLL(os.environ["BCNN_HIP_LIB"])
out = (C.c_ulonglong * 2)()
raw.bcnn_hip_debug_read_clock(out)
print("workgroup lifetime: %d shader cycles in %.2f us -> %.3f GHz" % (out[0], out[1] / 100.0, out[0] / (out[1] / 100.0) / 1e3))
