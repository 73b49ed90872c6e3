"""bcnn_amd -- MI355X (gfx950) back-end for bcnn's conv/GEMM hot path.

Layout:
  csrc/      hand-written HIP kernels + the C-ABI shim (-> lib/libbcnn_hip.so, declared in include/bcnn_hip.h)
  _lib.py    ctypes loader of the C-ABI (no fallback)
  ops.py     host-side mirror of the reference's per-layer interface (forward/backward of the
             conv, batchnorm, maxpool, avgpool, activation, depthwise nodes) on torch device tensors
"""
from . import _lib  # noqa: F401
