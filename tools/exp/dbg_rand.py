import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
libc = ctypes.CDLL(None)
libc.srand(7); base = [libc.rand() for _ in range(4)]
from bcnn_amd import _lib
L = _lib.load()
def probe(name, fn):
    libc.srand(7); fn(); got = [libc.rand() for _ in range(4)]
    print(name, "consumed rand()" if got != base else "clean", got[:2], base[:2])
probe("nothing", lambda: None)
probe("device_count", lambda: L.bcnn_hip_device_count())
probe("malloc small", lambda: L.bcnn_hip_malloc_f32(16))
probe("malloc 64MB", lambda: L.bcnn_hip_malloc_f32(1 << 24))
buf = L.bcnn_hip_malloc_f32(1 << 20)
import numpy as np
h = np.zeros(1 << 20, np.float32)
probe("h2d", lambda: L.bcnn_hip_memcpy_h2d(buf, h.ctypes.data, h.nbytes))
probe("fill", lambda: (L.bcnn_hip_fill_f32(buf, 1 << 20, ctypes.c_float(1.0)), L.bcnn_hip_sync()))
probe("stream create", lambda: L.bcnn_hip_stream_create())
probe("event create", lambda: L.bcnn_hip_event_create())
import time
probe("sleep 0.5", lambda: time.sleep(0.5))
