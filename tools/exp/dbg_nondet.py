import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
use_torch = os.environ.get("DBG_TORCH", "1") == "1"
if use_torch:
    import torch
from bcnn_amd import capi
ctypes.CDLL(None).srand(7)
net = capi.Net(mode=capi.MODE_TRAIN, w=32, h=32, c=3, n=8)
net.conv(64, 3, 1, 1, 1, 1, capi.ACT_RELU, "input", "c1")
net.maxpool(2, 2, capi.PADDING_SAME, "c1", "p1")
net.conv(64, 3, 1, 1, 1, 1, capi.ACT_RELU, "p1", "c2")
net.avgpool("c2", "gap")
net.fullc(10, capi.ACT_NONE, "gap", "fc")
net.softmax("fc", "prob")
net.cost("prob", "label", "cost", 1.0)
net.compile()
net.set_sgd(0.05, 0.9, 5e-4)
rs = np.random.RandomState(0)
net.data(0)[...] = rs.uniform(-1, 1, net.shape(0))
lab = np.zeros(net.shape(1), np.float32).reshape(8, 10); lab[np.arange(8), rs.randint(0, 10, 8)] = 1
net.data(1)[...] = lab.reshape(net.shape(1))
net.upload(0); net.upload(1)
nt = 0
while True:
    try:
        t = net.L.bcnn_peek_tensor(net.net, nt)
        if not t: break
        nt += 1
    except Exception:
        break
phase = sys.argv[1] if len(sys.argv) > 1 else "all"
if phase in ("fwd", "bwd", "all"): net.forward()
if phase in ("bwd", "all"): net.backward()
if phase == "all": net.update()
for i in range(nt):
    t = net.tensor(i)
    if not t.data: continue
    net.download(i)
    d = net.data(i).astype(np.float64)
    g = net.grad(i)
    print(i, t.name.decode(), "%.10e" % d.sum(), "%.10e" % (g.astype(np.float64).sum() if g is not None else 0.0))
