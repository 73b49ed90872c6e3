import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from oracle import ref_bind as rb
from bcnn_amd import capi
from tests.test_net_parity import GRAPHS
gname = sys.argv[1]
build, shp, has_cost = GRAPHS[gname]
rs = np.random.RandomState(7)
ref = rb.RefNet(mode=rb.MODE_TRAIN, **shp); ref.L.ref_set_threads(ref.net, 4)
hip = capi.Net(mode=capi.MODE_TRAIN, **shp)
build(ref); build(hip); ref.compile(); hip.compile()
nt = ref.L.ref_num_tensors(ref.net)
names = [ref.L.ref_tensor_name(ref.net, i).decode() for i in range(nt)]
for i in range(2, nt):
    d = ref.data(i)
    if names[i].endswith("_scales"): d[...] = rs.uniform(0.5, 1.5, d.shape)
    elif names[i].endswith("_b"): d[...] = rs.uniform(-0.2, 0.2, d.shape)
    hip.data(i)[...] = d; hip.upload(i)
x = rs.uniform(-1, 1, ref.shape(0)).astype(np.float32)
ref.data(0)[...] = x; hip.data(0)[...] = x; hip.upload(0)
ref.forward(); hip.forward()
last = nt - 1
dy = (rs.uniform(-1, 1, ref.shape(last)) * 0.1).astype(np.float32)
ref.grad(last)[...] = dy; hip.download(last); hip.grad(last)[...] = dy; hip.upload(last, with_grad=True)
ref.backward(); hip.backward()
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for i in range(nt):
    if not ref.tensor(i).data: continue
    hip.download(i)
    e = np.abs(hip.data(i) - ref.data(i)).max() / max(np.abs(ref.data(i)).max(), 1e-30)
    g = -1
    if ref.grad(i) is not None and i != 1:
        g = np.abs(hip.grad(i) - ref.grad(i)).max() / max(np.abs(ref.grad(i)).max(), 1e-30)
    print("%2d %-16s %-18s data %.2e grad %.2e" % (i, names[i], ref.shape(i), e, g))
    if g > 1e-3 and ref.grad(i).size <= 64:
        print("   hip", hip.grad(i).ravel()); print("   ref", ref.grad(i).ravel())
