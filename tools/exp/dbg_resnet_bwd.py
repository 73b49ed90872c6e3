import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, torch.nn.functional as F
import bench
from oracle import ref_bind as rb
from bcnn_amd import capi
shp = dict(w=96, h=96, c=3, n=8)
ref = rb.RefNet(mode=rb.MODE_TRAIN, **shp); ref.L.ref_set_threads(ref.net, 8)
hip = capi.Net(mode=capi.MODE_TRAIN, **shp)
bench.build_resnet18(ref, rb, classes=10); bench.build_resnet18(hip, capi, classes=10)
ref.compile(); hip.compile()
nt = ref.L.ref_num_tensors(ref.net)
names = [ref.L.ref_tensor_name(ref.net, i).decode() for i in range(nt)]
rs = np.random.RandomState(5)
for i in range(2, nt):
    d = ref.data(i)
    if names[i].endswith("_scales"): d[...] = rs.uniform(0.8, 1.2, d.shape)
    elif names[i].endswith("_b"): d[...] = rs.uniform(-0.1, 0.1, d.shape)
    hip.data(i)[...] = d; hip.upload(i)
x = rs.uniform(-1, 1, ref.shape(0)).astype(np.float32)
lab = np.zeros(ref.shape(1), np.float32); lab[np.arange(8), rs.randint(0, 10, 8)] = 1.0
for net in (ref, hip):
    net.data(0)[...] = x; net.data(1)[...] = lab
hip.upload(0); hip.upload(1)
ref.forward(); hip.forward(); ref.backward(); hip.backward()
for i in range(nt):
    if ref.tensor(i).data: hip.download(i)
ix, iy = names.index("s4b2_c1"), names.index("s4b2_c2")
iw = [i for i in range(nt) if names[i] == "s4b2_c1_w"][-1]
print("x match", np.abs(hip.data(ix) - ref.data(ix)).max(), "dy(post) match", np.abs(hip.grad(iy) - ref.grad(iy)).max(), np.abs(ref.grad(iy)).max())
xt = torch.tensor(ref.data(ix), dtype=torch.float64); dyt = torch.tensor(ref.grad(iy), dtype=torch.float64)
wt = torch.tensor(ref.data(iw), dtype=torch.float64, requires_grad=True)
y = F.conv2d(xt, wt, None, stride=1, padding=1); y.backward(dyt)
e = wt.grad.numpy()
print("dW: |ref - expected| %.3e  |hip - expected| %.3e   max|expected| %.3e" % (np.abs(ref.grad(iw) - e).max(), np.abs(hip.grad(iw) - e).max(), np.abs(e).max()))
# per-image dy norms
print("per-image |dy| (ref):", [float(np.abs(ref.grad(iy)[n]).max()) for n in range(8)])
print("per-image |dy| (hip):", [float(np.abs(hip.grad(iy)[n]).max()) for n in range(8)])
