"""The convolution has two generations of GEMM kernels: the LDS-DMA ones (conv_igemm_dma.hip, conv_dw_dma.hip)
take every shape they support, the register-staged ones (conv_igemm.hip, conv_bwd.hip) the rest. The product
library reads no environment; its EXPERIMENT build (bcnn_amd/lib/libbcnn_hip_exp.so: same sources compiled with
-DBCNN_HIP_EXPERIMENT, selected through BCNN_HIP_LIB) has the switch BCNN_HIP_NO_DMA=1 (read once per process) that
forces the second set, so the whole golden suite is replayed in a child process on that build to keep both
generations pinned to the reference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"BCNN_HIP_NO_DMA": "1"}, {"BCNN_HIP_NO_FUSED_STATS": "1"},
                                 {"BCNN_HIP_NO_SMALLC_DX": "1", "BCNN_HIP_NO_PREFETCH": "1"}, {"BCNN_HIP_NO_DW_LDS": "1"},
                                 {"BCNN_HIP_NO_DW_MARCH": "1"}, {"BCNN_HIP_BN_NO_CONSTS": "1", "BCNN_HIP_FLAT_MAP_DIV": "1"}],
                         ids=["register_staged_kernels", "unfused_bn_statistics", "implicit_gemm_dx_for_small_c",
                              "register_window_depthwise_kernels", "lds_staged_depthwise_kernels",
                              "batchnorm_constants_in_the_bodies"])
def test_golden_suite_on_the_other_code_path(env):
    e = dict(os.environ)
    e.update(env)
    e["BCNN_HIP_LIB"] = os.path.join(ROOT, "bcnn_amd", "lib", "libbcnn_hip_exp.so")
    assert os.path.exists(e["BCNN_HIP_LIB"]), "experiment build missing: __graft_entry__.build() makes it"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_parity.py"), "-m", "gpu",
                        "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


_TRAIN = r"""
import ctypes, hashlib, sys
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from bcnn_amd import capi
ctypes.CDLL(None).srand(11)
net = capi.Net(mode=capi.MODE_TRAIN, w=32, h=32, c=3, n=4)
bench.build_resnet18(net, capi, classes=10, base=16)
net.compile(); net.set_sgd(0.01, 0.9, 5e-4)
rs = np.random.RandomState(3)
for _ in range(3):
    net.data(0)[...] = rs.uniform(-1, 1, net.shape(0))
    lab = np.zeros(net.shape(1), np.float32); lab[np.arange(4), rs.randint(0, 10, 4)] = 1
    net.data(1)[...] = lab.reshape(net.shape(1))
    net.upload(0); net.upload(1)
    net.forward(); net.backward(); net.update()
net.sync()
p, n = net.parameter_arena()
print("SHA", hashlib.sha256(torch.as_tensor(capi.DeviceArray(p, n), device="cuda:0").cpu().numpy().tobytes()).hexdigest())
""" % ROOT


@pytest.mark.gpu
def test_experiment_host_runtime_with_every_gradient_fill_kept_trains_bit_identically():
    """bcnn_amd/lib/libbcnn_exp.so (host runtime with -DBCNN_HIP_EXPERIMENT, selected with BCNN_LIB) is the only build
    that reads BCNN_KEEP_ALL_GRAD_FILLS: with it every dst gradient is zeroed like in the reference (bcnn_net.c:361-375)
    and no backward worker is a 'sole writer'. Skipping the provably dead fills must not change a single bit of a
    ResNet-18-shaped training run (max-pool, eltwise, 1x1/s2 and Winograd layers all present)."""
    lib = os.path.join(ROOT, "bcnn_amd", "lib")
    assert os.path.exists(os.path.join(lib, "libbcnn_exp.so")), "experiment host runtime missing: build() makes it"
    shas = []
    # BCNN_NO_NODE_FUSION=1 on top: every worker runs alone, so the product run's convolution -> eltwise pairs (the add,
    # the activation and their backward riding on the convolution node's batch-norm sweeps, bcnn_link_conv_eltwise) are
    # pinned bit for bit against the separate workers as well
    # The stem's pooling backward folded into the convolution node's batch-norm backward takes that node's sums over the
    # pooled tensors -- the same sums in another order (pinned by tests/test_pool_bn_backward.py) -- so the fused side of
    # this bit-for-bit comparison runs without it (BCNN_NO_POOL_BWD_FUSION, experiment host runtime)
    exp = {"BCNN_LIB": os.path.join(lib, "libbcnn_exp.so"), "BCNN_HIP_LIB": os.path.join(lib, "libbcnn_hip_exp.so")}
    for env in (dict(exp, BCNN_NO_POOL_BWD_FUSION="1"), dict(exp, BCNN_KEEP_ALL_GRAD_FILLS="1", BCNN_NO_NODE_FUSION="1")):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", _TRAIN], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        shas.append([ln for ln in r.stdout.splitlines() if ln.startswith("SHA")][0])
    assert shas[0] == shas[1]


_TRAIN_MB = r"""
import ctypes, hashlib, sys
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from bcnn_amd import capi
ctypes.CDLL(None).srand(12)
net = capi.Net(mode=capi.MODE_TRAIN, w=64, h=64, c=3, n=4)
bench.build_mobilenet_v1(net, capi, classes=10)
net.compile(); net.set_sgd(0.01, 0.9, 5e-4)
rs = np.random.RandomState(4)
for _ in range(3):
    net.data(0)[...] = rs.uniform(-1, 1, net.shape(0))
    lab = np.zeros(net.shape(1), np.float32); lab[np.arange(4), rs.randint(0, 10, 4)] = 1
    net.data(1)[...] = lab.reshape(net.shape(1))
    net.upload(0); net.upload(1)
    net.forward(); net.backward(); net.update()
net.sync()
p, n = net.parameter_arena()
print("SHA", hashlib.sha256(torch.as_tensor(capi.DeviceArray(p, n), device="cuda:0").cpu().numpy().tobytes()).hexdigest())
""" % ROOT


@pytest.mark.gpu
def test_depthwise_batchnorm_backward_fusion_trains_bit_identically():
    """bcnn_link_depthwise_batchnorm (host/bcnn_layers_hot.c) lets a depthwise node and the batch-norm node behind it
    share work inside bcnn_forward / bcnn_backward. The experiment host runtime can switch the sharing off:
    BCNN_NO_NODE_FUSION=1 runs every worker on its own like the reference does, BCNN_NO_DW_STATS=1 only stops the
    forward statistics hand-off. With the SAME batch statistics and gradient sums (hand-offs off in both runs) the fused backward -- the
    batch-norm backward applied on the fly inside the depthwise kernel, no copy of the batch-norm input, gradient tensors
    not written -- performs the separate workers' operations in their order, so three SGD steps of MobileNet-v1 (64 x 64
    input: planes from 32 x 32 down to 2 x 2) must leave bit-identical parameters. (With the hand-off on, the statistics
    are the same sums in another fixed order: 1e-7 apart, which this randomly initialised 27-batch-norm net at batch 4
    amplifies to 1e-2 in the gradients -- on both sides of any comparison; that path is pinned against the reference by
    the mobilenet graph of tests/test_net_parity.py and by tests/test_depthwise_lds.py.)"""
    lib = os.path.join(ROOT, "bcnn_amd", "lib")
    exp = {"BCNN_LIB": os.path.join(lib, "libbcnn_exp.so"), "BCNN_HIP_LIB": os.path.join(lib, "libbcnn_hip_exp.so")}
    shas = []
    # BCNN_NO_BN_CONV_FUSION: the batch-norm backward sums otherwise come from the 1x1 convolution's data-gradient epilogue
    # (one partial per 64 pixels): the same sums in another order again (tests/test_bn_sums_from_conv.py pins that path)
    # BCNN_NO_DW_INSUMS: likewise the sums of a convolution node's batch-norm backward that the depthwise kernel behind it
    # leaves (tests/test_dw_insums.py pins that path)
    for extra in ({"BCNN_NO_DW_STATS": "1", "BCNN_NO_BN_CONV_FUSION": "1", "BCNN_NO_DW_INSUMS": "1"}, {"BCNN_NO_NODE_FUSION": "1"}):
        e = dict(os.environ); e.update(exp); e.update(extra)
        r = subprocess.run([sys.executable, "-c", _TRAIN_MB], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        shas.append([ln for ln in r.stdout.splitlines() if ln.startswith("SHA")][0])
    assert shas[0] == shas[1]


_PW_DW = r"""
import hashlib, sys
sys.path.insert(0, %r)
import torch
from bcnn_amd import ops
torch.manual_seed(5)
n, c, hw, f = 64, 128, 28, 256            # pointwise layer: 50176 pixels -> more than 16 splits of the reduction
x = torch.rand((n, c, hw, hw), device="cuda:0") * 2 - 1
dy = (torch.rand((n, f, hw, hw), device="cuda:0") * 2 - 1) * 0.01
w = torch.rand((f, c, 1, 1), device="cuda:0") - 0.5
y = torch.empty((n, f, hw, hw), device="cuda:0")
dw = torch.full((f, c, 1, 1), 0.25, device="cuda:0")      # accumulates onto what is there (momentum carry)
db = torch.zeros(f, device="cuda:0")
ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, hw, hw, f, 1, 1, 0, 1)), device="cuda:0")
ops.conv_backward(x, w, y, dy, None, dw, db, 1, 1, 0, 1, 0, ws)
torch.cuda.synchronize()
ref = torch.einsum("nfp,ncp->fc", dy.view(n, f, -1).double(), x.view(n, c, -1).double()) + 0.25
err = float((dw.view(f, c).double() - ref).abs().max() / ref.abs().max())
print("ERR", err)
print("SHA", hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest())
""" % ROOT


@pytest.mark.gpu
def test_sixteen_byte_weight_gradient_finalize_equals_the_four_byte_one_bit_for_bit():
    """conv_dw_dma_finalize_x4_kernel (pointwise layers: 16 bytes per thread over the splits) forms every element's sum in
    the association of conv_dw_dma_finalize_kernel: the weight gradient of a 1x1 layer must not change by a bit when the
    experiment library is told to use the four-byte kernel (BCNN_HIP_NO_DW_FINALIZE_X4), and it matches float64."""
    lib = os.path.join(ROOT, "bcnn_amd", "lib", "libbcnn_hip_exp.so")
    outs = []
    for extra in ({}, {"BCNN_HIP_NO_DW_FINALIZE_X4": "1"}):
        e = dict(os.environ, BCNN_HIP_LIB=lib, **extra)
        r = subprocess.run([sys.executable, "-c", _PW_DW], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = {ln.split()[0]: ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith(("ERR", "SHA"))}
        assert float(lines["ERR"]) < 1e-5, lines
        outs.append(lines["SHA"])
    assert outs[0] == outs[1]
