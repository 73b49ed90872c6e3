"""RCCL behind the C-ABI (bcnn_hip_comm_init / bcnn_hip_allreduce_sum / bcnn_hip_comm_join, include/bcnn_hip.h) and the
in-library data-parallel step on top of it (bcnn_set_data_parallel_comm, include/bcnn/bcnn.h). One GPU is available
here, so the real communicator runs at world size 1: the collective is then the identity and a training run with it
must be BIT-identical to the run without (same kernels, same order -- only the event-ordered detour over the
communicator's stream is added). The N > 1 arithmetic (global batch, momentum carry split over the ranks) is what
tests/test_dp_gloo.py and tests/test_dp_gpu.py pin; the bucket sequence is checked here against the arena layout."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bcnn_amd", "lib")


def test_header_declares_and_library_exports_the_comm_entry_points():
    from bcnn_amd import _lib
    names = set(_lib.declared_symbols())
    for n in ("bcnn_hip_comm_init", "bcnn_hip_comm_destroy", "bcnn_hip_comm_world", "bcnn_hip_comm_rank",
              "bcnn_hip_allreduce_sum", "bcnn_hip_comm_join"):
        assert n in names and n in _lib.SIGNATURES
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIB, "libbcnn_hip.so")], capture_output=True,
                         text=True).stdout
    assert " T bcnn_hip_allreduce_sum" in out
    # RCCL is resolved with dlopen at comm_init: no link-time dependency that a single-process user would pay for
    ldd = subprocess.run(["ldd", os.path.join(LIB, "libbcnn_hip.so")], capture_output=True, text=True).stdout
    assert "rccl" not in ldd


@pytest.mark.gpu
def test_allreduce_world1_is_the_identity_and_ordered_with_the_compute_stream(tmp_path):
    import torch
    from bcnn_amd import _lib
    L = _lib.load()
    assert L.bcnn_hip_comm_world() == 0
    L.bcnn_hip_comm_init(0, 1, str(tmp_path / "id").encode())
    try:
        assert L.bcnn_hip_comm_world() == 1 and L.bcnn_hip_comm_rank() == 0
        x = torch.rand(5_000_001, device="cuda:0")
        ref = x.clone()
        torch.cuda.synchronize()
        import ctypes as C
        # producer on the compute stream -> collective -> consumer on the compute stream, no host sync in between
        L.bcnn_hip_scal(x.numel(), C.c_float(0.5), x.data_ptr())
        L.bcnn_hip_allreduce_sum(x.data_ptr(), x.numel())
        L.bcnn_hip_comm_join()
        L.bcnn_hip_scal(x.numel(), C.c_float(4.0), x.data_ptr())
        L.bcnn_hip_sync()
        assert torch.equal(x, ref * 0.5 * 4.0)
    finally:
        L.bcnn_hip_comm_destroy()
    assert L.bcnn_hip_comm_world() == 0


def _build_example(tmp_path):
    exe = str(tmp_path / "dp_train")
    cmd = ["gcc", "-std=gnu99", "-O1", "-DBCNN_USE_HIP", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "dp_train.c"), "-o", exe, "-L", LIB, "-lbcnn", "-lbcnn_hip",
           "-Wl,-rpath," + LIB, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_c_consumer_compiles_against_the_public_headers(tmp_path):
    exe = _build_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr


@pytest.mark.gpu
def test_plain_c_program_trains_data_parallel_world1_bit_identical(tmp_path):
    exe = _build_example(tmp_path)
    outs = []
    for extra in ([], ["nocomm"]):
        r = subprocess.run([exe, "0", "1", str(tmp_path / "job.id"), "6"] + extra, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("rank 0/1")]
        assert len(line) == 1, r.stdout
        outs.append(line[0])
    assert outs[0] == outs[1], outs            # loss and parameter checksums, printed to 9 digits
    assert "nan" not in outs[0]


@pytest.mark.gpu
def test_net_level_comm_buckets_cover_the_arena_once(tmp_path):
    """bcnn_set_data_parallel_comm on a graph whose arena exceeds one bucket: parameters after two steps equal the
    plain run's bit for bit (world 1), i.e. every range was reduced exactly once and update waited for it."""
    import numpy as np
    import torch
    import bench
    from bcnn_amd import capi

    def run(with_comm):
        import ctypes
        ctypes.CDLL(None).srand(11)
        net = capi.Net(mode=capi.MODE_TRAIN, w=64, h=64, c=3, n=4)
        bench.build_resnet18(net, capi, classes=10, base=64)   # 11 M parameters = 45 MB: six 8 MB buckets
        net.compile()
        net.set_sgd(0.01, 0.9, 5e-4)
        if with_comm:
            net.set_data_parallel_comm(0, 1, str(tmp_path / "net.id"))
        rs = np.random.RandomState(3)
        net.data(0)[...] = rs.uniform(-1, 1, net.shape(0))
        lab = np.zeros(net.shape(1), np.float32)
        lab[np.arange(4), rs.randint(0, 10, 4)] = 1
        net.data(1)[...] = lab.reshape(net.shape(1))
        net.upload(0)
        net.upload(1)
        for _ in range(2):
            net.forward()
            net.backward()
            net.update()
        ptr, n = net.parameter_arena()
        params = torch.as_tensor(capi.DeviceArray(ptr, n), device="cuda:0").clone()
        net.sync()
        net.close()
        return params
    a, b = run(False), run(True)
    assert torch.equal(a, b)
