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


# ---- the file rendezvous by itself (no GPU, no RCCL): what keeps a stale id file of an earlier job from being used ----
_FETCH = r"""
import ctypes as C, os, sys
sys.path.insert(0, %r)
from bcnn_amd import _lib
L = _lib.load()
buf = C.create_string_buffer(128)
r = L.bcnn_hip_rendezvous_fetch(sys.argv[1].encode(), buf, 128, int(sys.argv[2]), int(sys.argv[3]))
print(r, buf.raw[:8].decode("latin1"))
""" % ROOT


def _publish(path, payload, world, nonce=None):
    import ctypes as C
    from bcnn_amd import _lib
    L = _lib.load()
    old = os.environ.pop("BCNN_HIP_JOB_NONCE", None)
    if nonce is not None:
        os.environ["BCNN_HIP_JOB_NONCE"] = nonce
    try:
        blob = payload.ljust(128, b"\0")
        return L.bcnn_hip_rendezvous_publish(str(path).encode(), blob, 128, world)
    finally:
        os.environ.pop("BCNN_HIP_JOB_NONCE", None)
        if old is not None:
            os.environ["BCNN_HIP_JOB_NONCE"] = old


def _fetch(path, world, timeout_ms, nonce=None, max_age=None, wait=True):
    env = dict(os.environ)
    env.pop("BCNN_HIP_JOB_NONCE", None)
    env.pop("BCNN_HIP_ID_MAX_AGE_S", None)
    if nonce is not None:
        env["BCNN_HIP_JOB_NONCE"] = nonce
    if max_age is not None:
        env["BCNN_HIP_ID_MAX_AGE_S"] = str(max_age)
    p = subprocess.Popen([sys.executable, "-c", _FETCH, str(path), str(world), str(timeout_ms)], env=env,
                         stdout=subprocess.PIPE, text=True)
    if not wait:
        return p
    out = p.communicate(timeout=120)[0].split()
    return int(out[0]), (out[1] if len(out) > 1 else "")


def test_rendezvous_fresh_record_is_fetched_and_world_mismatch_is_reported(tmp_path):
    path = tmp_path / "job.id"
    assert _publish(path, b"PAYLOAD1", 2) == 0
    assert _fetch(path, 2, 200) == (0, "PAYLOAD1")
    assert _fetch(path, 4, 200)[0] == 2
    assert _publish(tmp_path / "no" / "such" / "dir" / "id", b"x", 2) == -1
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]      # the temporary was renamed, not left behind


def test_rendezvous_ignores_a_record_of_another_job(tmp_path):
    """with a fixed id path a re-run used to read the PREVIOUS job's id before rank 0 had renamed the
    new one into place. A record with another nonce, or one that is older than the age bound, is now waited out."""
    import time
    path = tmp_path / "job.id"
    assert _publish(path, b"OLDJOB__", 2, nonce="job-41") == 0
    assert _fetch(path, 2, 300, nonce="job-42")[0] == 1                 # other nonce: as if no file existed
    assert _fetch(path, 2, 300, nonce="job-41") == (0, "OLDJOB__")
    assert _fetch(path, 2, 300)[0] == 1                                 # unset nonce != "job-41"
    # the waiting rank is started BEFORE rank 0 publishes and with the stale file in place: it must return the new id
    waiter = _fetch(path, 2, 20000, nonce="job-42", wait=False)
    time.sleep(1.0)
    assert waiter.poll() is None
    assert _publish(path, b"NEWJOB__", 2, nonce="job-42") == 0
    out = waiter.communicate(timeout=60)[0].split()
    assert (int(out[0]), out[1]) == (0, "NEWJOB__")
    # age bound: a record published 3 s ago is stale under a 1 s bound, fresh under the default
    time.sleep(3.0)
    assert _fetch(path, 2, 200, nonce="job-42", max_age=1)[0] == 1
    assert _fetch(path, 2, 200, nonce="job-42") == (0, "NEWJOB__")


@pytest.mark.gpu
def test_comm_init_removes_the_id_file_and_the_communicator_is_reference_counted(tmp_path):
    from bcnn_amd import _lib
    L = _lib.load()
    path = tmp_path / "job.id"
    L.bcnn_hip_comm_init(0, 1, str(path).encode())
    try:
        assert not path.exists()             # unlinked after the (collective) ncclCommInitRank
        L.bcnn_hip_comm_retain()
        L.bcnn_hip_comm_destroy()
        assert L.bcnn_hip_comm_world() == 1  # the second holder keeps it alive
    finally:
        L.bcnn_hip_comm_destroy()
    assert L.bcnn_hip_comm_world() == 0


@pytest.mark.gpu
def test_two_nets_share_the_communicator(tmp_path):
    """bcnn_end_net of one comm-active net must not tear the communicator down under the other."""
    import numpy as np
    from bcnn_amd import _lib, capi
    L = _lib.load()
    nets = []
    for k in range(2):
        net = capi.Net(mode=capi.MODE_TRAIN, w=8, h=8, c=3, n=2)
        net.conv(4, 3, 1, 1, act=capi.ACT_RELU, src="input", dst="c%d" % k)
        net.compile()
        net.set_data_parallel_comm(0, 1, str(tmp_path / "two.id"))
        nets.append(net)
    nets[0].close()
    assert L.bcnn_hip_comm_world() == 1
    nets[1].forward()
    nets[1].sync()
    nets[1].close()
    assert L.bcnn_hip_comm_world() == 0


@pytest.mark.gpu
def test_plain_c_program_trains_data_parallel_world2(tmp_path):
    """Two processes, two GPUs, the library's own RCCL path: both ranks end with the same parameters."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the pool hands out one)")
    exe = _build_example(tmp_path)
    env = dict(os.environ, BCNN_HIP_JOB_NONCE="w2-%d" % os.getpid())
    procs = [subprocess.Popen([exe, str(r), "2", str(tmp_path / "w2.id"), "6"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in (1, 0)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    sums = sorted(ln.split("checksum")[-1] for o in outs for ln in o[0].splitlines() if ln.startswith("rank "))
    assert len(sums) == 2 and sums[0] == sums[1], outs


@pytest.mark.gpu
def test_plain_c_program_world2_on_one_gpu_over_a_test_double_of_rccl(tmp_path):
    """World size 2 of the library's OWN collective path (bcnn_amd/csrc/comm.hip) where only one GPU exists: two processes
    of tools/dp_train.c share the device, and `librccl.so.1` resolves to tests/fake_rccl (a test double that stages the
    buffers through the host and exchanges them as files, sums in rank order). What runs for the first time before it meets
    eight GPUs: the non-zero rank's poll for the id record (started FIRST here, so it really waits), rank 0's publish +
    unlink after the collective ncclCommInitRank, the parameter broadcast, bucketed all-reduces from inside bcnn_backward
    ordered against the compute stream, and the momentum carry / weight decay split over two ranks. The ranks see
    different shards; identical parameter checksums at the end mean every gradient range was exchanged exactly once."""
    fake = tmp_path / "fake"
    fake.mkdir()
    r = subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c"), "-o", str(fake / "librccl.so.1"),
                        "-L/opt/rocm/lib", "-lamdhip64"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = _build_example(tmp_path)
    idf = tmp_path / "w2.id"
    env = dict(os.environ, BCNN_HIP_JOB_NONCE="fake-w2-%d" % os.getpid(), FAKE_RCCL_DIR=str(tmp_path),
               LD_LIBRARY_PATH=str(fake) + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    import time
    p1 = subprocess.Popen([exe, "1", "2", str(idf), "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    time.sleep(1.0)            # rank 1 is polling for a record that does not exist yet
    p0 = subprocess.Popen([exe, "0", "2", str(idf), "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    outs = [p.communicate(timeout=600) for p in (p0, p1)]
    assert p0.returncode == 0 and p1.returncode == 0, outs
    lines = sorted(ln for o in outs for ln in o[0].splitlines() if ln.startswith("rank "))
    assert len(lines) == 2 and lines[0].startswith("rank 0/2") and lines[1].startswith("rank 1/2"), outs
    sums = [ln.split("checksum")[-1] for ln in lines]
    assert sums[0] == sums[1] and "nan" not in sums[0], lines
    assert not idf.exists()    # rank 0 removed the record once the communicator stood
    # and the exchange mattered: one rank alone on its own shard ends somewhere else
    solo = subprocess.run([exe, "0", "1", "-", "5", "nocomm"], capture_output=True, text=True, timeout=600)
    assert solo.returncode == 0
    assert solo.stdout.split("checksum")[-1].strip() != sums[0].strip()
