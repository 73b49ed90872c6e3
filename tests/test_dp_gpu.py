"""GPU check of the data-parallel path of the C host: two bcnn_net replicas (virtual ranks on one GPU),
each with half of the batch, gradient arenas summed (what RCCL all-reduce does across GPUs), then
bcnn_update with bcnn_set_data_parallel(rank, 2). Weights after several steps must equal the UNMODIFIED
reference trained on the whole batch (a BN-free graph, where shard-local == global semantics)."""
import numpy as np
import pytest
import torch

from oracle import ref_bind as rb

pytestmark = pytest.mark.gpu


def graph(net):
    net.conv(8, 3, 1, 1, 1, 0, rb.ACT_RELU, "input", "c1")
    net.conv(8, 3, 2, 1, 1, 0, rb.ACT_RELU, "c1", "c2")
    net.maxpool(2, 2, rb.PADDING_SAME, "c2", "p1")
    net.fullc(5, rb.ACT_NONE, "p1", "fc")
    net.softmax("fc", "sm")
    net.cost("sm", "label", "cost", 1.0)


def test_two_replicas_with_summed_arena_match_reference_on_global_batch():
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    from bcnn_amd import capi
    shp = dict(w=12, h=12, c=3)
    ref = rb.RefNet(mode=rb.MODE_TRAIN, n=4, **shp)
    ref.L.ref_set_threads(ref.net, 4)
    graph(ref)
    ref.compile()
    ref.L.bcnn_set_sgd_optimizer(ref.net, 0.05, 0.9)
    ref.L.bcnn_set_weight_regularizer(ref.net, 5e-4)
    reps = []
    for r in range(2):
        net = capi.Net(mode=capi.MODE_TRAIN, n=2, **shp)
        graph(net)
        net.compile()
        net.set_sgd(0.05, 0.9, 5e-4)
        net.set_data_parallel(r, 2)
        reps.append(net)
    nt = ref.L.ref_num_tensors(ref.net)
    names = [ref.L.ref_tensor_name(ref.net, i).decode() for i in range(nt)]
    params = [i for i in range(2, nt) if names[i].endswith("_w") or names[i].endswith("_b")]
    for i in params:
        for net in reps:
            net.data(i)[...] = ref.data(i)
            net.upload(i)
    arenas = []
    for net in reps:
        p, n = net.gradient_arena()
        arenas.append(torch.as_tensor(capi.DeviceArray(p, n), device="cuda:0"))
    rs = np.random.RandomState(11)
    for step in range(3):
        x = rs.uniform(-1, 1, (4, 3, 12, 12)).astype(np.float32)
        lab = np.zeros((4, 5, 1, 1), np.float32)
        for b in range(4):
            lab[b, rs.randint(5)] = 1.0
        ref.data(0)[...] = x
        ref.data(1)[...] = lab
        ref.forward(); ref.backward(); ref.L.bcnn_update(ref.net)
        for r, net in enumerate(reps):
            net.data(0)[...] = x[2 * r:2 * r + 2]; net.upload(0)
            net.data(1)[...] = lab[2 * r:2 * r + 2]; net.upload(1)
            net.forward(); net.backward(); net.sync()
        total = arenas[0] + arenas[1]          # the all-reduce(sum)
        arenas[0].copy_(total); arenas[1].copy_(total)
        torch.cuda.synchronize()
        for net in reps:
            net.update(); net.sync()
        for i in params:
            for net in reps:
                net.download(i, with_grad=False)
                a, b = net.data(i), ref.data(i)
                err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
                assert err < 1e-4, (step, names[i], err)
    for net in reps:
        net.close()
    ref.close()


def test_gradient_ready_ranges_tile_the_arena():
    """bcnn_set_gradient_ready_callback reports growing TAIL ranges of the gradient arena during backward:
    contiguous, descending, and together exactly the arena -- what the bucketed all-reduce in bench.py relies on."""
    from bcnn_amd import capi
    net = capi.Net(mode=capi.MODE_TRAIN, n=2, w=12, h=12, c=3)
    graph(net)
    net.compile()
    net.set_sgd(0.05, 0.9, 5e-4)
    net.set_data_parallel(0, 1)
    _, gsize = net.gradient_arena()
    seen = []
    net.set_gradient_ready_callback(lambda first, count: seen.append((first, count)))
    rs = np.random.RandomState(3)
    net.data(0)[...] = rs.uniform(-1, 1, (2, 3, 12, 12)).astype(np.float32)
    lab = np.zeros((2, 5, 1, 1), np.float32)
    lab[0, 1] = lab[1, 3] = 1.0
    net.data(1)[...] = lab
    net.upload(0); net.upload(1)
    net.forward(); net.backward(); net.sync()
    assert seen, "callback never fired"
    hi = gsize
    for first, count in seen:
        assert count > 0 and first + count == hi, (seen, gsize)
        hi = first
    assert hi == 0, (seen, gsize)
    # a second backward reports the same ranges (state is reset per call), and None removes the callback
    n_first = len(seen)
    net.forward(); net.backward(); net.sync()
    assert seen[n_first:] == seen[:n_first]
    net.set_gradient_ready_callback(None)
    net.forward(); net.backward(); net.sync()
    assert len(seen) == 2 * n_first


def _run_bench_two_ranks(extra, port, self_launch):
    """self_launch: `python bench.py --gpus 2` bare (bench.py starts its own ranks, the form the round-end driver may
    use); otherwise under torch.distributed.run the way the driver's multi-GPU contract spells it"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_TEST_SAME_DEVICE="1", BENCH_TEST_CHECKSUM="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
            "--no-cpu-baseline"] + extra
    if self_launch:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    if self_launch:
        assert len(lines) == 1, r.stdout[-2000:]     # the parent relays rank 0's line and nothing else
    return json.loads([ln for ln in lines if ln.startswith("{")][-1])


def test_bench_two_ranks_overlapped_allreduce_equals_blocking_allreduce():
    """bench.py's data-parallel step, world_size 2 (both ranks on this one GPU, gloo): the bucketed all-reduce
    queued from inside bcnn_backward must leave the parameters exactly where one blocking all-reduce after backward
    leaves them (same summation: two ranks, a + b == b + a)."""
    import socket
    ports = []
    for _ in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            ports.append(s.getsockname()[1])
    a = _run_bench_two_ranks([], ports[0], self_launch=True)
    b = _run_bench_two_ranks(["--no-overlap"], ports[1], self_launch=False)
    for d in (a, b):
        assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert a["param_checksum"] == b["param_checksum"], (a["param_checksum"], b["param_checksum"])


def test_bench_comm_inlib_two_ranks_equal_the_torch_distributed_run(tmp_path):
    """`bench.py --comm inlib`: the product's own collective (csrc/comm.hip behind bcnn_set_data_parallel_comm -- parameter
    broadcast, bucketed all-reduce from inside bcnn_backward, update ordered behind the last bucket) under the bench, two
    ranks on this one GPU over the test double of librccl (tests/fake_rccl: buffers staged through the host, exchanged as
    files, summed in rank order). The parameters after three steps have to be the ones the torch.distributed run leaves
    (gloo here): the same sums of the same two shards, a + b == b + a."""
    import os
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "fake"
    fake.mkdir()
    r = subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        os.path.join(root, "tests", "fake_rccl", "fake_rccl.c"), "-o", str(fake / "libfake_rccl.so"),
                        "-L/opt/rocm/lib", "-lamdhip64"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    ports = []
    for _ in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            ports.append(s.getsockname()[1])
    a = _run_bench_two_ranks(["--no-overlap"], ports[0], self_launch=False)
    old = {k: os.environ.get(k) for k in ("BCNN_HIP_RCCL_LIB", "FAKE_RCCL_DIR", "BENCH_COMM_ID_PATH")}
    os.environ.update(BCNN_HIP_RCCL_LIB=str(fake / "libfake_rccl.so"), FAKE_RCCL_DIR=str(tmp_path),
                      BENCH_COMM_ID_PATH=str(tmp_path / "bench.id"))
    try:
        b = _run_bench_two_ranks(["--comm", "inlib"], ports[1], self_launch=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for d in (a, b):
        assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert "in-library" in b["config"]["comm"] and "torch.distributed" in a["config"]["comm"]
    assert a["param_checksum"] == b["param_checksum"], (a["param_checksum"], b["param_checksum"])


def test_bench_rccl_path_on_one_gpu_prints_one_line_and_leaves_the_same_parameters():
    """BENCH_FORCE_DP=1 runs bench.py's data-parallel step with a real RCCL communicator of world size 1: the
    gradient-ready callback, the bucketed asynchronous all-reduce on RCCL's stream ordered against the library's
    own HIP stream, and the stream-side waits before the update. A one-rank sum is the identity, so the parameters
    must end up bit-identical to the plain single-GPU run; and stdout must carry the JSON line only (RCCL prints a
    version banner through C stdio)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra_env):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, BENCH_TEST_CHECKSUM="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                            "--batch", "16", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])
    plain = run({})
    forced = run({"BENCH_FORCE_DP": "1"})
    blocking = run({"BENCH_FORCE_DP": "1", "BENCH_NO_OVERLAP": "1"})
    assert forced["param_checksum"] == plain["param_checksum"]
    assert blocking["param_checksum"] == plain["param_checksum"]


def test_bench_refuses_more_gpus_than_the_node_has():
    """`bench.py --gpus 64` on this box: non-zero exit and a message, never a smaller run under that label"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_TEST_SAME_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not r.stdout.strip() and "64" in r.stderr
