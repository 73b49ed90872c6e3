"""Host-runtime behaviours the round-1 review found untested:

* Adam under data parallelism: the learner's sample counter `seen` keys Adam's bias correction and the learning-rate
  schedules (bcnn_learner.c:29-65, 111-112). Two replicas with half the batch each (gradient arenas summed = the
  all-reduce) and bcnn_set_data_parallel(r, 2) have to reproduce the UNMODIFIED reference trained on the whole batch,
  with a `step` schedule that changes the rate inside the run.
* Nodes added after bcnn_compile_net: the next compile re-packs the parameter / gradient arenas and rebuilds the
  cached one-launch SGD table -- the late layers train exactly like in a net that was built in one go."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ref_bind as rb

pytestmark = pytest.mark.gpu

CFG = """
[network]
input_width=10
input_height=8
input_channels=3
batch_size=%d
optimizer=adam
learning_rate=0.01
momentum=0.9
decay=0.0005
beta1=0.9
beta2=0.999
decay_type=step
step=2
scale=0.5

[convolutional]
filters=8
size=3
stride=1
pad=1
function=relu
src=input
dst=conv1

[depthwise-conv]
size=3
stride=1
pad=1
function=relu
src=conv1
dst=dw1

[maxpool]
size=2
stride=2
src=dw1
dst=pool1

[connected]
output=5
src=pool1
dst=fc

[softmax]
src=fc
dst=prob

[cost]
src=prob
dst=out
loss=euclidean
metric=error
"""


def _load(L, path, mode):
    net = C.c_void_p()
    assert L.bcnn_init_net(C.byref(net), mode) == 0
    return net


def test_adam_two_replicas_match_reference_on_global_batch(tmp_path):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    from bcnn_amd import capi
    from tests.test_load_net import load_both
    (tmp_path / "g.conf").write_text(CFG % 4)
    (tmp_path / "h.conf").write_text(CFG % 2)
    C.CDLL(None).srand(20240607)
    ref, st_ref, tmp, st = load_both(str(tmp_path / "g.conf"), None, rb.MODE_TRAIN)   # reference: global batch 4
    assert st_ref == 0 and st == 0
    tmp.L.bcnn_end_net(C.byref(tmp.net))
    assert ref.L.bcnn_compile_net(ref.net) == 0
    reps = []
    for r in range(2):
        net = capi.Net.__new__(capi.Net)
        net.L, net.net = capi.lib(), C.c_void_p()
        assert net.L.bcnn_init_net(C.byref(net.net), capi.MODE_TRAIN) == 0
        net.L.bcnn_set_log_context(net.net, None, 4)
        assert net.L.bcnn_load_net(net.net, str(tmp_path / "h.conf").encode(), None) == 0   # half the batch per rank
        assert net.L.bcnn_compile_net(net.net) == 0
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
    for step in range(5):   # the step schedule halves the rate after iterations 2 and 4
        x = rs.uniform(-1, 1, ref.shape(0)).astype(np.float32)
        lab = np.zeros(ref.shape(1), np.float32)
        lab[np.arange(4), rs.randint(0, 5, 4)] = 1.0
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
                err = float(np.abs(net.data(i) - ref.data(i)).max() / max(float(np.abs(ref.data(i)).max()), 1e-30))
                assert err < 1e-4, (step, names[i], err)
    for net in reps:
        net.close()
    ref.close()


def test_nodes_added_after_compile_are_trained_after_the_next_compile():
    from bcnn_amd import capi

    def head(net):
        net.conv(8, 3, 1, 1, 1, 1, capi.ACT_RELU, "input", "c1")
        net.maxpool(2, 2, capi.PADDING_SAME, "c1", "p1")

    def tail(net):
        net.conv(8, 3, 1, 1, 1, 0, capi.ACT_RELU, "p1", "c2")
        net.fullc(5, capi.ACT_NONE, "c2", "fc")
        net.softmax("fc", "prob")
        net.cost("prob", "label", "cost", 1.0)

    shp = dict(w=12, h=12, c=3, n=4)
    C.CDLL(None).srand(5)
    whole = capi.Net(mode=capi.MODE_TRAIN, **shp)
    head(whole); tail(whole)
    whole.compile()
    whole.set_sgd(0.05, 0.9, 5e-4)
    C.CDLL(None).srand(5)                      # the same rand() draws in the same order: identical initial parameters
    grown = capi.Net(mode=capi.MODE_TRAIN, **shp)
    head(grown)
    grown.compile()                            # arenas and (after one update) the SGD table exist for the head only
    grown.set_sgd(0.05, 0.9, 5e-4)
    _, n_head = grown.parameter_arena()
    tail(grown)
    grown.compile()                            # re-pack: old members copied across, new ones moved in
    _, n_all = grown.parameter_arena()
    _, n_whole = whole.parameter_arena()
    assert n_all == n_whole and n_head < n_all
    rs = np.random.RandomState(2)
    nt = 0
    while whole.L.bcnn_peek_tensor(whole.net, nt):
        nt += 1
    for step in range(3):
        x = rs.uniform(-1, 1, whole.shape(0)).astype(np.float32)
        lab = np.zeros(whole.shape(1), np.float32)
        lab[np.arange(4), rs.randint(0, 5, 4)] = 1.0
        for net in (whole, grown):
            net.data(0)[...] = x; net.upload(0)
            net.data(1)[...] = lab; net.upload(1)
            net.forward(); net.backward(); net.update(); net.sync()
    pw, n = whole.parameter_arena()
    pg, _ = grown.parameter_arena()
    a = torch.as_tensor(capi.DeviceArray(pw, n), device="cuda:0")
    b = torch.as_tensor(capi.DeviceArray(pg, n), device="cuda:0")
    assert torch.equal(a, b)                   # same kernels on the same values: bit-identical
    for i in range(2, nt):                     # and the late layers really moved
        t = whole.tensor(i)
        if t.name and t.name.decode() in ("p1_w", "c2_w"):
            before = whole.data(i).copy()
            whole.download(i, False)
            assert np.abs(whole.data(i) - before).max() > 0
    whole.close()
    grown.close()


def test_second_consumer_added_after_compile_keeps_its_gradient_without_a_recompile():
    """bcnn_net_add_node kept the sole-writer marks of the last compile. conv -> maxpool makes the
    max-pool the only writer of d(conv output), so its backward ASSIGNS 0 + sum and the zero fill is skipped. A second
    consumer of the same tensor (a global avg-pool here) appended afterwards accumulates first (reverse node order) and
    was then overwritten. With the marks dropped in add_node the un-recompiled net equals the net built in one go."""
    from bcnn_amd import capi

    def head(net):
        net.conv(8, 3, 1, 1, 1, 0, capi.ACT_RELU, "input", "c1")
        net.maxpool(2, 2, capi.PADDING_SAME, "c1", "p1")

    def tail(net):
        net.avgpool("c1", "gap")           # second reader of c1, behind the max-pool in node order

    def grads(net, x):
        rs = np.random.RandomState(9)
        net.data(0)[...] = x; net.upload(0)
        net.forward()
        for name in ("p1", "gap"):
            i = net.index(name)
            net.download(i)
            net.grad(i)[...] = rs.uniform(-1, 1, net.shape(i)).astype(np.float32)
            net.upload(i, with_grad=True)
        net.backward(); net.sync()
        i = net.index("input_w")
        assert i >= 0
        net.download(i, with_grad=True)
        return net.grad(i).copy()

    shp = dict(w=12, h=12, c=3, n=2)
    x = np.random.RandomState(1).uniform(-1, 1, (2, 3, 12, 12)).astype(np.float32)
    C.CDLL(None).srand(5)
    whole = capi.Net(mode=capi.MODE_TRAIN, **shp)
    head(whole); tail(whole); whole.compile()
    C.CDLL(None).srand(5)
    grown = capi.Net(mode=capi.MODE_TRAIN, **shp)
    head(grown); grown.compile(); tail(grown)          # NOT recompiled
    a, b = grads(whole, x), grads(grown, x)
    assert np.abs(a).max() > 0
    assert np.array_equal(a, b)
    whole.close(); grown.close()


def test_reader_of_a_fused_away_tensor_added_after_compile_sees_its_values():
    """the fusion links of the last compile outlived bcnn_net_add_node. conv(+batch-norm, ReLU) -> maxpool
    links the pair in a TRAIN net: the convolution stops after its batch statistics, the pooling kernel normalises on the
    fly and the convolution's output tensor is never written. A global avg-pool over that tensor appended afterwards read
    stale memory until the next compile. add_node now re-derives the links on the graph with the new node in it: the
    un-recompiled net equals the net built in one go, forward and backward, over two passes."""
    from bcnn_amd import capi

    def head(net):
        net.conv(8, 3, 1, 1, 1, 1, capi.ACT_RELU, "input", "c1")   # fused batch-norm: the conv -> maxpool link applies
        net.maxpool(2, 2, capi.PADDING_SAME, "c1", "p1")

    def tail(net):
        net.avgpool("c1", "gap")

    def run(net, x, seed):
        rs = np.random.RandomState(seed)
        net.data(0)[...] = x; net.upload(0)
        net.forward()
        out = {}
        for name in ("p1", "gap"):
            i = net.index(name)
            net.download(i)
            out[name] = net.data(i).copy()
            net.grad(i)[...] = rs.uniform(-1, 1, net.shape(i)).astype(np.float32)
            net.upload(i, with_grad=True)
        net.backward(); net.sync()
        i = net.index("input_w")
        net.download(i, with_grad=True)
        out["dw"] = net.grad(i).copy()
        return out

    shp = dict(w=12, h=12, c=3, n=2)
    C.CDLL(None).srand(5)
    whole = capi.Net(mode=capi.MODE_TRAIN, **shp)
    head(whole); tail(whole); whole.compile()
    C.CDLL(None).srand(5)
    grown = capi.Net(mode=capi.MODE_TRAIN, **shp)
    head(grown); grown.compile()
    x0 = np.random.RandomState(0).uniform(-1, 1, (2, 3, 12, 12)).astype(np.float32)
    grown.data(0)[...] = x0; grown.upload(0)
    grown.forward(); grown.sync()                      # one linked pass: c1 is pending, not written
    tail(grown)                                        # NOT recompiled
    for seed in (1, 2):
        x = np.random.RandomState(seed).uniform(-1, 1, (2, 3, 12, 12)).astype(np.float32)
        a, b = run(whole, x, seed), run(grown, x, seed)
        assert np.abs(a["gap"]).max() > 0 and np.abs(a["dw"]).max() > 0
        for k in a:
            assert np.array_equal(a[k], b[k]), (seed, k)
    whole.close(); grown.close()
