"""Adam through the whole host runtime. The only way to select it is the INI key `optimizer=adam` (quirk 6: the
bcnn_set_*_optimizer setters never touch learner->optimizer), so both the reference and this build load the same
config file, get the same parameters and inputs, and run four training steps: every weight / bias has to agree,
and so must the learning rate the `step` policy produces. Reference: bcnn_adam_update_cpu (bcnn_learner.c:106-131)
and the per-layer update functions (bcnn_conv_layer.c:810-855, bcnn_depthwise_conv_layer.c:565-610,
bcnn_fc_layer.c:303-348)."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref_bind as rb
from tests.test_load_net import load_both, same_graph

pytestmark = pytest.mark.gpu

CFG = """
[network]
input_width=10
input_height=8
input_channels=3
batch_size=4
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
bn=1
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

[conv]
filters=6
size=1
stride=1
pad=0
src=dw1
dst=pw1

[maxpool]
size=2
stride=2
src=pw1
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

TOL = 1e-4  # relative, per tensor -- the conv / batch-norm bar; Adam itself agrees to ~1e-6 (tests/golden/optim_adam*)


def test_adam_training_steps_match_reference(tmp_path):
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    from bcnn_amd import capi
    cfg = tmp_path / "adam.conf"
    cfg.write_text(CFG)
    C.CDLL(None).srand(20240607)  # the loaders' Xavier fillers draw from libc rand(): same parameters in every run
    ref, st_ref, raw, st = load_both(str(cfg), None, rb.MODE_TRAIN)
    assert st_ref == 0 and st == 0
    nt = same_graph(ref, raw)
    assert ref.L.bcnn_compile_net(ref.net) == 0 and raw.L.bcnn_compile_net(raw.net) == 0
    hip = capi.Net.__new__(capi.Net)
    hip.L, hip.net = raw.L, raw.net
    names = [ref.L.ref_tensor_name(ref.net, i).decode() for i in range(nt)]
    params = [i for i in range(2, nt) if names[i].endswith("_w") or names[i].endswith("_b")]
    assert len(params) == 8
    for i in range(2, nt):  # same (Xavier, rand()-seeded on the reference side) parameters on both sides
        if any(names[i].endswith(s) for s in ("_w", "_b", "_scales", "_run_mean", "_run_var")):
            hip.data(i)[...] = ref.data(i)
            hip.upload(i)
    rs = np.random.RandomState(11)
    for step in range(4):
        x = rs.uniform(-1, 1, ref.shape(0)).astype(np.float32)
        lab = np.zeros(ref.shape(1), np.float32)
        lab[np.arange(4), rs.randint(0, 5, 4)] = 1.0
        for net in (ref, hip):
            net.data(0)[...] = x
            net.data(1)[...] = lab
        hip.upload(0)
        hip.upload(1)
        ref.forward()
        hip.forward()
        ref.backward()
        hip.backward()
        ref.L.bcnn_update(ref.net)
        hip.update()
        for i in params:
            hip.download(i)
            err = float(np.abs(hip.data(i) - ref.data(i)).max() / np.abs(ref.data(i)).max())
            assert err < TOL, (step, names[i], err)
            if names[i].endswith("_w"):  # Adam leaves the weight gradient zeroed (no momentum carry)
                assert not hip.grad(i).any() and not ref.grad(i).any(), names[i]
    # the parameters really moved (4 steps of ~lr each)
    moved = max(float(np.abs(hip.grad(i)).max()) for i in params if names[i].endswith("_b"))
    assert moved > 0
