"""Dataset readers and online augmentation (bcnn_amd/host/bcnn_data.c, bip_augment.c) -- the callers that feed the hot path
in the reference's own examples (examples/mnist: BCNN_LOAD_MNIST + shift + rotation; examples/cifar10: BCNN_LOAD_CIFAR10 +
flip + colour adjustment). Everything here is integer / byte work, so the bar is BIT-EXACT against the unmodified reference
(oracle/_ref) on the same synthetic files with the same libc rand() seed:

  CPU   the image operations of libbip (crop with negative origins, flip, rotation, contrast, brightness) and
        bcnn_apply_data_augmentation as a whole (same parameter draws, same bytes);
  GPU   the readers through the public API: MNIST idx files (wrap-around at end of file, centre crop to a smaller net
        input, VALID mode rewinds the test streams), CIFAR-10 records, classification / regression list files with PNG images
        (random crop origin from rand()), and a few training steps of the MNIST example's graph fed by the loader.
"""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import ref_bind as rb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bcnn_amd", "lib")
needs_ref = pytest.mark.skipif(not rb.available(), reason="oracle/_ref not present")
libc = C.CDLL(None)


class Aug(C.Structure):
    """bcnn_data_augmenter: same member order in the reference (src/bcnn_data.h:52-103) and in bcnn_internal.h"""
    _fields_ = ([(n, C.c_int) for n in ("range_shift_x", "range_shift_y", "random_fliph", "min_brightness",
                                        "max_brightness", "swap_to_bgr", "no_input_norm", "max_random_spots")] +
                [(n, C.c_float) for n in ("min_scale", "max_scale", "rotation_range", "min_contrast", "max_contrast",
                                          "max_distortion", "mean_r", "mean_g", "mean_b")] +
                [(n, C.c_int) for n in ("use_precomputed", "brightness", "apply_fliph", "shift_x", "shift_y")] +
                [(n, C.c_float) for n in ("rotation", "scale", "contrast", "distortion", "distortion_kx", "distortion_ky")])


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _bip_pair():
    mine = C.CDLL(os.path.join(LIB, "libbip.so"))
    ref = rb.lib()
    sz, i32, f, u8p = C.c_size_t, C.c_int32, C.c_float, C.POINTER(C.c_uint8)
    for L in (mine, ref):
        L.bip_crop_image.argtypes = [u8p, sz, sz, sz, i32, i32, u8p, sz, sz, sz, sz]
        L.bip_fliph_image.argtypes = [u8p, sz, sz, sz, sz, u8p, sz]
        L.bip_rotate_image.argtypes = [u8p, sz, sz, sz, u8p, sz, sz, sz, sz, f, i32, i32, C.c_int]
        L.bip_contrast_stretch.argtypes = [u8p, sz, sz, sz, sz, u8p, sz, f]
        L.bip_image_brightness.argtypes = [u8p, sz, sz, sz, sz, u8p, sz, i32]
    return mine, ref


@needs_ref
def test_bip_augmentation_primitives_are_byte_identical_to_the_reference():
    mine, ref = _bip_pair()
    rs = np.random.RandomState(4)
    for (w, h, d) in ((28, 28, 1), (32, 32, 3), (17, 23, 3), (9, 5, 4)):
        img = rs.randint(0, 256, (h, w, d)).astype(np.uint8)
        outs = []
        for L in (mine, ref):
            res = []
            for (x0, y0, dw, dh) in ((0, 0, w, h), (-3, 2, w, h), (4, -5, w, h), (2, 3, w - 4, h - 5), (-w - 2, 0, w, h),
                                     (w + 1, 0, w, h), (0, -h, w, h)):
                dst = np.full((dh, dw, d), 128, np.uint8)
                L.bip_crop_image(_u8(img), w, h, w * d, x0, y0, _u8(dst), dw, dh, dw * d, d)
                res.append(dst)
            dst = np.zeros_like(img)
            L.bip_fliph_image(_u8(img), w, h, d, w * d, _u8(dst), w * d)
            res.append(dst)
            for ang in (0.0, 0.1, -0.26, 0.5235988, 1.5707964, -3.0, 7.0):
                for interp in (0, 1):
                    dst = np.full_like(img, 128)
                    L.bip_rotate_image(_u8(img), w, h, w * d, _u8(dst), w, h, w * d, d, ang, w // 2, h // 2, interp)
                    res.append(dst)
            for gain in (0.5, 0.93, 1.0, 1.37, 2.5):
                dst = img.copy()
                L.bip_contrast_stretch(_u8(dst), w * d, w, h, d, _u8(dst), w * d, gain)   # in place, as the augmenter does
                res.append(dst)
            for b in (-300, -40, 0, 17, 255):
                dst = img.copy()
                L.bip_image_brightness(_u8(dst), w * d, w, h, d, _u8(dst), w * d, b)
                res.append(dst)
            outs.append(res)
        assert len(outs[0]) == len(outs[1])
        for k, (a, b) in enumerate(zip(*outs)):
            assert np.array_equal(a, b), ((w, h, d), k)


@needs_ref
@pytest.mark.parametrize("cfg", [
    dict(range_shift_x=5, range_shift_y=5, rotation_range=30.0),                                        # examples/mnist
    dict(random_fliph=1, apply_fliph=1, min_brightness=-20, max_brightness=20, min_contrast=0.8, max_contrast=1.2),
    dict(range_shift_x=4, range_shift_y=0, min_scale=0.9, max_scale=1.1, rotation_range=10.0, min_contrast=0.5,
         max_contrast=1.5, min_brightness=-50, max_brightness=10),
    dict(apply_fliph=1),                                      # the API flag alone: the reference never flips
    dict(range_shift_x=3, range_shift_y=3, use_precomputed=1, shift_x=-2, shift_y=1),
], ids=["mnist_example", "cifar10_example_with_ini_flip", "everything", "flip_flag_alone", "precomputed"])
def test_apply_data_augmentation_draws_and_bytes_match_the_reference(cfg):
    from bcnn_amd import capi
    mine = capi.lib()     # (imports torch before the HIP back-end: one HIP runtime per process, bcnn_amd/_lib.py)
    ref = rb.lib()
    rs = np.random.RandomState(7)
    for (w, h, d) in ((28, 28, 1), (32, 32, 3)):
        imgs = rs.randint(0, 256, (6, h, w, d)).astype(np.uint8)
        got = []
        for L in (mine, ref):
            L.bcnn_apply_data_augmentation.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(Aug),
                                                       C.POINTER(C.c_uint8)]
            aug = Aug(**cfg)
            libc.srand(99)
            out, draws = [], []
            for im in imgs:
                im = im.copy()
                scratch = np.zeros_like(im)
                assert L.bcnn_apply_data_augmentation(_u8(im), w, h, d, C.byref(aug), _u8(scratch)) == 0
                out.append(im)
                draws.append((aug.shift_x, aug.shift_y, aug.rotation, aug.scale, aug.contrast, aug.brightness))
            got.append((np.stack(out), draws, libc.rand()))
        assert got[0][1] == got[1][1]                 # the same parameters were drawn ...
        assert got[0][2] == got[1][2]                 # ... from the same number of rand() calls
        assert np.array_equal(got[0][0], got[1][0])   # ... and applied to the same effect


# ---- readers through the public API (device upload included) ---------------------------------------------------------
def _write_mnist(tmp, name, n, h=28, w=28, seed=0):
    rs = np.random.RandomState(seed)
    img = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    lab = rs.randint(0, 10, n).astype(np.uint8)
    pi, pl = str(tmp / (name + "-images-idx3-ubyte")), str(tmp / (name + "-labels-idx1-ubyte"))
    with open(pi, "wb") as f:
        f.write(struct.pack(">IIII", 2051, n, h, w) + img.tobytes())
    with open(pl, "wb") as f:
        f.write(struct.pack(">II", 2049, n) + lab.tobytes())
    return pi, pl


def _write_cifar(tmp, name, n, seed=0):
    rs = np.random.RandomState(seed)
    p = str(tmp / (name + ".bin"))
    with open(p, "wb") as f:
        for _ in range(n):
            f.write(bytes([int(rs.randint(0, 10))]) + rs.randint(0, 256, 3072).astype(np.uint8).tobytes())
    return p


def _pair(w, h, c, n, classes, mode=rb.MODE_TRAIN):
    """the same one-layer graph (+ label tensor of `classes` values) on the reference and on this build"""
    from bcnn_amd import capi
    nets = []
    for mod, cls in ((rb, rb.RefNet), (capi, capi.Net)):
        net = cls(mode=mode, w=w, h=h, c=c, n=n)
        net.fullc(classes, mod.ACT_NONE, "input", "fc")
        net.softmax("fc", "prob")
        net.cost("prob", "label", "cost", 1.0)
        nets.append(net)
    return nets


def _set_loader(net, kind, a, b, c, d):
    enc = lambda s: s.encode() if s else None
    net.L.bcnn_set_data_loader.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    net.L.bcnn_set_data_loader.restype = C.c_int
    return net.L.bcnn_set_data_loader(net.net, kind, enc(a), enc(b), enc(c), enc(d))


def _next(net):
    net.L.bcnn_loader_next.argtypes = [C.c_void_p]
    net.L.bcnn_loader_next.restype = C.c_int
    assert net.L.bcnn_loader_next(net.net) == 0
    return net.data(0).copy(), net.data(1).copy()


def _augment(net, **kw):
    L = net.L
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.bcnn_augment_data_with_shift.argtypes = [vp, i, i]
    L.bcnn_augment_data_with_rotation.argtypes = [vp, f]
    L.bcnn_augment_data_with_flip.argtypes = [vp, i, i]
    L.bcnn_augment_data_with_color_adjustment.argtypes = [vp, i, i, f, f]
    L.bcnn_augment_data_with_scale.argtypes = [vp, f, f]
    if "shift" in kw: L.bcnn_augment_data_with_shift(net.net, *kw["shift"])
    if "rotation" in kw: L.bcnn_augment_data_with_rotation(net.net, kw["rotation"])
    if "flip" in kw: L.bcnn_augment_data_with_flip(net.net, *kw["flip"])
    if "color" in kw: L.bcnn_augment_data_with_color_adjustment(net.net, *kw["color"])
    if "scale" in kw: L.bcnn_augment_data_with_scale(net.net, *kw["scale"])


def _compare_batches(ref, net, batches, seed):
    outs = []
    for n in (ref, net):
        libc.srand(seed)
        outs.append([_next(n) for _ in range(batches)])
    for k, ((xa, ya), (xb, yb)) in enumerate(zip(*outs)):
        assert np.array_equal(xa, xb), ("input", k)
        assert np.array_equal(ya, yb), ("label", k)
    return outs[1]


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("side", [28, 24], ids=["native_size", "centre_crop_to_24"])
def test_mnist_loader_matches_reference(tmp_path, side):
    import torch
    from bcnn_amd import capi
    tr = _write_mnist(tmp_path, "train", 37, seed=1)      # 37 samples, batches of 16: wraps around inside the 3rd batch
    te = _write_mnist(tmp_path, "t10k", 20, seed=2)
    ref, net = _pair(side, side, 1, 16, 10)
    for n in (ref, net):
        assert _set_loader(n, 0, tr[0], tr[1], te[0], te[1]) == 0
        _augment(n, shift=(5, 5), rotation=30.0)           # examples/mnist/mnist_example.c:138-139
        n.compile()
    got = _compare_batches(ref, net, 5, seed=123)
    assert len({g[0].tobytes() for g in got}) == 5          # augmentation + wrap-around: no two batches alike
    # the batch reached the device (the reference's H2D hook)
    t = net.tensor(0)
    dev = torch.as_tensor(capi.DeviceArray(t.data_gpu, t.n * t.c * t.h * t.w), device="cuda:0").cpu().numpy()
    assert np.array_equal(dev, got[-1][0].ravel())
    # VALID mode: the test streams, rewound, no augmentation; twice the same samples
    first = None
    for rep in range(2):
        for n in (ref, net):
            n.L.bcnn_set_mode(n.net, rb.MODE_VALID)
        v = _compare_batches(ref, net, 2, seed=5)
        first = first or v
        assert np.array_equal(v[0][0], first[0][0])
        for n in (ref, net):
            n.L.bcnn_set_mode(n.net, rb.MODE_TRAIN)
        _compare_batches(ref, net, 1, seed=6)
    ref.close(); net.close()


@needs_ref
@pytest.mark.gpu
def test_cifar10_loader_matches_reference(tmp_path):
    tr, te = _write_cifar(tmp_path, "data_batch_1", 21, seed=3), _write_cifar(tmp_path, "test_batch", 9, seed=4)
    for side in (32, 28):
        ref, net = _pair(side, side, 3, 8, 10)
        for n in (ref, net):
            assert _set_loader(n, 1, tr, None, te, None) == 0
            _augment(n, flip=(1, 0), color=(-20, 20, 0.8, 1.2), shift=(4, 4))   # examples/cifar10 (+ a shift)
            n.compile()
        _compare_batches(ref, net, 4, seed=77)
        for n in (ref, net):
            n.L.bcnn_set_mode(n.net, rb.MODE_VALID)
        _compare_batches(ref, net, 2, seed=78)
        ref.close(); net.close()


@needs_ref
@pytest.mark.gpu
def test_list_loaders_match_reference(tmp_path):
    bip = C.CDLL(os.path.join(LIB, "libbip.so"))
    rs = np.random.RandomState(8)
    lines_c, lines_r = [], []
    for k in range(7):
        h, w = (20, 24) if k % 2 else (16, 16)              # some images are larger than the 16x16 net input: cropped
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        p = str(tmp_path / ("img%d.png" % k))
        assert bip.bip_write_image(p.encode(), _u8(img), w, h, 3, w * 3) == 0
        lines_c.append("%s %d" % (p, k % 4))
        lines_r.append("%s %.3f %.3f %.3f" % (p, *rs.uniform(-1, 1, 3)))
    try:      # the usual case in practice: JPEG files (decoded to the same pixels as the reference's stb_image, tests/test_bip.py)
        from PIL import Image
        for k in range(3):
            p = str(tmp_path / ("photo%d.jpg" % k))
            Image.fromarray(rs.randint(0, 256, (18 + 2 * k, 22, 3)).astype(np.uint8)).save(p, "JPEG", quality=70 + 10 * k,
                                                                                          subsampling=k, progressive=bool(k & 1))
            lines_c.append("%s %d" % (p, k))
            lines_r.append("%s %.3f %.3f %.3f" % (p, *rs.uniform(-1, 1, 3)))
    except ImportError:
        pass
    lines_c.insert(3, str(tmp_path / "missing.png") + " 1")   # unreadable sample: skipped, the next one takes its slot
    (tmp_path / "c.txt").write_text("\n".join(lines_c) + "\n")
    (tmp_path / "r.txt").write_text("\n".join(lines_r) + "\n")
    for kind, path, classes in ((2, "c.txt", 4), (3, "r.txt", 3)):
        ref, net = _pair(16, 16, 3, 4, classes)
        for n in (ref, net):
            assert _set_loader(n, kind, str(tmp_path / path), None, str(tmp_path / path), None) == 0
            _augment(n, color=(-10, 30, 0.7, 1.3), rotation=20.0)
            n.compile()
        _compare_batches(ref, net, 5, seed=31)
        for n in (ref, net):
            n.L.bcnn_set_mode(n.net, rb.MODE_VALID)
        _compare_batches(ref, net, 2, seed=32)
        ref.close(); net.close()


@pytest.mark.gpu
def test_loader_errors_are_statuses_not_crashes(tmp_path):
    from bcnn_amd import capi
    net = capi.Net(mode=capi.MODE_TRAIN, w=28, h=28, c=1, n=4)
    net.L.bcnn_set_log_context(net.net, None, 4)
    assert _set_loader(net, 0, str(tmp_path / "nope"), str(tmp_path / "nope2"), None, None) != 0    # cannot open
    tr = _write_mnist(tmp_path, "train", 5)
    assert _set_loader(net, 0, tr[0], tr[1], None, None) == 0                                         # test set is optional
    assert _set_loader(net, 4, tr[0], None, None, None) != 0                                          # detection list: not built
    bad = tmp_path / "short"
    bad.write_bytes(b"\0" * 10)
    assert _set_loader(net, 0, str(bad), tr[1], None, None) != 0                                      # truncated header
    net.close()


@needs_ref
@pytest.mark.gpu
def test_mnist_example_graph_trains_from_the_loader_like_the_reference(tmp_path):
    """BASELINE configs[0]: the graph and the training set-up of examples/mnist/mnist_example.c:30-141, fed by the MNIST
    reader with the example's augmentation, a few iterations of bcnn_train_on_batch on both sides."""
    from bcnn_amd import capi
    tr = _write_mnist(tmp_path, "train", 64, seed=11)
    te = _write_mnist(tmp_path, "t10k", 32, seed=12)
    nets = []
    for mod, cls in ((rb, rb.RefNet), (capi, capi.Net)):
        libc.srand(2024)
        net = cls(mode=mod.MODE_TRAIN, w=28, h=28, c=1, n=16)
        net.conv(32, 3, 1, 1, 1, 0, mod.ACT_RELU, "input", "conv1")
        net.batchnorm("conv1", "bn1")
        net.maxpool(2, 2, mod.PADDING_SAME, "bn1", "pool1")
        net.conv(32, 3, 1, 1, 1, 0, mod.ACT_RELU, "pool1", "conv2")
        net.batchnorm("conv2", "bn2")
        net.maxpool(2, 2, mod.PADDING_SAME, "bn2", "pool2")
        net.fullc(256, mod.ACT_RELU, "pool2", "fc1")
        net.batchnorm("fc1", "bn3")
        net.fullc(10, mod.ACT_RELU, "bn3", "fc2")
        net.softmax("fc2", "softmax")
        net.cost("softmax", "label", "cost", 1.0)
        L = net.L   # mnist_example.c:127-131
        L.bcnn_set_sgd_optimizer.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.bcnn_set_learning_rate_policy.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.bcnn_set_weight_regularizer.argtypes = [C.c_void_p, C.c_float]
        L.bcnn_set_sgd_optimizer(net.net, 0.003, 0.9)
        L.bcnn_set_learning_rate_policy(net.net, 5, 0.00002, 0.0, 0.0, 50000, 40000)   # BCNN_LR_DECAY_SIGMOID
        L.bcnn_set_weight_regularizer(net.net, 0.0005)
        assert _set_loader(net, 0, tr[0], tr[1], te[0], te[1]) == 0
        _augment(net, shift=(5, 5), rotation=30.0)
        net.compile()
        nets.append(net)
    losses = []
    for net in nets:
        libc.srand(7)
        net.L.bcnn_train_on_batch.argtypes = [C.c_void_p]
        net.L.bcnn_train_on_batch.restype = C.c_float
        losses.append([net.L.bcnn_train_on_batch(net.net) for _ in range(6)])
    assert np.array_equal(nets[0].data(0), nets[1].data(0))          # both saw the same sixth batch
    a, b = np.array(losses[0]), np.array(losses[1])
    assert np.all(np.isfinite(b)) and np.abs(a - b).max() <= 1e-3 * np.abs(a).max(), (a, b)
    for net in nets:
        net.close()
