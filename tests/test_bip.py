"""libbip.so -- the slice of the reference's image library that its unchanged consumers link against
(include/bip/bip.h): PNG writer, PNG / PNM / BMP reader, fixed-point bilinear resize. Checked against the
reference's own bip (compiled into oracle/_ref from src/bip/src/bip.c + stb) where that library is present:
the resize is bit-identical, and images written by either side are read back identically by the other."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import ref_bind as rb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8p = C.POINTER(C.c_uint8)


def _bind(L):
    L.bip_write_image.argtypes = [C.c_char_p, u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.bip_write_image.restype = C.c_int
    L.bip_load_image.argtypes = [C.c_char_p, C.POINTER(u8p), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32)]
    L.bip_load_image.restype = C.c_int
    L.bip_resize_bilinear.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_size_t, u8p, C.c_size_t, C.c_size_t, C.c_size_t,
                                      C.c_size_t]
    L.bip_resize_bilinear.restype = C.c_int
    return L


@pytest.fixture(scope="module")
def ours():
    from bcnn_amd import capi
    capi.build()
    return _bind(C.CDLL(os.path.join(ROOT, "bcnn_amd", "lib", "libbip.so")))


@pytest.fixture(scope="module")
def ref():
    if not rb.available():
        pytest.skip("oracle/_ref not present")
    return _bind(C.CDLL(rb.REF_SO))


def _ptr(a):
    return a.ctypes.data_as(u8p)


def _load(L, path):
    p, w, h, c = u8p(), C.c_int32(), C.c_int32(), C.c_int32()
    st = L.bip_load_image(str(path).encode(), C.byref(p), C.byref(w), C.byref(h), C.byref(c))
    if st != 0:
        return st, None
    img = np.ctypeslib.as_array(p, shape=(h.value, w.value, c.value)).copy()
    C.CDLL(None).free(p)
    return 0, img


@pytest.mark.parametrize("sw,sh,dw,dh,depth", [(37, 23, 224, 224, 3), (640, 480, 224, 224, 3), (224, 224, 224, 224, 3),
                                               (300, 200, 17, 31, 1), (64, 64, 129, 65, 4), (2, 2, 9, 9, 2)])
def test_resize_bilinear_is_bit_identical_to_the_reference(ours, ref, sw, sh, dw, dh, depth):
    rs = np.random.RandomState(sw * 7 + dh)
    stride = sw * depth + 5                                  # strides in bytes, with slack
    src = rs.randint(0, 256, (sh, stride)).astype(np.uint8)
    a = np.full((dh, dw * depth + 3), 77, np.uint8)
    b = a.copy()
    assert ours.bip_resize_bilinear(_ptr(src), sw, sh, stride, _ptr(a), dw, dh, a.shape[1], depth) == 0
    assert ref.bip_resize_bilinear(_ptr(src), sw, sh, stride, _ptr(b), dw, dh, b.shape[1], depth) == 0
    assert np.array_equal(a, b)
    assert np.all(a[:, dw * depth:] == 77)                   # nothing written past a row


def test_resize_rejects_bad_arguments(ours):
    img = np.zeros((4, 4), np.uint8)
    assert ours.bip_resize_bilinear(None, 4, 4, 4, _ptr(img), 2, 2, 2, 1) == 1      # BIP_INVALID_PTR
    assert ours.bip_resize_bilinear(_ptr(img), 0, 4, 4, _ptr(img), 2, 2, 2, 1) == 2  # BIP_INVALID_SIZE
    assert ours.bip_resize_bilinear(_ptr(img), 4, 4, 4, _ptr(img), 2, 2, 2, 5) == 3  # BIP_INVALID_PARAMETER


@pytest.mark.parametrize("depth", [1, 3, 4])
def test_png_written_here_is_read_back_by_both_sides(ours, ref, tmp_path, depth):
    rs = np.random.RandomState(depth)
    w, h = 301, 270                                          # > 65535 bytes: several stored deflate blocks
    img = rs.randint(0, 256, (h, w, depth)).astype(np.uint8)
    path = tmp_path / "ours.png"
    assert ours.bip_write_image(str(path).encode(), _ptr(img), w, h, depth, w * depth) == 0
    for L in (ours, ref):
        st, got = _load(L, path)
        assert st == 0 and np.array_equal(got, img)


@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_png_written_by_the_reference_is_read_here(ours, ref, tmp_path, depth):
    """stb's writer emits real deflate streams (fixed Huffman codes, matches, filtered scanlines)"""
    w, h = 123, 77
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * (c + 1) + yy * 3) % 251 for c in range(depth)], axis=-1).astype(np.uint8)  # compressible
    img[10:30, 20:60] = 200
    path = tmp_path / "ref.png"
    assert ref.bip_write_image(str(path).encode(), _ptr(np.ascontiguousarray(img)), w, h, depth, w * depth) == 0
    st, got = _load(ours, path)
    assert st == 0 and np.array_equal(got, img)


def test_png_with_dynamic_huffman_blocks_and_all_filters(ours, tmp_path):
    """zlib level 9 output (dynamic codes) with every PNG filter type, hand-assembled"""
    import zlib
    w, h, ch = 64, 40, 3
    rs = np.random.RandomState(9)
    img = (rs.randint(0, 4, (h, w, ch)) * 60 + np.arange(w)[None, :, None]).astype(np.uint8)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        f = y % 5
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        ul = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if f == 0:
            enc = cur
        elif f == 1:
            enc = cur - left
        elif f == 2:
            enc = cur - prev
        elif f == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = cur - pred
        raw.append(f)
        raw += bytes((enc & 255).astype(np.uint8))
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    z = zlib.compress(bytes(raw), 9)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
           chunk(b"IDAT", z[:100]) + chunk(b"IDAT", z[100:]) + chunk(b"IEND", b""))
    path = tmp_path / "dyn.png"
    path.write_bytes(png)
    st, got = _load(ours, path)
    assert st == 0 and np.array_equal(got, img)


def test_pnm_and_bmp_and_unsupported(ours, tmp_path):
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (5, 7, 3)).astype(np.uint8)
    (tmp_path / "a.ppm").write_bytes(b"P6\n# comment\n7 5\n255\n" + img.tobytes())
    st, got = _load(ours, tmp_path / "a.ppm")
    assert st == 0 and np.array_equal(got, img)
    (tmp_path / "a.pgm").write_bytes(b"P5 7 5 255\n" + img[:, :, 0].tobytes())
    st, got = _load(ours, tmp_path / "a.pgm")
    assert st == 0 and np.array_equal(got[:, :, 0], img[:, :, 0])
    (tmp_path / "a3.ppm").write_text("P3\n7 5\n255\n" + " ".join(str(int(v)) for v in img.reshape(-1)) + "\n")
    st, got = _load(ours, tmp_path / "a3.ppm")
    assert st == 0 and np.array_equal(got, img)
    stride = (7 * 3 + 3) & ~3
    rows = b"".join(bytes(img[y, :, ::-1].tobytes()).ljust(stride, b"\0") for y in range(4, -1, -1))
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 7, 5, 1, 24, 0,
                                                                               len(rows), 2835, 2835, 0, 0)
    (tmp_path / "a.bmp").write_bytes(hdr + rows)
    st, got = _load(ours, tmp_path / "a.bmp")
    assert st == 0 and np.array_equal(got, img)
    (tmp_path / "a.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 64)                  # a JPEG signature with nothing behind it
    st, _ = _load(ours, tmp_path / "a.jpg")
    assert st == 4                                           # BIP_UNKNOWN_ERROR, like the reference on an undecodable file
    st, _ = _load(ours, tmp_path / "missing.png")
    assert st == 4


# ---- JPEG: byte-identical to the reference's decoder (stb_image 2.08 behind bip_load_image) ----------------------------
def _jpeg_cases():
    """(name, kwargs for PIL's encoder, mode, size, kind of content)"""
    cases = []
    for size in ((64, 48), (33, 17), (8, 8), (1, 1), (17, 33), (250, 121)):
        for kind in ("photo", "noise"):
            for q in (5, 50, 90, 100):
                for prog in (False, True):
                    cases.append(("grey", dict(quality=q, progressive=prog), "L", size, kind))
                    for sub in (0, 1, 2):   # 4:4:4, 4:2:2, 4:2:0
                        cases.append(("ycc%d" % sub, dict(quality=q, progressive=prog, subsampling=sub), "RGB", size, kind))
    for size in ((48, 40), (37, 29)):       # 4:1:1 (chroma at a quarter of the width: the nearest-neighbour upsampler)
        for prog in (False, True):
            cases.append(("ycc411", dict(quality=80, progressive=prog, subsampling="4:1:1"), "RGB", size, "photo"))
    for size in ((40, 24), (37, 29)):       # restart intervals: the predictors and the bit reservoir are reset at RSTn
        for prog in (False, True):
            for rst in (1, 3):
                cases.append(("restart", dict(quality=75, progressive=prog, subsampling=2, restart_marker_blocks=rst), "RGB",
                              size, "photo"))
    return cases


def test_jpeg_pixels_are_byte_identical_to_the_reference(ours, ref, tmp_path):
    """The lossy stages of JPEG (inverse DCT, chroma upsampling, colour conversion) are implementation-defined; a drop-in
    loader has to reproduce the reference's. ~400 files from PIL's encoder: baseline and progressive, three subsamplings,
    grey, qualities 5..100, sizes that are not multiples of the MCU, restart intervals; plus two photographs written by
    other encoders (sklearn's sample images)."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 24
    rs = np.random.RandomState(0)
    photo = None
    for cand in ("sklearn/datasets/images/china.jpg", "sklearn/datasets/images/flower.jpg"):
        for base in __import__("sys").path:
            p = os.path.join(base, cand)
            if os.path.exists(p):
                photo = photo or Image.open(p).convert("RGB")
                st, a = _load(ours, p)
                st2, b = _load(ref, p)
                assert st == 0 and st2 == 0 and a.shape == b.shape and np.array_equal(a, b), cand
                break
    n = 0
    for name, kw, mode, size, kind in _jpeg_cases():
        w, h = size
        if kind == "photo" and photo is not None:
            img = photo.resize(size)
        else:
            img = Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8))
        path = tmp_path / "t.jpg"
        try:
            img.convert(mode).save(str(path), "JPEG", **kw)
        except (TypeError, ValueError, OSError):       # an encoder option this PIL does not have
            continue
        st, a = _load(ours, path)
        st2, b = _load(ref, path)
        assert st == 0 and st2 == 0, (name, kw, size)
        assert a.shape == b.shape == (h, w, 1 if mode == "L" else 3), (name, kw, size, a.shape, b.shape)
        assert np.array_equal(a, b), (name, kw, size, int(np.abs(a.astype(int) - b.astype(int)).max()))
        n += 1
    assert n >= 390


def test_jpeg_unsupported_and_truncated_streams_fail_cleanly(ours, tmp_path):
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rs = np.random.RandomState(3)
    img = Image.fromarray(rs.randint(0, 256, (24, 40, 3)).astype(np.uint8))
    img.save(str(tmp_path / "ok.jpg"), "JPEG", quality=80)
    data = (tmp_path / "ok.jpg").read_bytes()
    for cut in (2, 20, len(data) // 2, len(data) - 2):     # truncated at the header, in the tables, mid-scan, before EOI
        (tmp_path / "cut.jpg").write_bytes(data[:cut])
        st, _ = _load(ours, tmp_path / "cut.jpg")
        assert st == 4, cut
    img.convert("CMYK").save(str(tmp_path / "cmyk.jpg"), "JPEG")   # four components: refused, like stb_image 2.08
    st, _ = _load(ours, tmp_path / "cmyk.jpg")
    assert st == 4
