"""Memory safety of libbip's decoders, which take files from outside: tools/fuzz_bip.c + the decoder sources compiled with
AddressSanitizer and UndefinedBehaviorSanitizer (CPU build; the GPU pool has no sanitizer runs), ~10 000 mutated JPEG / PNG
streams per run. The first run of this fuzzer found an out-of-bounds table write for a Huffman table with more codes than
its length can hold (bip_jpeg.c huff_build) and the left shifts of negative intermediates in the inverse DCT."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "bcnn_amd", "host")


def test_mutated_image_files_never_trip_the_sanitizers(tmp_path):
    pytest.importorskip("PIL")
    from PIL import Image
    exe = str(tmp_path / "fuzz")
    cmd = ["gcc", "-std=gnu99", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "fuzz_bip.c")] + \
          [os.path.join(HOST, f) for f in ("bip_decode.c", "bip_jpeg.c", "bip_min.c", "bip_augment.c")] + ["-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    rs = np.random.RandomState(1)
    seeds = []
    for i, (w, h) in enumerate(((33, 17), (64, 48), (8, 8))):
        img = Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8))
        for mode in ("RGB", "L"):
            for prog in (False, True):
                for sub in ((0, 2, "4:1:1") if mode == "RGB" else (0,)):
                    kw = dict(quality=60, progressive=prog)
                    if mode == "RGB":
                        kw["subsampling"] = sub
                    p = str(tmp_path / ("s%d_%s_%d_%s.jpg" % (i, mode, prog, str(sub).replace(":", ""))))
                    img.convert(mode).save(p, "JPEG", **kw)
                    seeds.append(p)
        p = str(tmp_path / ("r%d.jpg" % i))
        try:
            img.save(p, "JPEG", quality=70, restart_marker_blocks=2)
            seeds.append(p)
        except TypeError:
            pass
        p = str(tmp_path / ("p%d.png" % i))
        img.save(p)
        seeds.append(p)
    # hostile seeds no encoder writes: sampling factors that do not divide the largest one (3x1 under 4x1; the
    # upsampler's integer ratio truncates and a plane row ends before the image row does -- ADVICE round 3), on a
    # tall narrow frame where the last row sits at the end of the plane's allocation
    for j, (w, h) in enumerate(((1024, 8), (16, 200))):
        p = str(tmp_path / ("odd%d.jpg" % j))
        Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(p, "JPEG", quality=60, subsampling=0)
        raw = bytearray(open(p, "rb").read())
        sof = raw.find(b"\xff\xc0")
        assert sof > 0 and raw[sof + 9] == 3
        for k, hv in enumerate((0x31, 0x41, 0x41) if j == 0 else (0x13, 0x14, 0x12)):
            raw[sof + 10 + 3 * k + 1] = hv
        open(p, "wb").write(bytes(raw))
        seeds.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + seeds, capture_output=True, text=True, env=env, timeout=600)
    tail = "\n".join(ln for ln in (r.stdout + r.stderr).splitlines() if not ln.startswith("[ERROR]"))[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail and "LeakSanitizer" not in tail, tail
    assert "mutated files" in r.stdout
