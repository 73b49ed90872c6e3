"""The small link-closure pieces that unchanged consumers of the reference pull in: libbip.so's PNG writer
(include/bip/bip.h) and the bh/*.h helper headers (exercised through a C program built with gcc)."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bcnn_amd", "lib")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "bcnn_amd", "host"), "../lib/libbip.so"], check=True,
                   capture_output=True)


def _decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(tag + body) & 0xffffffff == crc, tag
        chunks.append((tag, body))
        pos += 12 + n
    w, h, bits, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    raw = zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT"))
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    rows = np.frombuffer(raw, np.uint8).reshape(h, w * ch + 1)
    assert bits == 8 and (rows[:, 0] == 0).all() and chunks[-1][0] == b"IEND"
    return rows[:, 1:].reshape(h, w, ch)


def test_bip_write_image_produces_a_valid_png(tmp_path):
    _build()
    bip = C.CDLL(os.path.join(LIB, "libbip.so"))
    bip.bip_write_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    bip.bip_write_image.restype = C.c_int
    rs = np.random.RandomState(0)
    for (h, w, ch, pad) in ((5, 7, 3, 0), (300, 260, 3, 4), (9, 4, 1, 3), (3, 3, 4, 0)):   # 300x260x3 > one 64 KiB block
        img = rs.randint(0, 256, (h, w * ch + pad)).astype(np.uint8)
        p = str(tmp_path / ("t%d.png" % h)).encode()
        assert bip.bip_write_image(p, img.ctypes.data, w, h, ch, w * ch + pad) == 0
        got = _decode_png(p.decode())
        np.testing.assert_array_equal(got.reshape(h, w * ch), img[:, :w * ch])
    assert bip.bip_write_image(b"/nonexistent_dir/x.png", img.ctypes.data, 3, 3, 4, 12) != 0
    assert bip.bip_write_image(str(tmp_path / "bad.png").encode(), img.ctypes.data, 3, 3, 2, 6) != 0


C_PROG = r"""
#include <bh/bh_ini.h>
#include <bh/bh_log.h>
#include <bh/bh_mem.h>
#include <bh/bh_string.h>
#include <bh/bh_timer.h>
int main(int argc, char **argv) {
    bh_ini_parser *cfg = bh_ini_parser_create(argv[1]);
    if (!cfg) return 2;
    printf("%d\n", cfg->num_sections);
    for (int i = 0; i < cfg->num_sections; ++i) {
        printf("%s %d\n", cfg->sections[i].name, cfg->sections[i].num_keys);
        for (int j = 0; j < cfg->sections[i].num_keys; ++j)
            printf("  %s|%s\n", cfg->sections[i].keys[j].name, cfg->sections[i].keys[j].val);
    }
    bh_ini_parser_destroy(cfg);
    char **tok = NULL;
    char s[] = ",a,,bc,d";
    int n = bh_strsplit(s, ',', &tok);
    printf("%d %s %s %s\n", n, tok[0], tok[1], tok[2]);
    for (int i = 0; i < n; ++i) bh_free(tok[i]);
    bh_free(tok);
    char *d = NULL;
    bh_strfill(&d, "x y");
    bh_strstrip(d);
    printf("%s %d\n", d, tok == NULL);
    bh_free(d);
    bh_timer t = {0};
    bh_timer_start(&t); bh_timer_stop(&t);
    bh_log(BH_LOG_SILENT, "never printed\n");
    return bh_timer_get_msec(&t) >= 0.0 ? 0 : 1;
}
"""


def test_bh_headers_behave_like_the_reference(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(C_PROG)
    cfg = tmp_path / "c.conf"
    cfg.write_text("# comment\n[network]\nsource_train = ./a b.idx\n batch_size=16\n\n; note\n[conv]\nfilters = 8\n! bang\nsrc=input\n")
    exe = str(tmp_path / "t")
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe, str(cfg)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split("\n")[:8] == ["2", "[network] 2", "  source_train|./ab.idx", "  batch_size|16", "[conv] 2",
                                         "  filters|8", "  src|input", "3 a bc d"]
    assert out.stdout.split("\n")[8] == "xy 1"
    bad = tmp_path / "bad.conf"
    bad.write_text("[net]\nthis line has no equals sign\n")
    assert subprocess.run([exe, str(bad)], capture_output=True).returncode == 2
