"""The marching depthwise kernels keep their rows "requested PF steps ahead" only while the compiler can COUNT the requests
between a load and its use: one memory instruction under divergent control flow (or guarded by a run-time flag) and every wait
becomes `s_waitcnt vmcnt(0)` -- the kernels then run with no lead at all and nothing but the ISA shows it (DESIGN.md 4.0,
4.7g: a round and a half went by that way). This test compiles depthwise_march.hip to gfx950 assembly (no GPU needed) and
checks the inner loops of the product's overwrite-mode backward and the forward kernels: no full drain, counted waits present,
no memory instruction behind an exec-mask skip branch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bcnn_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa") / "depthwise_march.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
           "-Wno-inline-asm", "-S", "--cuda-device-only", "-o", str(out), "depthwise_march.hip"]
    r = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.read_text().split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            kernels[cur].append(line)
            if "s_endpgm" in line:
                cur = None
    return kernels


def _march_loop(body):
    """the lines of the first depth-1 loop (the march; the band-sum loops come after it): header label to its back edge"""
    hdr = next(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
    label = body[hdr].split(":")[0].strip()
    back = [i for i, l in enumerate(body) if i > hdr and re.search(r"\ts_c?branch\w*\s+" + re.escape(label) + r"\s*$", l)]
    assert back, "no back edge to " + label
    return body[hdr:back[-1] + 1]


def _check(loop, name):
    mem = [l for l in loop if re.search(r"\t(buffer|global)_(load|store)", l)]
    assert mem, name
    assert not any(re.search(r"\tglobal_(load|store)", l) for l in loop), name + ": a flat / global access in the march"
    assert not any("s_waitcnt vmcnt(0)" in l for l in loop), name + ": the march drains its requests"
    assert any(re.search(r"s_waitcnt vmcnt\([1-9]", l) for l in loop), name + ": no counted wait"
    # no memory instruction between an exec-mask skip branch and its target
    for i, l in enumerate(loop):
        m = re.search(r"s_cbranch_execz\s+(\.LBB\w+)", l)
        if not m:
            continue
        for k in loop[i + 1:]:
            if k.startswith(m.group(1) + ":"):
                break
            assert not re.search(r"\t(buffer|global)_(load|store)", k), name + ": memory instruction under a skip branch"


# template arguments as mangled: S, V, BN, BNIN, RELU, PF, OVW (backward) / S, V, BNIN, PF, RELU (forward)
BWD = [(1, 4, 1), (2, 4, 1), (1, 2, 4), (2, 2, 4), (1, 1, 4)]
FWD = [(1, 4, 2), (2, 4, 4), (1, 2, 4), (2, 2, 4), (1, 1, 4)]


@pytest.mark.parametrize("s,v,pf", BWD, ids=lambda x: str(x))
def test_backward_march_waits_are_counted(asm, s, v, pf):
    pat = "dwm_bwd_kernelILi%dELi%dELb1ELb1ELb1ELi%dELb1EEE" % (s, v, pf)   # both fusions, ReLU, overwrite: MobileNet's instance
    names = [k for k in asm if pat in k]
    assert len(names) == 1, (pat, names)
    _check(_march_loop(asm[names[0]]), names[0])


@pytest.mark.parametrize("s,v,pf", FWD, ids=lambda x: str(x))
def test_forward_march_waits_are_counted(asm, s, v, pf):
    pat = "dwm_fwd_kernelILi%dELi%dELb1ELi%dELb1EEE" % (s, v, pf)
    names = [k for k in asm if pat in k]
    assert len(names) == 1, (pat, names)
    _check(_march_loop(asm[names[0]]), names[0])
