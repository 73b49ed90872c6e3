"""The CMake entry point (CMakeLists.txt, option USE_HIP mirroring the reference's USE_CUDA, reference
CMakeLists.txt:5-10, 88-120, 181-236): configure + build in a scratch directory, all three libraries come out and
export the public API; where the reference tree is mounted its bcnn-cl and examples are built from their own
unmodified sources against them."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(shutil.which("cmake") is None or not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="cmake / hipcc not available")
def test_cmake_use_hip_builds_the_libraries(tmp_path):
    build = str(tmp_path / "build")
    cfg = ["cmake", "-S", ROOT, "-B", build, "-DUSE_HIP=ON"]
    if shutil.which("ninja"):
        cfg += ["-G", "Ninja"]
    have_ref = os.path.isfile(os.path.join(REF, "src", "cli", "bcnn_cl.c"))
    if have_ref:
        cfg.append("-DBCNN_REFERENCE_DIR=" + REF)
    r = subprocess.run(cfg, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Build with HIP" in r.stdout
    r = subprocess.run(["cmake", "--build", build, "-j8"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    for lib in ("libbcnn_hip.so", "libbcnn.so", "libbip.so"):
        assert os.path.exists(os.path.join(build, "lib", lib)), lib
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(build, "lib", "libbcnn.so")],
                          capture_output=True, text=True).stdout
    for name in ("bcnn_init_net", "bcnn_add_convolutional_layer", "bcnn_train_on_batch", "bcnn_load_net"):
        assert " T " + name in syms, name
    if have_ref:
        for exe in ("bcnn-cl", "mnist-example", "cifar10-example", "inference-benchmark"):
            assert os.path.exists(os.path.join(build, "bin", exe)), exe
    # USE_HIP=OFF is refused with a message instead of silently building nothing
    r = subprocess.run(["cmake", "-S", ROOT, "-B", str(tmp_path / "b2"), "-DUSE_HIP=OFF"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)
