"""Source compatibility of the drop-in boundary: the reference's OWN example programs (read from
/root/reference, never copied) must compile and link unchanged against include/bcnn/bcnn.h and
libbcnn.so. Runs only where the reference tree is mounted (the build container)."""
import os
import subprocess

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples")), reason="reference tree not mounted")
@pytest.mark.parametrize("src", ["examples/mnist/mnist_example.c", "examples/cifar10/cifar10_example.c"])
def test_reference_example_links_unchanged(tmp_path, src):
    from bcnn_amd import capi
    capi.build()
    exe = str(tmp_path / "example")
    cmd = ["gcc", "-std=gnu99", "-O1", "-DBCNN_USE_HIP", "-I", os.path.join(ROOT, "include"),
           os.path.join(REF, src), "-o", exe, "-L", os.path.join(ROOT, "bcnn_amd", "lib"), "-lbcnn", "-lbcnn_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "bcnn_amd", "lib"), "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    assert os.path.exists(exe)


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src", "cli", "bcnn_cl.c")), reason="reference tree not mounted")
def test_bcnn_cl_links_unchanged(tmp_path):
    """The reference's command-line tool (src/cli/bcnn_cl.c, the `bcnn-cl` target of its CMakeLists.txt:219-228)
    compiles and links unchanged: public API from include/bcnn, the internal header names it includes
    (bcnn_tensor.h, bcnn_utils.h, bcnn_yolo.h) from bcnn_amd/host, bh/*.h and bip/bip.h from include/,
    libbcnn.so + libbip.so. Its own header bcnn_cl.h is taken from the reference's src/cli."""
    from bcnn_amd import capi
    capi.build()
    exe = str(tmp_path / "bcnn-cl")
    lib = os.path.join(ROOT, "bcnn_amd", "lib")
    cmd = ["gcc", "-std=gnu99", "-O1", "-DBCNN_USE_HIP", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "bcnn_amd", "host"), "-I", os.path.join(REF, "src", "cli"),
           os.path.join(REF, "src", "cli", "bcnn_cl.c"), "-o", exe, "-L", lib, "-lbcnn", "-lbip", "-lbcnn_hip",
           "-Wl,-rpath," + lib, "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    # no GPU needed for the usage path: it returns before touching the library
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode != 0 and "Usage" in run.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples", "inference_benchmark")), reason="reference tree not mounted")
def test_inference_benchmark_links_unchanged(tmp_path):
    """examples/inference_benchmark/inference_benchmark.c (the `inference-benchmark` target):
    bcnn_load_net + bcnn_fill_tensor_with_image + bcnn_forward, and from libbip.so bip_load_image (:54) and
    bip_resize_bilinear (:72)."""
    from bcnn_amd import capi
    capi.build()
    exe = str(tmp_path / "inference-benchmark")
    lib = os.path.join(ROOT, "bcnn_amd", "lib")
    cmd = ["gcc", "-std=gnu99", "-O1", "-DBCNN_USE_HIP", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "bcnn_amd", "host"),
           os.path.join(REF, "examples", "inference_benchmark", "inference_benchmark.c"), "-o", exe, "-L", lib,
           "-lbcnn", "-lbip", "-lbcnn_hip", "-Wl,-rpath," + lib, "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True)   # usage path: returns before touching the device
    assert run.returncode != 0 and "Usage" in run.stderr
