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
