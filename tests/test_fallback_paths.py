"""The convolution has two generations of GEMM kernels: the LDS-DMA ones (conv_igemm_dma.hip, conv_dw_dma.hip)
take every shape they support, the register-staged ones (conv_igemm.hip, conv_bwd.hip) the rest. The product
library reads no environment; its EXPERIMENT build (bcnn_amd/lib/libbcnn_hip_exp.so: same sources compiled with
-DBCNN_HIP_EXPERIMENT, selected through BCNN_HIP_LIB) has the switch BCNN_HIP_NO_DMA=1 (read once per process) that
forces the second set, so the whole golden suite is replayed in a child process on that build to keep both
generations pinned to the reference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"BCNN_HIP_NO_DMA": "1"}, {"BCNN_HIP_NO_FUSED_STATS": "1"},
                                 {"BCNN_HIP_NO_SMALLC_DX": "1", "BCNN_HIP_NO_PREFETCH": "1"}],
                         ids=["register_staged_kernels", "unfused_bn_statistics", "implicit_gemm_dx_for_small_c"])
def test_golden_suite_on_the_other_code_path(env):
    e = dict(os.environ)
    e.update(env)
    e["BCNN_HIP_LIB"] = os.path.join(ROOT, "bcnn_amd", "lib", "libbcnn_hip_exp.so")
    assert os.path.exists(e["BCNN_HIP_LIB"]), "experiment build missing: __graft_entry__.build() makes it"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_parity.py"), "-m", "gpu",
                        "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
