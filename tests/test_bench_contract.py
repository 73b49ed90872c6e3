"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and
`cpu_baseline` objects; checked here on a short run so that a broken bench shows up in the GPU suite."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["resnet18", "conv3x3", "mobilenet"])
def test_bench_prints_the_contract_line(workload):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--workload", workload], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert roof["peak"] in (8000.0, 157.3) and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert "traffic" in roof and "traffic_source" in roof   # null unless a PMC summary of THESE kernel sources is on file
    if workload == "resnet18":   # the default run also times configs[1] and configs[4] briefly
        side = d["workloads"]
        assert set(side) == {"conv3x3", "mobilenet"}
        for w in side.values():
            assert w["images_per_s"] > 0 and w["roofline"]["bound"] in ("hbm", "mfma") and w["kernel_classes"]
            assert w["cpu_baseline"]["kind"] == "reference" and w["cpu_baseline"]["value"] > 0  # per workload
    else:
        assert "workloads" not in d
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert cpu["host_cores"] == os.cpu_count()
    if "winograd" in roof["kernel"]:
        assert 0 < roof["frac_unpadded"] <= roof["frac"]
