cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bn20; rm -rf $O; mkdir -p $O
cat > /tmp/bn20.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from bcnn_amd import _lib, ops
L = _lib.load()
n, c, hw = 128, 64, 112
x = torch.rand((n, c, hw, hw), device="cuda") * 2 - 1
y = torch.empty_like(x); ws = torch.empty_like(x)
Z = lambda v=0.0: torch.full((c,), v, device="cuda")
rm, rv, sc, b, sm, sv = Z(), Z(1.0), Z(1.0), Z(), Z(), Z()
for _ in range(20):
    ops.batchnorm_forward(x, y, rm, rv, sc, b, sm, sv, ws, 1)
L.bcnn_hip_sync()
PY
rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 /tmp/bn20.py > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/k/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "StatsF" in r["Kernel_Name"] or "BnApply" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows if "StatsF" in r["Kernel_Name"]])
print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows if "BnApply" in r["Kernel_Name"]])
PY
