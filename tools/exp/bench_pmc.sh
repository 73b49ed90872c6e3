#!/bin/bash
# HBM traffic of every kernel of the bench.py ResNet-18 step: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# separate passes (MI355X_MICROARCH.md: they do not fit one pass), summed per kernel name and divided by launches.
# Writes gpurun_out/${ROUND}_<workload>_pmc.json (copy to profiles/). usage: [ROUND=r02] bench_pmc.sh [workload]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bpmc; rm -rf $O; mkdir -p $O
WL=${1:-resnet18}
ROUND=${ROUND:-r03}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads > $O/$C.log 2>&1
  tail -1 $O/$C.log | cut -c1-160
done
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
import bench
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
STEPS = 4
KIB = 1024.0  # FETCH_SIZE / WRITE_SIZE are reported in KiB
rows = {}
for k in acc:
    n = max(cnt[k].values())
    f_raw = acc[k].get("FETCH_SIZE", 0.0) * KIB
    w = acc[k].get("WRITE_SIZE", 0.0) * KIB
    rows[k] = {"launches_per_step": n / STEPS, "fetch_bytes_raw_per_launch": f_raw / n, "write_bytes_per_launch": w / n,
               "hbm_bytes_per_launch": (2 * f_raw + w) / n, "hbm_bytes_per_step": (2 * f_raw + w) / STEPS}
def cls(pred):
    ks = [k for k in rows if pred(k)]
    # launches of the class = its main kernels (one per layer call); helpers (packing, transforms, finalize) only add bytes.
    # The three-kernel Winograd dW of a layer is counted through its dy transform (its GEMM is a conv_dw_dma launch).
    main = [k for k in ks if "wino_dy_transform" in k or not any(t in k for t in ("finalize", "cache_prefetch", "pack_weights", "transform", "finish", "accumulate", "bn_stats", "bn_bwd_finalize"))]
    launches = sum(rows[k]["launches_per_step"] for k in main)
    tot = sum(rows[k]["hbm_bytes_per_step"] for k in ks)
    return {"kernels": sorted(ks), "launches_per_step": launches, "hbm_bytes_per_step": tot,
            "hbm_bytes_per_launch": tot / launches if launches else None}
out = {
 "csrc_sha": bench.kernel_source_digest(),
 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --workload $WL --steps 3 --warmup 1 (tools/exp/bench_pmc.sh), $ROUND",
 "correction": "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE: gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section); the dword LDS-DMA gathers of the conv kernels are not separately calibrated, WRITE_SIZE as reported; both counters are in KiB",
 "classes": {
   # keys = bench.py's kernel classes. The fused Winograd kernel serves forward AND dX; the LDS-DMA GEMMs serve the
   # direct forward / dX layers and (as a 16-group GEMM) the three-kernel Winograd dW of the 14x14 / 7x7 stages.
   "conv_dw": cls(lambda k: ("conv_dw" in k) and "wino" not in k),
   "conv_fwd": cls(lambda k: "conv_fwd_direct" in k or "conv_fwd_window" in k or "cache_prefetch" in k or "conv_igemm" in k or "conv_pack_weights" in k),
   "conv_dx": cls(lambda k: "conv_igemm" in k or "conv_pack_weights" in k),
   "conv_fwd_winograd": cls(lambda k: "wino_fused_kernel" in k or "wino_pack_weights" in k),
   "conv_dx_winograd": cls(lambda k: "wino_fused_kernel" in k or "wino_pack_weights" in k),
   "conv_dw_winograd": cls(lambda k: "wino_dw" in k or "wino_input_transform" in k or "wino_dy_transform" in k),
   "bn_fwd": cls(lambda k: "BnApplyBody" in k or "bn_stats" in k or "StatsF" in k),
   "bn_bwd": cls(lambda k: "BnBwd" in k or "bn_bwd" in k or "BwdSumsF" in k),
   "depthwise_fwd": cls(lambda k: "dw3_fwd" in k or "dw_fwd" in k or "dwl_fwd" in k),
   "depthwise_bwd": cls(lambda k: "dw3_bwd" in k or "dw_bwd" in k or "dwl_bwd" in k or "dwl_finalize" in k or "ActBwdSumF" in k or "dw_weight_accumulate" in k),
 },
 "kernels": {k[:120]: v for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:24]},
 "total_hbm_bytes_per_step": sum(v["hbm_bytes_per_step"] for v in rows.values()),
}
json.dump(out, open("$R/gpurun_out/${ROUND}_%s_pmc.json" % "$WL", "w"), indent=1)
for c, v in out["classes"].items():
    print(c, v["launches_per_step"], "launches/step", "%.1f MB/launch" % ((v["hbm_bytes_per_launch"] or 0) / 1e6), "%.2f GB/step" % (v["hbm_bytes_per_step"] / 1e9))
print("total %.2f GB/step" % (out["total_hbm_bytes_per_step"] / 1e9))
PY
