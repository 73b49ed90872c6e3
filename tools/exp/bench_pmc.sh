#!/bin/bash
# HBM traffic of every kernel of the bench.py ResNet-18 step: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# separate passes (MI355X_MICROARCH.md: they do not fit one pass), summed per kernel name and divided by launches.
# Writes gpurun_out/${ROUND}_<workload>_pmc.json (copy to profiles/). usage: [ROUND=r02] bench_pmc.sh [workload]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bpmc; rm -rf $O; mkdir -p $O
WL=${1:-resnet18}
ROUND=${ROUND:-r05}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads > $O/$C.log 2>&1
  tail -1 $O/$C.log | cut -c1-160
done
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
import bench
# Every dispatch of every pass, in dispatch order. A step ends with the one-launch SGD kernel; inside a step the backward
# pass starts at the first kernel only backward launches (cost / softmax gradient aside, the first weight-gradient,
# batch-norm-backward, pooling-backward or depthwise-backward kernel). The same LDS-DMA GEMM and the same fused Winograd
# kernel serve forward and the data gradient: the phase, not the name, decides which class a launch belongs to.
BWD_MARK = ("conv_dw", "wino_dw", "wino43_dw", "BnBwd", "bn_bwd", "BwdSums", "_bwd_kernel", "maxpool_bwd", "avgpool_bwd", "dwm_bwd", "dwl_bwd", "dw3_bwd", "eltwise_bwd", "cost_bwd", "softmax_bwd", "ActBwd")
# Per pass (= per counter: FETCH_SIZE and WRITE_SIZE come from separate runs, whose number of steps may differ now that
# bench.py stretches its warm-up by wall time) the sums are divided by THAT pass's number of steps. A step is counted where
# the backward phase begins (every step has exactly one such place, also the workloads without an SGD launch: configs[1]);
# the phase goes back to "fwd" after the optimizer launch or, without one, at the next forward-only kernel.
FWD_MARK = ("conv_fwd_window", "conv_fwd_direct", "conv_fwd_stem")
per_step = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, phase) -> counter -> value per step
lps = collections.defaultdict(float)                                          # (kernel, phase) -> launches per step
steps_by_counter = {}
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    rows_f = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    phase, nsteps = "fwd", 0
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in rows_f:
        k = r["Kernel_Name"]
        if phase == "bwd" and any(t in k for t in FWD_MARK): phase = "fwd"
        if phase == "fwd" and any(t in k for t in BWD_MARK): phase, nsteps = "bwd", nsteps + 1
        key = (k, phase)
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
        if "sgd_chunks" in k or "adam_chunks" in k: phase = "fwd"
    nsteps = max(nsteps, 1)
    for key in acc:
        for c, v in acc[key].items():
            per_step[key][c] += v / nsteps
            steps_by_counter[c] = nsteps
            lps[key] = max(lps[key], cnt[key][c] / nsteps)
STEPS = steps_by_counter
KIB = 1024.0  # FETCH_SIZE / WRITE_SIZE are reported in KiB
rows = {}
for key in per_step:
    n = lps[key]
    f_raw = per_step[key].get("FETCH_SIZE", 0.0) * KIB
    w = per_step[key].get("WRITE_SIZE", 0.0) * KIB
    rows[key] = {"launches_per_step": n, "fetch_bytes_raw_per_launch": f_raw / n, "write_bytes_per_launch": w / n,
                 "hbm_bytes_per_launch": (2 * f_raw + w) / n, "hbm_bytes_per_step": 2 * f_raw + w}
def cls(pred, phase=None):
    ks = [key for key in rows if pred(key[0]) and (phase is None or key[1] == phase)]
    # launches of the class = its main kernels (one per layer call); helpers (packing, transforms, finalize) only add bytes.
    # The three-kernel Winograd dW of a layer is counted through its dy transform (its GEMM is a conv_dw_dma launch).
    main = [key for key in ks if "wino_dy_transform" in key[0] or not any(t in key[0] for t in ("finalize", "cache_prefetch", "pack_weights", "pack_kernel", "transform", "finish", "fixup", "accumulate", "bn_stats", "bn_bwd_finalize", "bnfold"))]
    launches = sum(rows[key]["launches_per_step"] for key in main)
    tot = sum(rows[key]["hbm_bytes_per_step"] for key in ks)
    return {"kernels": sorted(set(key[0][:100] for key in ks)), "phase": phase or "any", "launches_per_step": launches, "hbm_bytes_per_step": tot,
            "hbm_bytes_per_launch": tot / launches if launches else None}
out = {
 "csrc_sha": bench.kernel_source_digest(),
 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --workload $WL --steps 3 --warmup 1 (tools/exp/bench_pmc.sh), $ROUND",
 "correction": "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE: gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section); the dword LDS-DMA gathers of the conv kernels are not separately calibrated, WRITE_SIZE as reported; both counters are in KiB",
 "attribution": "launches are split into forward / backward by dispatch order inside each step (backward starts at the first backward-only kernel, a step ends with the SGD launch): kernels that serve both passes are no longer counted twice",
 "steps": STEPS,
 "classes": {
   "conv_dw": cls(lambda k: ("conv_dw" in k) and "wino" not in k),
   "conv_fwd": cls(lambda k: "conv_fwd_direct" in k or "conv_fwd_window" in k or "conv_fwd_stem" in k or "cache_prefetch" in k or "conv_igemm" in k or "conv_pack_weights" in k or "dma_pack" in k, "fwd"),
   "conv_dx": cls(lambda k: "conv_igemm" in k or "conv_pack_weights" in k or "dma_pack" in k or "conv_dx" in k, "bwd"),
   "conv_fwd_winograd": cls(lambda k: "wino_fused_kernel" in k or "wino_tail_fixup" in k or "wino_pack_weights" in k or "wino_pack" in k, "fwd"),
   "conv_dx_winograd": cls(lambda k: "wino_fused_kernel" in k or "wino_tail_fixup" in k or "wino_pack_weights" in k or "wino_pack" in k, "bwd"),
   "conv_fwd_winograd43": cls(lambda k: "wino43b" in k, "fwd"),
   "conv_dx_winograd43": cls(lambda k: "wino43b" in k, "bwd"),
   "conv_dw_winograd43": cls(lambda k: "wino43_dw" in k),
   "conv_dw_winograd": cls(lambda k: "wino_dw" in k or "wino_input_transform" in k or "wino_dy_transform" in k),
   "bn_fwd": cls(lambda k: "BnApplyBody" in k or "bn_stats" in k or "StatsF" in k or "bn_fwd" in k, "fwd"),
   "bn_bwd": cls(lambda k: "BnBwd" in k or "bn_bwd" in k or "BwdSumsF" in k),
   "depthwise_fwd": cls(lambda k: "dw3_fwd" in k or "dw_fwd" in k or "dwl_fwd" in k or "dwm_fwd" in k),
   "depthwise_bwd": cls(lambda k: "dw3_bwd" in k or "dw_bwd" in k or "dwl_bwd" in k or "dwm_bwd" in k or "dwl_finalize" in k or "ActBwdSumF" in k or "dw_weight_accumulate" in k),
 },
 "kernels": {"%s [%s]" % (key[0][:110], key[1]): v for key, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:28]},
 "total_hbm_bytes_per_step": sum(v["hbm_bytes_per_step"] for v in rows.values()),
}
json.dump(out, open("$R/gpurun_out/${ROUND}_%s_pmc.json" % "$WL", "w"), indent=1)
for c, v in out["classes"].items():
    print(c, v["launches_per_step"], "launches/step", "%.1f MB/launch" % ((v["hbm_bytes_per_launch"] or 0) / 1e6), "%.2f GB/step" % (v["hbm_bytes_per_step"] / 1e9))
print("total %.2f GB/step" % (out["total_hbm_bytes_per_step"] / 1e9))
PY
