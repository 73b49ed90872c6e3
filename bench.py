#!/usr/bin/env python3
"""bench.py -- images/s forward+backward of bcnn's conv hot path on MI355X, one process per GPU.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched under
torch.distributed.run, one rank per GPU, RCCL all-reduce of the weight-gradient arena after backward).
Rank 0 prints ONE JSON line: whole-job throughput, the roofline of the dominant kernel class (durations
measured live with HIP events on the launch stream, inside the timed region: every 4th timed step carries the
per-class event timers -- `profiled_steps` in the output -- because ~200 event records per step cost 4 % of a
ResNet step; BENCH_PROFILE_EVERY=1 times every step) and a CPU baseline (the unmodified reference,
oracle/_ref, timed on this host on a bounded sample; single-GPU runs only).
Data-parallel runs overlap the gradient all-reduce with backward (bcnn_amd/dp.py); --no-overlap disables it.

Workloads (BASELINE.json configs):
  resnet18  configs[2]/[3] (the configuration the metric is quoted on): ResNet-18, 224x224, N=128 per GPU,
            built through the bcnn_net C API (libbcnn.so): topology of the reference's
            examples/cifar10/cifar10_example.c:65-143 (fused-BN convs, eltwise-ReLU shortcuts, 1x1/s2
            projections) behind an ImageNet stem (7x7 s2 p3 + maxpool 3/2 SAME; the reference ships only
            the CIFAR stem), avgpool, fc-1000, softmax, cost. A step = bcnn_forward + bcnn_backward
            (+ all-reduce) + bcnn_update (SGD), i.e. one full training step on synthetic data.
  mobilenet configs[4]: MobileNet-v1 224x224, N=256 per GPU (depthwise 3x3 -> batch-norm -> 1x1 conv+BN+ReLU, x13).
  conv3x3   configs[1]: one 3x3 s1 p1 conv, N=128 x 3 x 224 x 224 -> 64; step = forward + backward(dW, dbias)
            straight through the C-ABI (the layer is the net input: no dX, bcnn_net.c:283).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA dense peak (v_mfma_f32_32x32x2_f32)


# ---------------------------------------------------------------------------------------------------
# graph
# ---------------------------------------------------------------------------------------------------
def build_resnet18(net, A, classes=1000, base=64):
    """A = module with ACT_* / PADDING_* constants (capi for the device net, ref_bind for the reference).
    base = stem width (64 = ResNet-18; tests/test_resnet18_parity.py also builds the half-width graph, the widest
    one the reference's in-tree gemm computes correctly -- DESIGN.md section 5, quirk 8)."""
    net.conv(base, 7, 2, 3, 1, 1, A.ACT_RELU, "input", "conv0")
    net.maxpool(3, 2, A.PADDING_SAME, "conv0", "pool0")
    src = "pool0"
    for stage, width in enumerate((base, 2 * base, 4 * base, 8 * base), start=1):
        for blk in (1, 2):
            down = stage > 1 and blk == 1
            a, b, out = "s%db%d_c1" % (stage, blk), "s%db%d_c2" % (stage, blk), "s%db%d" % (stage, blk)
            net.conv(width, 3, 2 if down else 1, 1, 1, 1, A.ACT_RELU, src, a)
            net.conv(width, 3, 1, 1, 1, 1, A.ACT_NONE, a, b)
            if down:  # 1x1 / s2 projection shortcut (raw-view addressing quirk of the reference included)
                proj = "s%db%d_proj" % (stage, blk)
                net.conv(width, 1, 2, 0, 1, 1, A.ACT_NONE, src, proj)
                net.eltwise(A.ACT_RELU, proj, b, out)
            else:
                net.eltwise(A.ACT_RELU, src, b, out)
            src = out
    net.avgpool(src, "gap")
    net.fullc(classes, A.ACT_NONE, "gap", "fc")
    net.softmax("fc", "prob")
    net.cost("prob", "label", "cost", 1.0)


def build_mobilenet_v1(net, A, classes=1000):
    """MobileNet-v1 (1.0, 224): conv3x3/s2 stem, 13 depthwise-separable blocks, global average pool, fc. In
    bcnn terms: the depthwise layer has no fused batch-norm, so each block is depthwise(+ReLU) -> stand-alone
    batch-norm -> 1x1 convolution with fused batch-norm + ReLU (BASELINE configs[4])."""
    net.conv(32, 3, 2, 1, 1, 1, A.ACT_RELU, "input", "conv0")
    src = "conv0"
    cfg = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2)] + [(512, 1)] * 5 + [(1024, 2), (1024, 1)]
    for i, (width, stride) in enumerate(cfg, start=1):
        dw, bn, pw = "b%d_dw" % i, "b%d_bn" % i, "b%d_pw" % i
        net.depthwise(3, stride, 1, A.ACT_RELU, src, dw)
        net.batchnorm(dw, bn)
        net.conv(width, 1, 1, 0, 1, 1, A.ACT_RELU, bn, pw)
        src = pw
    net.avgpool(src, "gap")
    net.fullc(classes, A.ACT_NONE, "gap", "fc")
    net.softmax("fc", "prob")
    net.cost("prob", "label", "cost", 1.0)


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the unmodified reference on a bounded sample of the same workload
# ---------------------------------------------------------------------------------------------------
def cpu_baseline(workload, sample_n):
    import numpy as np
    from oracle import ref_bind as rb
    if not rb.available():
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref/libbcnn_ref.so not present on this host"}
    rs = np.random.RandomState(0)
    best = None
    # the reference sizes its OpenMP team from the core count and oversubscribes badly on big hosts
    # (nested parallel-for in its gemm): try a few team sizes on one iteration each, time the best.
    ncpu = os.cpu_count() or 8
    for threads in sorted({8, 16, min(32, ncpu)}):
        if workload == "conv3x3":
            net = rb.RefNet(mode=rb.MODE_TRAIN, w=224, h=224, c=3, n=sample_n)
            net.conv(64, 3, 1, 1, 1, 0, rb.ACT_NONE, "input", "conv1")
        else:
            net = rb.RefNet(mode=rb.MODE_TRAIN, w=224, h=224, c=3, n=sample_n)
            (build_mobilenet_v1 if workload == "mobilenet" else build_resnet18)(net, rb)
        net.compile()
        net.L.ref_set_threads(net.net, threads)
        net.data(0)[...] = rs.uniform(-1, 1, net.shape(0)).astype(np.float32)
        t, _, _ = net.time_fwd_bwd(1, 2)
        net.close()
        if best is None or t < best[0]:
            best = (t, threads)
    t, threads = best
    return {"value": round(sample_n / t, 2), "unit": "images/s", "cores": int(threads), "kind": "reference",
            "sample": "unmodified reference (oracle/_ref, AVX2+OpenMP, in-tree gemm), same %s graph at N=%d, "
                      "bcnn_forward+bcnn_backward, best of 2 after 1 warm-up, best OpenMP team of {8,16,32}"
                      % (workload, sample_n)}


def read_profile(L):
    out = {}
    for cls in range(L.bcnn_hip_profile_num_classes()):
        ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        L.bcnn_hip_profile_read(cls, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
        if n.value:
            out[L.bcnn_hip_profile_class_name(cls).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value,
                                                                    bytes=by.value)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="resnet18", choices=["resnet18", "conv3x3", "mobilenet"])
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 128; 256 for mobilenet)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--input-grad", action="store_true",
                    help="conv3x3 only: give the source a gradient so that backward also runs dX (SURVEY.md 8d, config #2 variant)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="data-parallel runs: one blocking all-reduce after backward instead of overlapped buckets")
    args = ap.parse_args()

    # stdout carries exactly one line, the result: everything else that writes to file descriptor 1 while the job
    # runs (RCCL's version banner comes through C stdio, from whichever rank initialises first) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from bcnn_amd import _lib, capi, ops

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_dp_gpu.py): several ranks on ONE GPU over gloo, to exercise the data-parallel step
    # logic where only a single device exists. Never set by the driver.
    same_device = os.environ.get("BENCH_TEST_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    L = _lib.load()
    L.bcnn_hip_set_device(local_rank)
    if world > 1 or os.environ.get("BENCH_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if same_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:  # "nccl" IS RCCL on ROCm; device_id binds the communicator to this rank's GPU up front
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    n = args.batch if args.batch else (256 if args.workload == "mobilenet" else 128)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)   # every rank owns different images

    # launch on an explicit stream of our own; HIP events are recorded on that same stream
    stream = L.bcnn_hip_stream_create()
    L.bcnn_hip_set_stream(stream)

    if args.workload in ("resnet18", "mobilenet"):
        import numpy as np
        net = capi.Net(mode=capi.MODE_TRAIN, w=224, h=224, c=3, n=n)
        import random
        # identical parameters on every rank: the builders draw from libc rand()
        C.CDLL(None).srand(7)
        (build_resnet18 if args.workload == "resnet18" else build_mobilenet_v1)(net, capi)
        net.compile()
        net.set_sgd(0.01, 0.9, 5e-4)
        net.set_data_parallel(rank, world)
        x = torch.rand((n, 3, 224, 224), device=dev, generator=gen) * 2 - 1
        lab = torch.zeros((n, 1000), device=dev)
        lab[torch.arange(n, device=dev), torch.randint(0, 1000, (n,), device=dev, generator=gen)] = 1.0
        torch.cuda.synchronize()
        t_in, t_lab = net.tensor(0), net.tensor(1)
        L.bcnn_hip_memcpy_d2d(t_in.data_gpu, x.data_ptr(), x.numel() * 4)
        L.bcnn_hip_memcpy_d2d(t_lab.data_gpu, lab.data_ptr(), lab.numel() * 4)
        L.bcnn_hip_sync()
        gptr, gsize = net.gradient_arena()
        dp = world > 1 or os.environ.get("BENCH_FORCE_DP") == "1"   # the env switch exercises the DP plumbing on 1 GPU
        grads = torch.as_tensor(capi.DeviceArray(gptr, gsize), device=dev) if dp else None
        overlap = dp and not args.no_overlap and os.environ.get("BENCH_NO_OVERLAP") != "1"
        if overlap:
            # Gradient all-reduce overlapped with backward: the C executor reports growing tail ranges of the
            # gradient arena as their nodes finish (bcnn_set_gradient_ready_callback); ranges are gathered into
            # ~8 MB buckets and each bucket is all-reduced asynchronously on RCCL's stream, ordered after the
            # work queued so far on OUR stream (torch sees it as an ExternalStream). Nothing blocks the host;
            # update() is ordered behind the last bucket by work.wait() on the same stream.
            from bcnn_amd.dp import BucketedAllReduce
            ext = torch.cuda.ExternalStream(stream, device=dev)
            bar = BucketedAllReduce(grads, (8 << 20) // 4, lambda: torch.cuda.stream(ext))
            net.set_gradient_ready_callback(bar.on_ready)

        def step():
            net.forward()
            if overlap and bar.failed is None:
                bar.begin()
                net.backward()
                left = bar.finish()         # stream-side waits: update() below is ordered after every bucket
                if bar.failed is not None:
                    # finish this step correctly ([0, left) has not been reduced); later steps use the blocking path
                    print("bench.py: overlapped all-reduce failed (%r); falling back to the blocking path"
                          % (bar.failed,), file=sys.stderr, flush=True)
                    net.set_gradient_ready_callback(None)
                    L.bcnn_hip_sync()
                    torch.cuda.synchronize()
                    if left > 0:
                        dist.all_reduce(grads[:left])
                    torch.cuda.synchronize()
            else:
                net.backward()
                if dp:
                    L.bcnn_hip_sync()           # gradients complete on our stream before RCCL reads them
                    dist.all_reduce(grads)      # ONE all-reduce of the flat weight/bias-gradient arena (xGMI)
                    torch.cuda.synchronize()
            net.update()
        if args.workload == "resnet18":
            desc = ("ResNet-18 224x224 (BASELINE configs[2]), N=%d per GPU, fwd+bwd+SGD through bcnn_net C API; "
                    "CIFAR-example topology + ImageNet stem, fc-1000" % n)
        else:
            desc = ("MobileNet-v1 224x224 (BASELINE configs[4]), N=%d per GPU, fwd+bwd+SGD through bcnn_net C API; "
                    "depthwise 3x3 -> batch-norm -> 1x1 conv(+BN+ReLU) x13, fc-1000" % n)
        sample_n = 2
    else:
        c, h, w, f, k, s, p = 3, 224, 224, 64, 3, 1, 1
        oh, ow = ops.conv_out_hw(h, w, k, s, p)
        x = torch.rand((n, c, h, w), device=dev, generator=gen) * 2 - 1
        wgen = torch.Generator(device=dev).manual_seed(7)            # identical weights on every rank
        a = (3.0 / (c * k * k)) ** 0.5
        params = torch.empty(f * c * k * k + f, device=dev)
        grads = torch.zeros_like(params)
        wt = params[: f * c * k * k].view(f, c, k, k)
        bias = params[f * c * k * k:]
        wt.copy_((torch.rand(wt.shape, device=dev, generator=wgen) * 2 - 1) * a)
        bias.copy_((torch.rand(f, device=dev, generator=wgen) - 0.5) * 0.2)
        dw = grads[: f * c * k * k].view(f, c, k, k)
        db = grads[f * c * k * k:]
        y = torch.empty((n, f, oh, ow), device=dev)
        dy = (torch.rand((n, f, oh, ow), device=dev, generator=gen) * 2 - 1) * 1e-2
        ws = torch.zeros(max(1, ops.conv_workspace_size(n, c, h, w, f, k, s, p, 1)), device=dev)
        dxg = torch.empty_like(x) if args.input_grad else None
        torch.cuda.synchronize()

        def step():
            ops.conv_forward(x, wt, bias, y, k, s, p, 1, 0)
            ops.conv_backward(x, wt, y, dy, dxg, dw, db, k, s, p, 1, 0, ws)
            if world > 1:
                L.bcnn_hip_sync()
                dist.all_reduce(grads)
                torch.cuda.synchronize()
        desc = ("conv3x3 s1 p1, N=%d x 3 x 224 x 224 -> 64 (BASELINE configs[1]), fwd + bwd(dW, dbias%s)"
                % (n, ", dX: variant with a gradient-carrying source" if args.input_grad else
                   "); no dX: the layer's source is the net input"))
        sample_n = 16

    for _ in range(args.warmup):
        step()
    L.bcnn_hip_sync()
    torch.cuda.synchronize()
    L.bcnn_hip_profile_reset()
    # The per-class HIP-event timers cost ~3.5 us of GPU pipeline per event (~200 events per ResNet step = 4 % of
    # the step), so inside the timed region they sample every PROFILE_EVERY-th step instead of all of them; the
    # averages they give are per profiled launch, the throughput is over all steps.
    profile_every = max(1, int(os.environ.get("BENCH_PROFILE_EVERY", "4")))
    profiled_steps = 0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        prof_on = (it % profile_every) == 0
        L.bcnn_hip_profile_enable(1 if prof_on else 0)
        profiled_steps += 1 if prof_on else 0
        step()
    L.bcnn_hip_sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    L.bcnn_hip_profile_enable(0)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        prof = read_profile(L)
        # dominant kernel class by time inside the timed region
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else None
        roof = None
        if dom:
            name, d = dom
            avg_ms = d["ms"] / d["launches"]
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            ai = d["flops"] / d["bytes"] if d["bytes"] else 0.0
            ridge = MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
            if ai >= ridge:   # MFMA-bound class
                roof = {"kernel": name, "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF,
                        "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4)}
            else:
                roof = {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
            # HBM bytes per launch of that class from the PMC passes (rocprofv3 cannot run inside this process):
            # the committed summaries of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over this same
            # command (tools/exp/bench_pmc.sh), corrected as MI355X_MICROARCH.md prescribes
            traffic = None
            if args.workload in ("conv3x3", "resnet18") and n == 128 and not args.input_grad:
                pmc = os.path.join(ROOT, "profiles", "r01_%s_pmc.json" % args.workload)
                if os.path.exists(pmc):
                    traffic = (json.load(open(pmc)).get("classes", {}).get(name) or {}).get("hbm_bytes_per_launch")
            roof.update({"traffic": traffic, "launches": d["launches"], "avg_ms": round(avg_ms, 4),
                         "algorithmic_flops_per_launch": d["flops"] / d["launches"],
                         "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                         "tflops": round(tf, 2), "gbs": round(gbs, 1)})
        out = {
            "metric": "images/sec fwd+bwd", "value": round(args.steps * n * world / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "batch_per_gpu": n, "global_batch": n * world, "parallelism": "dp%d" % world},
            "roofline": roof,
            "profiled_steps": profiled_steps,
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / profiled_steps, 4), "launches_per_step": v["launches"] // profiled_steps,
                                   "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                                   "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in prof.items()},
        }
        if not args.no_cpu_baseline and world == 1:  # reported by the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.workload, sample_n)
        if os.environ.get("BENCH_TEST_CHECKSUM") == "1" and args.workload != "conv3x3":
            pptr, psize = net.parameter_arena()
            params = torch.as_tensor(capi.DeviceArray(pptr, psize), device=dev)
            out["param_checksum"] = [float(params.double().sum()), float(params.double().abs().sum())]
        line = json.dumps(out)
    else:
        line = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)   # C stdio is block-buffered on a pipe: push the banner out while fd 1 still is stderr
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
