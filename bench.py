#!/usr/bin/env python3
"""bench.py -- images/s forward+backward of bcnn's conv hot path on MI355X, one process per GPU.

Contract: `python bench.py --gpus N --steps K --warmup W`: one rank per GPU, RCCL all-reduce of the weight-gradient
arena after backward. For N > 1 either a launcher starts the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: WORLD_SIZE is set and must equal N) or, run bare, bench.py starts that launcher itself as a
child process before anything touches the GPU and relays rank 0's line (launch_ranks below); fewer than N visible
GPUs is an error, never a smaller run under the asked-for label.
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
        return {"value": None, "unit": "images/s", "cores": 0, "host_cores": os.cpu_count(), "kind": "reference",
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
    out = {"value": round(sample_n / t, 2), "unit": "images/s", "cores": int(threads), "host_cores": ncpu, "kind": "reference",
           "sample": "unmodified reference (oracle/_ref, AVX2+OpenMP, in-tree gemm), same %s graph at N=%d, "
                     "bcnn_forward+bcnn_backward, best of 2 after 1 warm-up, best OpenMP team of {8,16,32} "
                     "(`cores` = that team, `host_cores` = os.cpu_count())"
                     % (workload, sample_n)}
    if workload != "conv3x3" and sample_n < 8:
        # the reference runs its GEMMs per image, so the batch does not change their shapes -- but at N=2 the fixed costs of a
        # pass (OpenMP team start-up per gemm, batch-norm over two images) weigh more: the same graph at N=8 on the team
        # chosen above is the headline CPU figure, the N=2 sample stays beside it
        net = rb.RefNet(mode=rb.MODE_TRAIN, w=224, h=224, c=3, n=8)
        (build_mobilenet_v1 if workload == "mobilenet" else build_resnet18)(net, rb)
        net.compile()
        net.L.ref_set_threads(net.net, threads)
        net.data(0)[...] = rs.uniform(-1, 1, net.shape(0)).astype(np.float32)
        t8, _, _ = net.time_fwd_bwd(1, 2)
        net.close()
        out["also"] = {"n": sample_n, "value": out["value"]}
        out["value"] = round(8 / t8, 2)
        out["sample"] = out["sample"].replace("at N=%d" % sample_n, "at N=8 (team chosen on N=%d, which gave %.2f images/s)"
                                              % (sample_n, out["also"]["value"]))
    return out


# (tests that compare parameter checksums between two runs need the same number of steps in both: no stretching there)
MIN_WARM_S = 0.0 if os.environ.get("BENCH_TEST_CHECKSUM") == "1" else float(os.environ.get("BENCH_MIN_WARM_S", "0.08"))


def read_profile(L):
    out = {}
    for cls in range(L.bcnn_hip_profile_num_classes()):
        ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        L.bcnn_hip_profile_read(cls, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
        if n.value:
            out[L.bcnn_hip_profile_class_name(cls).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value,
                                                                    bytes=by.value,
                                                                    useful_flops=L.bcnn_hip_profile_read_useful_flops(cls))
    return out


def kernel_source_digest():
    """sha256 over the kernel sources: a PMC summary under profiles/ only speaks for the build it was taken on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "bcnn_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "bcnn_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, cls_name):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    this same command (tools/exp/bench_pmc.sh; rocprofv3 cannot run inside this process). Only reported when the
    summary was taken on THESE kernel sources (its `csrc_sha` field); otherwise null plus the reason."""
    import glob
    digest = kernel_source_digest()
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.json" % workload)), reverse=True)
    for path in cands:
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("csrc_sha") != digest:
            continue
        v = (d.get("classes", {}).get(cls_name) or {}).get("hbm_bytes_per_launch")
        if v is not None:
            return v, "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command; 2 x FETCH + WRITE; " \
                      "kernel sources %s)" % (os.path.relpath(path, ROOT), digest)
    why = "no PMC summary under profiles/ was taken on these kernel sources (sha %s)" % digest
    if cands:
        why += "; newest on file: %s" % os.path.relpath(cands[0], ROOT)
    return None, why


def class_table(prof, profiled_steps):
    out = {}
    for k, v in prof.items():
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else None
        gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        out[k] = {"ms_per_step": round(v["ms"] / profiled_steps, 4), "launches_per_step": v["launches"] // profiled_steps,
                  "tflops": round(tf, 2) if tf else None, "gbs": round(gbs, 1),
                  "frac_mfma": round(tf / MFMA_F32_PEAK_TF, 4) if tf else None,
                  "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def roofline_of(prof, workload, traffic_ok, pick=None):
    """roofline of the class that takes the most time inside the timed region (pick: of that class instead)"""
    if not prof:
        return None
    # Each Winograd kernel serves the forward AND the dX class of its layers (wino_fused_kernel: F(2x2,3x3), classes
    # *_winograd; wino43b_kernel: F(4x4,3x3), classes *_winograd43): a dominant KERNEL is judged on both of its classes together
    # (the per-class figures stay in `kernel_classes`)
    prof = dict(prof)
    for suffix in ("winograd", "winograd43"):
        fw, dx = prof.pop("conv_fwd_" + suffix, None), prof.pop("conv_dx_" + suffix, None)
        if fw or dx:
            parts = [p for p in (fw, dx) if p]
            prof["conv_fwd_dx_" + suffix] = {k: sum(p[k] for p in parts) for k in ("ms", "launches", "flops", "bytes", "useful_flops")}
    if pick is not None and pick not in prof:
        return None
    name, d = (pick, prof[pick]) if pick is not None else max(prof.items(), key=lambda kv: kv[1]["ms"])
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
    traffic, source = (pmc_traffic(workload, name.replace("conv_fwd_dx_", "conv_fwd_")) if traffic_ok
                       else (None, "non-default batch or variant: no PMC summary applies"))
    if "winograd" in name:
        # the class timers of the Winograd kernels carry the FLOPs the MFMAs really execute -- 16 instead of 36 multiplies
        # per 2 x 2 outputs for F(2x2,3x3), 36 instead of 144 per 4 x 4 outputs for F(4x4,3x3)
        # (section 4.9): the direct-convolution count of the same layers is 2.25x / 4x that
        factor = 4.0 if name.endswith("43") else 2.25
        roof["flops_counted"] = ("executed (Winograd transformed domain); direct-equivalent rate = %.2f x the rate of the "
                                 "multiplies that are not tile padding" % factor)
        useful_tf = d.get("useful_flops", d["flops"]) / (d["ms"] * 1e-3) / 1e12
        roof["direct_equivalent_tflops"] = round(factor * useful_tf, 2)  # algorithmic (direct) FLOPs of the layers / time
        # `frac` counts what the MFMAs execute, tile padding included (7 x 7 planes: 16 tiles cover 8 x 8);
        # frac_unpadded counts only the multiplies of tiles' cells that exist
        roof["frac_unpadded"] = round(d.get("useful_flops", d["flops"]) / (d["ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)
    roof.update({"traffic": traffic, "traffic_source": source, "launches": d["launches"], "avg_ms": round(avg_ms, 4),
                 "algorithmic_flops_per_launch": d["flops"] / d["launches"],
                 "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                 "tflops": round(tf, 2), "gbs": round(gbs, 1),
                 "frac_mfma": round(tf / MFMA_F32_PEAK_TF, 4), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)})
    return roof


def comm_id_path():
    """where rank 0 publishes the RCCL unique id for --comm inlib: one file per launch, the same on every rank"""
    tag = os.environ.get("TORCHELASTIC_RUN_ID", "none") + "_" + os.environ.get("MASTER_PORT", "0")
    os.environ.setdefault("BCNN_HIP_JOB_NONCE", tag)   # records of another launch at a reused path are ignored (csrc/comm.hip)
    return os.environ.get("BENCH_COMM_ID_PATH") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "bcnn_bench_comm_%s.id" % tag)


class Workload:
    """one synthetic workload resident in HBM: .step() runs one pass of the hot path over one batch"""

    def __init__(self, name, n, rank, world, dev, L, stream, input_grad=False, overlap=True, comm="torch"):
        import torch
        import torch.distributed as dist
        from bcnn_amd import capi, ops
        self.name, self.n, self.world, self.L, self.net = name, n, world, L, None
        self.dev, self.warmup_extra = dev, 0
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)   # every rank owns different images
        dp = world > 1 or os.environ.get("BENCH_FORCE_DP") == "1"     # the env switch exercises the DP plumbing on 1 GPU
        if name in ("resnet18", "mobilenet"):
            net = self.net = capi.Net(mode=capi.MODE_TRAIN, w=224, h=224, c=3, n=n)
            C.CDLL(None).srand(7)   # identical parameters on every rank: the builders draw from libc rand()
            (build_resnet18 if name == "resnet18" else build_mobilenet_v1)(net, capi)
            net.compile()
            net.set_sgd(0.01, 0.9, 5e-4)
            inlib = dp and comm == "inlib"
            if inlib:
                # the product's own collective (csrc/comm.hip): parameters broadcast from rank 0, ~8 MB tail buckets of the
                # gradient arena all-reduced from inside bcnn_backward on the communicator's stream, bcnn_update ordered
                # behind the last bucket -- nothing of torch.distributed touches a gradient in this mode
                net.set_data_parallel_comm(rank, world, comm_id_path())
            else:
                net.set_data_parallel(rank, world)
            x = torch.rand((n, 3, 224, 224), device=dev, generator=gen) * 2 - 1
            lab = torch.zeros((n, 1000), device=dev)
            lab[torch.arange(n, device=dev), torch.randint(0, 1000, (n,), device=dev, generator=gen)] = 1.0
            torch.cuda.synchronize()
            L.bcnn_hip_memcpy_d2d(net.tensor(0).data_gpu, x.data_ptr(), x.numel() * 4)
            L.bcnn_hip_memcpy_d2d(net.tensor(1).data_gpu, lab.data_ptr(), lab.numel() * 4)
            L.bcnn_hip_sync()
            del x, lab
            gptr, gsize = net.gradient_arena()
            grads = torch.as_tensor(capi.DeviceArray(gptr, gsize), device=dev) if dp and not inlib else None
            if dp and not inlib:   # every rank continues from rank 0's parameters, whatever its own initialisation drew
                pptr, psize = net.parameter_arena()
                dist.broadcast(torch.as_tensor(capi.DeviceArray(pptr, psize), device=dev), 0)
                torch.cuda.synchronize()
            bar = None
            if dp and overlap and not inlib:
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
                if bar is not None:
                    bar.begin()
                    net.backward()
                    bar.finish()            # stream-side waits: update() below is ordered after every bucket
                    if bar.failed is not None:
                        # A failure is only known to THIS rank: the others have already queued their buckets, so any
                        # local "fall back to one blocking all-reduce" would issue a different collective sequence
                        # and hang the job or reduce the wrong ranges. Fail the run instead (the launcher tears the
                        # other ranks down); --no-overlap selects the blocking path for every rank up front.
                        print("bench.py: overlapped all-reduce failed on rank %d: %r" % (rank, bar.failed),
                              file=sys.stderr, flush=True)
                        os._exit(3)
                else:
                    net.backward()                  # --comm inlib: the all-reduce happens in here
                    if dp and not inlib:
                        L.bcnn_hip_sync()           # gradients complete on our stream before RCCL reads them
                        dist.all_reduce(grads)      # ONE all-reduce of the flat weight/bias-gradient arena (xGMI)
                        torch.cuda.synchronize()
                net.update()
            self.step = step
            if name == "resnet18":
                self.desc = ("ResNet-18 224x224 (BASELINE configs[2]), N=%d per GPU, fwd+bwd+SGD through bcnn_net C API; "
                             "CIFAR-example topology + ImageNet stem, fc-1000" % n)
            else:
                self.desc = ("MobileNet-v1 224x224 (BASELINE configs[4]), N=%d per GPU, fwd+bwd+SGD through bcnn_net C API; "
                             "depthwise 3x3 -> batch-norm -> 1x1 conv(+BN+ReLU) x13, fc-1000" % n)
            self.sample_n = 2
            self.default_n = 256 if name == "mobilenet" else 128
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
            dxg = torch.empty_like(x) if input_grad else None
            torch.cuda.synchronize()
            self._keep = (x, params, grads, y, dy, ws, dxg)

            inlib = world > 1 and comm == "inlib"
            if inlib:
                L.bcnn_hip_comm_init(rank, world, comm_id_path().encode())

            def step():
                ops.conv_forward(x, wt, bias, y, k, s, p, 1, 0)
                ops.conv_backward(x, wt, y, dy, dxg, dw, db, k, s, p, 1, 0, ws)
                if inlib:
                    L.bcnn_hip_allreduce_sum(grads.data_ptr(), grads.numel())   # queued behind the stream's work so far
                    L.bcnn_hip_comm_join()                                      # the stream waits for it, the host does not
                elif world > 1:
                    L.bcnn_hip_sync()
                    dist.all_reduce(grads)
                    torch.cuda.synchronize()
            self.step = step
            self.desc = ("conv3x3 s1 p1, N=%d x 3 x 224 x 224 -> 64 (BASELINE configs[1]), fwd + bwd(dW, dbias%s)"
                         % (n, ", dX: variant with a gradient-carrying source" if input_grad else
                            "); no dX: the layer's source is the net input"))
            self.sample_n = 16
            self.default_n = 128

    def run(self, steps, warmup):
        """W untimed steps, then exactly K timed steps between barrier + synchronize; returns (seconds, per-class
        profile, profiled steps). The per-class HIP-event timers cost ~3.5 us of GPU pipeline per event (~200 events
        per ResNet step = 4 % of the step), so inside the timed region they sample every PROFILE_EVERY-th step
        instead of all of them; the averages they give are per profiled launch, the throughput is over all steps."""
        import torch
        import torch.distributed as dist
        L = self.L
        tw = time.perf_counter()
        for _ in range(warmup):
            self.step()
        L.bcnn_hip_sync()
        torch.cuda.synchronize()
        # Warm clocks: after an idle phase (building a workload is one) this part runs its first ~40 ms of load with the memory
        # side not yet at speed -- tools/exp/pair_gap.py: the configs[1] step takes 0.70 ms for its first 24 repetitions,
        # 0.60 for the next 24, 0.585 from there on, and again 0.70 after the host slept 2 s. W steps of a 0.6 ms workload
        # are over long before that, so the untimed phase is stretched to MIN_WARM_S of wall time by further UNTIMED steps
        # (their number is reported as `warmup_extra`); the timed region below is exactly `steps` steps either way.
        self.warmup_extra = 0
        if self.world > 1:
            # every rank has to run the same number of steps (each one holds collectives): agree on the slowest rank's clock
            spent = torch.tensor([time.perf_counter() - tw], dtype=torch.float64, device=self.dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(spent, op=dist.ReduceOp.MAX)
            spent = float(spent.item())
            per_step = spent / warmup if warmup > 0 else MIN_WARM_S / 4
            extra = int(max(0.0, MIN_WARM_S - spent) / max(per_step, 1e-6) + 0.999)
            for _ in range(min(extra, 1000)):
                self.step()
                self.warmup_extra += 1
            L.bcnn_hip_sync()
        else:
            while time.perf_counter() - tw < MIN_WARM_S:
                self.step()
                L.bcnn_hip_sync()
                self.warmup_extra += 1
        torch.cuda.synchronize()
        L.bcnn_hip_profile_reset()
        profile_every = max(1, int(os.environ.get("BENCH_PROFILE_EVERY", "4")))
        profiled = 0
        if self.world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for it in range(steps):
            prof_on = (it % profile_every) == 0
            L.bcnn_hip_profile_enable(1 if prof_on else 0)
            profiled += 1 if prof_on else 0
            self.step()
        L.bcnn_hip_sync()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        L.bcnn_hip_profile_enable(0)
        prof = read_profile(L)
        # `alone` leg, OUTSIDE the timed region: a few more steps with the weight gradients on the caller's stream
        # (bcnn_set_weight_gradient_stream(net, 0)), every launch timed -- what each kernel class takes when it has the chip to
        # itself. Inside the timed region the weight gradients run next to the sweeps and data gradients of the layers in
        # front, and a class's events span the time it shares the chip.
        self.prof_alone = None
        if self.net is not None and self.world == 1 and os.environ.get("BENCH_NO_ALONE_LEG") != "1":
            capi_lib = self.net.L
            capi_lib.bcnn_set_weight_gradient_stream(self.net.net, 0)
            self.step(); L.bcnn_hip_sync()
            L.bcnn_hip_profile_reset()
            L.bcnn_hip_profile_enable(1)
            for _ in range(3):
                self.step()
            L.bcnn_hip_sync()
            L.bcnn_hip_profile_enable(0)
            self.prof_alone = (read_profile(L), 3)
            capi_lib.bcnn_set_weight_gradient_stream(self.net.net, 1)
        return dt, prof, profiled

    def close(self):
        self.L.bcnn_hip_sync()
        if self.net is not None:
            self.net.set_gradient_ready_callback(None)
            self.net.close()
            self.net = None
        self._keep = None
        self.step = None


def visible_gpus():
    """number of GPUs a rank would see, counted by a short-lived child so that THIS process never initialises HIP"""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                       capture_output=True, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks (one per GPU, the reference's
    one-device-per-process model, /root/reference/src/cli/bcnn_cl.c:281-285) under torch.distributed.run as a CHILD
    process, relay rank 0's single JSON line and return the child's exit code. The parent imports neither torch nor
    the HIP library: a process that has initialised the GPU must not start or become another program."""
    import socket
    import subprocess
    if os.environ.get("BENCH_TEST_SAME_DEVICE") != "1":
        have = visible_gpus()
        if have < n:
            print("bench.py: --gpus %d asked for, %d visible on this node: refusing to print a %d-GPU number under "
                  "another label" % (n, have, have), file=sys.stderr, flush=True)
            return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    out, _ = child.communicate()
    result = None
    for ln in out.splitlines():
        if ln.startswith("{") and result is None:
            result = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if child.returncode == 0 and result is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr, flush=True)
        return 4
    if result is not None:
        print(result, flush=True)
    return child.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="resnet18", choices=["resnet18", "conv3x3", "mobilenet"])
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 128; 256 for mobilenet)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="single-GPU default run only: skip the short conv3x3 (configs[1]) and mobilenet (configs[4]) "
                         "timings reported under `workloads`")
    ap.add_argument("--input-grad", action="store_true",
                    help="conv3x3 only: give the source a gradient so that backward also runs dX (SURVEY.md 8d, config #2 variant)")
    ap.add_argument("--comm", default="torch", choices=["torch", "inlib"],
                    help="data-parallel runs: who runs the gradient all-reduce -- torch.distributed (RCCL through PyTorch, the "
                         "default) or the library's own communicator (csrc/comm.hip behind bcnn_set_data_parallel_comm: what "
                         "a plain C consumer of libbcnn.so uses, INTEGRATION.md C1)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="data-parallel runs: one blocking all-reduce after backward instead of overlapped buckets")
    args = ap.parse_args()
    if args.steps < 1:
        ap.error("--steps must be at least 1 (the timed region runs exactly that many steps)")
    if args.warmup < 0:
        ap.error("--warmup must not be negative")
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus))   # parent of N ranks: never touches the GPU itself
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))

    # stdout carries exactly one line, the result: everything else that writes to file descriptor 1 while the job
    # runs (RCCL's version banner comes through C stdio, from whichever rank initialises first) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from bcnn_amd import _lib, capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_dp_gpu.py): several ranks on ONE GPU over gloo, to exercise the data-parallel step
    # logic where only a single device exists. Never set by the driver.
    same_device = os.environ.get("BENCH_TEST_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d wants GPU %d, %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    L = _lib.load()
    L.bcnn_hip_set_device(local_rank)
    if world > 1 or os.environ.get("BENCH_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if same_device or args.comm == "inlib":
            # gloo: with --comm inlib torch.distributed only carries the barriers and the max over the ranks' clocks
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:  # "nccl" IS RCCL on ROCm; device_id binds the communicator to this rank's GPU up front
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    n = args.batch if args.batch else (256 if args.workload == "mobilenet" else 128)

    # launch on an explicit stream of our own; HIP events are recorded on that same stream
    stream = L.bcnn_hip_stream_create()
    L.bcnn_hip_set_stream(stream)

    overlap = not args.no_overlap and os.environ.get("BENCH_NO_OVERLAP") != "1"
    wl = Workload(args.workload, n, rank, world, dev, L, stream, input_grad=args.input_grad, overlap=overlap, comm=args.comm)
    dt, prof, profiled_steps = wl.run(args.steps, args.warmup)

    def with_alone(roof, w, workload):
        """the dominant class once more from the untimed `alone` leg (weight gradients on the caller's stream)"""
        pa = getattr(w, "prof_alone", None)
        if roof and pa:
            al = roofline_of(pa[0], workload, False, pick=roof["kernel"])
            if al:
                roof["alone"] = {"avg_ms": al["avg_ms"], "achieved": al["achieved"], "frac": al["frac"], "unit": al["unit"],
                                 "note": "same class in 3 untimed steps with every kernel on one stream: in the timed region it "
                                         "shares the chip with the other stream's kernels and its events span that time"}
        return roof

    if world > 1:
        tmax = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    line = None
    if rank == 0:
        default_shape = n == wl.default_n and not args.input_grad
        out = {
            "metric": "images/sec fwd+bwd", "value": round(args.steps * n * world / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_extra": wl.warmup_extra,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.desc, "batch_per_gpu": n, "global_batch": n * world, "parallelism": "dp%d" % world,
                       "comm": ("none (one rank)" if world == 1 and os.environ.get("BENCH_FORCE_DP") != "1" else
                                "in-library RCCL (csrc/comm.hip, bcnn_set_data_parallel_comm)" if args.comm == "inlib" else
                                "torch.distributed (%s)" % dist.get_backend())},
            "roofline": with_alone(roofline_of(prof, args.workload, default_shape), wl, args.workload),
            "profiled_steps": profiled_steps,
            "kernel_classes": class_table(prof, profiled_steps),
            "kernel_classes_alone": class_table(*wl.prof_alone) if getattr(wl, "prof_alone", None) else None,
            # bcnn_backward queues the weight-gradient kernels on a second HIP stream of the library (include/bcnn_hip.h:
            # bcnn_hip_conv_side_stream_mode), where they run next to the batch-norm / pooling sweeps and the data gradients
            # of the layers in front. Each class is timed with HIP events on the stream its kernels are launched on, so the
            # *_dw classes overlap the others in time: the classes' sum exceeds ms_per_step, and a class's rate is what it
            # reaches while it shares the chip.
            "streams": "weight gradients on a second stream: per-class times overlap (sum > ms_per_step)",
        }
        if os.environ.get("BENCH_TEST_CHECKSUM") == "1" and wl.net is not None:
            pptr, psize = wl.net.parameter_arena()
            params = torch.as_tensor(capi.DeviceArray(pptr, psize), device=dev)
            out["param_checksum"] = [float(params.double().sum()), float(params.double().abs().sum())]
    sample_n = wl.sample_n
    wl.close()
    del wl

    # The other single-GPU configurations of BASELINE.json, timed briefly by the same process so that the driver's
    # clock covers them too: configs[1] (the 3x3 conv `north_star` quotes its >= 50 % MFMA target on) and configs[4]
    # (MobileNet-v1, the HBM-bound config). Same step definition and timers as above; `value` stays the headline's.
    if rank == 0 and world == 1 and args.workload == "resnet18" and args.batch is None and not args.no_side_workloads \
            and os.environ.get("BENCH_FORCE_DP") != "1":
        side = {}
        for name, sn, ssteps, swarm in (("conv3x3", 128, 40, 3), ("mobilenet", 256, 6, 2)):
            torch.cuda.empty_cache()
            w2 = Workload(name, sn, 0, 1, dev, L, stream)
            sdt, sprof, sprofiled = w2.run(ssteps, swarm)
            side[name] = {"config": w2.desc, "images_per_s": round(ssteps * sn / sdt, 2),
                          "ms_per_step": round(sdt / ssteps * 1e3, 4), "steps": ssteps, "warmup": swarm,
                          "warmup_extra": w2.warmup_extra,
                          "roofline": with_alone(roofline_of(sprof, name, True), w2, name),
                          "kernel_classes": class_table(sprof, sprofiled),
                          "kernel_classes_alone": class_table(*w2.prof_alone) if getattr(w2, "prof_alone", None) else None}
            side_sample = w2.sample_n
            w2.close()
            del w2
            if not args.no_cpu_baseline:  # the reference on the same host, a bounded sample of THIS workload
                side[name]["cpu_baseline"] = cpu_baseline(name, side_sample)
        out["workloads"] = side
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # reported by the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.workload, sample_n)
        line = json.dumps(out)
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
