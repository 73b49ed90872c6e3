#!/usr/bin/env python3
"""bench.py -- images/s forward+backward of bcnn's conv hot path on MI355X, one process per GPU.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched under
torch.distributed.run, one rank per GPU, RCCL all-reduce of the weight gradients after backward).
Rank 0 prints ONE JSON line with the whole-job throughput, the roofline of the dominant kernel
(duration measured live with HIP events on the launch stream) and a CPU baseline.

Workloads (BASELINE.json configs):
  conv3x3   configs[1]: one 3x3 s1 p1 conv, N=128 x 3 x 224 x 224 -> 64 channels (default)
A step = forward (conv + bias) and backward (bias gradient + dW; the layer is the net's first node so
its source carries no gradient and the reference computes no dX, bcnn_net.c:283 / bcnn_conv_layer.c:560).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA dense peak


def cpu_baseline_conv(c, h, w, f, k, s, p, sample_n, iters):
    """Reference (oracle/_ref, the unmodified bcnn built by oracle/Makefile) or, if that library did
    not travel, the repo's C restatement; bounded sample of the same layer. Checker code: only this
    baseline leg may touch oracle/."""
    import numpy as np
    from oracle import ref_bind
    rs = np.random.RandomState(0)
    if ref_bind.available():
        net = ref_bind.RefNet(mode=ref_bind.MODE_TRAIN, w=w, h=h, c=c, n=sample_n)
        node = net.conv(f, k, s, p, 1, 0, ref_bind.ACT_NONE, "input", "conv1")
        net.compile()
        net.data(net.node_src(node, 0))[...] = rs.uniform(-1, 1, (sample_n, c, h, w)).astype(np.float32)
        t, tf, tb = net.time_fwd_bwd(1, iters)
        cores = net.threads()
        net.close()
        kind = "reference"
    else:
        from oracle import orc_bind
        oh, ow = orc_bind.conv_out_hw(h, w, k, s, p)
        case = dict(op="conv", n=sample_n, c=c, h=h, w=w, f=f, k=k, s=s, p=p, g=1, bn=0, act=0,
                    input_grad=0, mode=1,
                    x=rs.uniform(-1, 1, (sample_n, c, h, w)).astype(np.float32),
                    wt=rs.uniform(-0.3, 0.3, (f, c, k, k)).astype(np.float32),
                    bias=np.zeros(f, np.float32),
                    dy=rs.uniform(-0.01, 0.01, (sample_n, f, oh, ow)).astype(np.float32))
        orc_bind.run_oracle(case)
        best = 1e30
        for _ in range(max(1, iters // 4)):
            t0 = time.perf_counter()
            orc_bind.run_oracle(case)
            best = min(best, time.perf_counter() - t0)
        t, cores, kind = best, os.cpu_count(), "port"
    return {"value": round(sample_n / t, 2), "unit": "images/s", "cores": int(cores), "kind": kind,
            "sample": "same conv layer, N=%d, fwd+bwd(dW+bias), best of %d iterations" % (sample_n, iters)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="conv3x3")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bcnn_amd import _lib, ops

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    L = _lib.load()
    L.bcnn_hip_set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # "nccl" IS RCCL on ROCm
    dev = torch.device("cuda", local_rank)

    assert args.workload == "conv3x3", "only the conv3x3 microbench is wired up in this round"
    n, c, h, w, f, k, s, p = args.batch, 3, 224, 224, 64, 3, 1, 1
    oh, ow = ops.conv_out_hw(h, w, k, s, p)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)   # every rank owns different images
    x = torch.rand((n, c, h, w), device=dev, generator=gen) * 2 - 1
    wgen = torch.Generator(device=dev).manual_seed(7)            # identical weights on every rank
    a = (3.0 / (c * k * k)) ** 0.5
    # one flat arena for parameters and one for their gradients => a single all-reduce per step
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

    # launch on an explicit stream of our own and time with HIP events recorded on that same stream
    stream = L.bcnn_hip_stream_create()
    L.bcnn_hip_set_stream(stream)
    torch.cuda.synchronize()
    ev = [[L.bcnn_hip_event_create() for _ in range(3)] for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            L.bcnn_hip_event_record(ev[i][0])
        ops.conv_forward(x, wt, bias, y, k, s, p, 1, 0)
        if i is not None:
            L.bcnn_hip_event_record(ev[i][1])
        ops.conv_backward(x, wt, y, dy, None, dw, db, k, s, p, 1, 0, ws)
        if i is not None:
            L.bcnn_hip_event_record(ev[i][2])
        if world > 1:
            L.bcnn_hip_sync()                 # gradients complete on our stream before RCCL reads them
            dist.all_reduce(grads)            # sum over ranks of the flat gradient arena (xGMI)
            torch.cuda.synchronize()          # the next backward accumulates into `grads`

    for _ in range(args.warmup):
        step()
    L.bcnn_hip_sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    L.bcnn_hip_sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    fwd_ms = sum(L.bcnn_hip_event_elapsed_ms(e[0], e[1]) for e in ev) / args.steps
    bwd_ms = sum(L.bcnn_hip_event_elapsed_ms(e[1], e[2]) for e in ev) / args.steps
    # algorithmic traffic per launch: forward reads x and W once, writes y once;
    # backward-dW reads x and dy once.
    fwd_bytes = 4.0 * (n * c * h * w + f * c * k * k + n * f * oh * ow)
    bwd_bytes = 4.0 * (n * c * h * w + n * f * oh * ow + f * c * k * k)
    flops = 2.0 * n * f * oh * ow * c * k * k  # per direction
    fwd_gbs = fwd_bytes / (fwd_ms * 1e-3) / 1e9
    bwd_gbs = bwd_bytes / (bwd_ms * 1e-3) / 1e9

    if rank == 0:
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_conv3x3_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("conv_fwd_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "images/sec fwd+bwd", "value": round(args.steps * n * world / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "conv3x3 s1 p1, N=%d x 3 x 224 x 224 -> 64 (BASELINE configs[1]), "
                                   "fwd + bwd(dW, dbias); no dX: the layer's source is the net input" % n,
                       "batch_per_gpu": n, "global_batch": n * world, "parallelism": "dp%d" % world},
            "roofline": {"kernel": "conv_fwd_igemm", "bound": "hbm", "achieved": round(fwd_gbs, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fwd_gbs / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "algorithmic_bytes": fwd_bytes, "avg_ms": round(fwd_ms, 4),
                         "mfma_tflops": round(flops / (fwd_ms * 1e-3) / 1e12, 2),
                         "mfma_frac": round(flops / (fwd_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)},
            "roofline_bwd": {"kernel": "conv_dw_kernel+finalize", "bound": "hbm", "achieved": round(bwd_gbs, 1),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bwd_gbs / HBM_PEAK_GBS, 4),
                             "algorithmic_bytes": bwd_bytes, "avg_ms": round(bwd_ms, 4),
                             "mfma_tflops": round(flops / (bwd_ms * 1e-3) / 1e12, 2)},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_conv(c, h, w, f, k, s, p, sample_n=16, iters=12)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
