"""Data-parallel step, world_size 2 over gloo on CPU: batch sharding + ONE all-reduce(sum) of the flat
gradient arena + the DP-aware SGD rule (divide by the GLOBAL batch, leave momentum*g/world in the buffer)
must reproduce a single-process run on the concatenated batch, including the reference's
momentum-inside-the-gradient-buffer behaviour (bcnn_learner.c:67-83).
The math runs on the CPU oracle here; the same rule is implemented in C in bcnn_node_sgd_step
(bcnn_amd/host/bcnn_core.c) and checked on the GPU by tests/test_dp_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import orc_bind

SHAPE = dict(c=3, h=8, w=8, f=8, k=3, s=1, p=1)
STEPS, LR, MOM, DECAY = 3, 0.05, 0.9, 5e-4


def _fwd_bwd(x, wt, bias, dy, dw, db):
    n = x.shape[0]
    case = dict(op="conv", n=n, g=1, bn=0, act=2, input_grad=0, mode=1, x=x, wt=wt, bias=bias, dy=dy,
                dw0=dw, db0=db, **SHAPE)
    out = orc_bind.orc_conv(case)
    return out["dw"], out["db"]


def _data(world_batch):
    rs = np.random.RandomState(3)
    c, h, w, f, k = (SHAPE[q] for q in "chwfk")
    x = rs.uniform(-1, 1, (STEPS, world_batch, c, h, w)).astype(np.float32)
    dy = (rs.uniform(-1, 1, (STEPS, world_batch, f, h, w)) * 0.1).astype(np.float32)
    wt = rs.uniform(-0.3, 0.3, (f, c, k, k)).astype(np.float32)
    bias = rs.uniform(-0.1, 0.1, (f,)).astype(np.float32)
    return x, dy, wt, bias


def _single(world_batch):
    x, dy, wt, bias = _data(world_batch)
    dw, db = np.zeros_like(wt), np.zeros_like(bias)
    L = orc_bind.lib()
    for s in range(STEPS):
        dw, db = _fwd_bwd(x[s], wt, bias, dy[s], dw, db)
        L.orc_sgd_update(orc_bind.P(wt), orc_bind.P(bias), orc_bind.P(dw), orc_bind.P(db), wt.size, bias.size,
                         world_batch, LR, MOM, DECAY)
    return wt, bias, dw, db


def _worker(rank, world, port, local_batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, dy, wt, bias = _data(local_batch * world)
    lo, hi = rank * local_batch, (rank + 1) * local_batch      # this rank's shard of every batch
    arena = np.zeros(wt.size + bias.size, np.float32)           # flat gradient arena: [dW | db]
    dw, db = arena[:wt.size].reshape(wt.shape), arena[wt.size:]
    L = orc_bind.lib()
    for s in range(STEPS):
        ndw, ndb = _fwd_bwd(x[s, lo:hi].copy(), wt, bias, dy[s, lo:hi].copy(), dw.copy(), db.copy())
        dw[...] = ndw
        db[...] = ndb
        t = torch.from_numpy(arena)
        dist.all_reduce(t)                                       # the one collective of a step
        # DP rule: global batch, momentum/world so that the next all-reduce reconstitutes ONE carry
        L.orc_sgd_update(orc_bind.P(wt), orc_bind.P(bias), orc_bind.P(dw), orc_bind.P(db), wt.size, bias.size,
                         local_batch * world, LR, MOM / world, DECAY)
    if rank == 0:
        q.put((wt.copy(), bias.copy(), dw.copy() * world, db.copy() * world))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_process_on_global_batch():
    orc_bind.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    wt, bias, dw, db = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ewt, ebias, edw, edb = _single(4)
    for name, a, b in (("w", wt, ewt), ("b", bias, ebias), ("dw carry", dw, edw), ("db carry", db, edb)):
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert err < 1e-5, (name, err)


# ---- the bucketed, overlapped all-reduce used by bench.py (bcnn_amd/dp.py) over gloo ----------------------------
def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bcnn_amd.dp import BucketedAllReduce
    rs = np.random.RandomState(100 + rank)
    size = 10_000
    arena = torch.from_numpy(rs.uniform(-1, 1, size).astype(np.float32))
    mine = arena.clone()
    bar = BucketedAllReduce(arena, 2048)
    # tail ranges as bcnn_backward reports them: uneven node sizes, last node first
    cuts = [size, 9_990, 9_000, 8_999, 6_000, 5_000, 1_500, 100, 0]
    result = {}
    for step in range(2):                       # begin() must reset the state between steps
        arena.copy_(mine)
        bar.begin()
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            bar.on_ready(lo, hi - lo)
        left = bar.finish()
        assert bar.failed is None and left == 0
        covered = sorted(bar.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == size
        assert all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:])), covered
        assert 2 <= len(covered) <= 6 and all(h - l >= 2048 for l, h in covered[1:]), covered
        result[step] = arena.clone()
    # a range that does not continue the tail is recorded as a failure, never raised from the callback
    bar.begin()
    bar.on_ready(size - 10, 5)
    assert bar.failed is not None
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered)
    if rank == 0:
        q.put((float((result[0] - expect).abs().max()), float((result[1] - expect).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_over_gloo_sums_every_range_exactly_once():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    e0, e1 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert e0 < 1e-6 and e1 < 1e-6, (e0, e1)
