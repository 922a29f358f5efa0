#!/usr/bin/env python3
"""Does RCCL come up on this box at all?  One process per visible GPU (WORLD_SIZE from torchrun, or a single rank started directly),
backend "nccl" through utils.parallel's own conventions (device_id, 127.0.0.1 rendezvous), then every collective the step uses
-- all_reduce SUM / MAX / MIN (asynchronous, waited on the compute stream), all_gather_into_tensor, broadcast, all_gather_object,
barrier -- on device tensors, with the results checked.  Prints ONE JSON line on rank 0.

A communicator of ONE rank runs no ring (RCCL copies or returns), but it does go through ncclCommInitRank, ProcessGroupNCCL's
stream / event plumbing and the dmabuf IPC setting of the image: what a 1-GPU box can show of the N > 1 path's plumbing.
    python tools/rccl_probe.py                                   # one rank
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/rccl_probe.py
"""
import datetime
import json
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "no GPU visible"
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=rk, world_size=ws, device_id=dev, timeout=datetime.timedelta(seconds=120))
    t_init = time.perf_counter() - t0
    out = {"world": ws, "backend": dist.get_backend(), "init_s": round(t_init, 3), "device": torch.cuda.get_device_name(dev)}
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                        # informative only
        out["rccl_version"] = "n/a (%s)" % type(e).__name__

    # all_reduce, asynchronous, issued from a side stream's producer the way GradientAllReducer does it
    n = 4 << 20
    flat = torch.full((n,), float(rk + 1), device=dev)
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    work.wait()
    want = ws * (ws + 1) / 2
    ok_sum = bool((flat == want).all())
    av = torch.full((1024,), float(rk + 1), device=dev)                          # the gradient reducer averages in the collective
    try:
        dist.all_reduce(av, op=dist.ReduceOp.AVG)
        ok_avg = bool((av - (ws + 1) / 2.0).abs().max() < 1e-6)
    except Exception as e:                                                       # (the reducer falls back to SUM + scaling)
        ok_avg = "refused: %s" % type(e).__name__
    mx = torch.tensor([float(rk)], device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    mn = torch.tensor([float(rk)], device=dev)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    ok_minmax = float(mx) == ws - 1 and float(mn) == 0.0

    stat = torch.arange(512, dtype=torch.float32, device=dev) + 1000.0 * rk
    gathered = torch.empty(ws, 512, device=dev)
    dist.all_gather_into_tensor(gathered.view(-1), stat.view(-1))
    ok_gather = all(bool((gathered[r] == torch.arange(512, device=dev) + 1000.0 * r).all()) for r in range(ws))

    conf = torch.full((19 * 19 + 2,), rk + 1, dtype=torch.int64, device=dev)      # the evaluation's exchange (networks/evaluate.py)
    dist.all_reduce(conf)
    ok_int = bool((conf == ws * (ws + 1) // 2).all())
    t64 = torch.tensor([float(rk)], device=dev, dtype=torch.float64)             # bench.py's max-over-ranks of the timed region
    dist.all_reduce(t64, op=dist.ReduceOp.MAX)
    ok_int = ok_int and float(t64) == ws - 1

    b = torch.full((1024,), float(rk + 7), device=dev)
    dist.broadcast(b, 0)
    ok_bcast = bool((b == 7.0).all())

    objs = [None] * ws
    dist.all_gather_object(objs, ("rank", rk))
    ok_obj = objs == [("rank", r) for r in range(ws)]
    dist.barrier(device_ids=[dev.index])

    # bandwidth of a gradient-bucket-sized all-reduce (16 MiB, the reducer's bucket): informative
    bucket = torch.ones(4 << 20, device=dev)
    for _ in range(3):
        dist.all_reduce(bucket)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        dist.all_reduce(bucket)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / reps
    out.update({"all_reduce_sum": ok_sum, "all_reduce_min_max": ok_minmax, "all_gather_into_tensor": ok_gather, "broadcast": ok_bcast,
                "all_reduce_int64_and_f64": ok_int, "all_reduce_avg": ok_avg, "all_gather_object": ok_obj, "all_reduce_16MiB_us": round(us, 1),
                "all_reduce_16MiB_busbw_GBs": round(2 * (ws - 1) / ws * bucket.numel() * 4 / us / 1e3, 2) if ws > 1 else None})
    good = ok_sum and ok_minmax and ok_gather and ok_bcast and ok_obj and ok_int
    out["ok"] = good
    if rk == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()
