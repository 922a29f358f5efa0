"""GPU diagnostic: gloo all_reduce of large CUDA tensors with N ranks sharing one device (the transport of the N-ranks-on-one-GPU
tests) -- async, several buckets in flight, the way GradientAllReducer issues them."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    bufs = [torch.full((n,), float(rank + 1) * (i + 1), device=dev) for i, n in enumerate((13_000_000, 4_200_000, 7, 3_300_000))]
    side = torch.cuda.Stream()
    works = [dist.all_reduce(b, async_op=True) for b in bufs[:2]]
    with torch.cuda.stream(side):
        works += [dist.all_reduce(b, async_op=True) for b in bufs[2:]]
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    want = world * (world + 1) / 2
    bad = [(i, float(b.min()), float(b.max()), want * (i + 1)) for i, b in enumerate(bufs) if not torch.all(b == want * (i + 1))]
    print("rank %d of %d: %s" % (rank, world, "all buckets exact" if not bad else "MISMATCH %s" % bad), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)
