"""Probe: does the gloo backend run all_gather_into_tensor(async_op=True) / all_reduce / barrier / broadcast on HIP tensors when two
ranks share one GPU?  (Rehearsal of the multi-rank path on the 1-GPU boxes: RCCL refuses two ranks on one device.)"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x = torch.full((3, 5), float(rank + 1), device=dev)
    out = torch.empty(world, 3, 5, device=dev)
    w = dist.all_gather_into_tensor(out.view(world * 3, 5), x, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    t = torch.tensor([float(rank)], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    b = torch.tensor([7.0 if rank == 0 else 0.0], device=dev)
    dist.broadcast(b, 0)
    dist.barrier()
    print(f"rank {rank}: gathered {out[:, 0, 0].tolist()} max {t.item()} bcast {b.item()}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29511), nprocs=2, join=True)
    print("PROBE OK")
