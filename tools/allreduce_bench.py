"""RCCL all-reduce micro-benchmark of the train step's gradient bucket (one process per GPU):

    python tools/allreduce_bench.py                                  # world 1 on cuda:0 (proves librccl + stream order)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/allreduce_bench.py --mb 27.8 35.2 74 124

For each payload (MB of fp32; defaults = the cfg2 bucket measured by Trainer.grad_bytes(), the SURVEY section 8e sizes
for cfg2/cfg4/cfg5) it times `iters` all-reduce(SUM) calls on the compute stream with hipEvents, after the bucket was
written by a kernel launched through the C ABI on the same stream (dgm_adam_step via ctypes: checks that RCCL and the
ctypes-launched kernels order correctly on torch's current stream).  Rank 0 prints one JSON line per payload:
algorithmic bus bandwidth = 2 (W-1)/W * bytes / time.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, nargs="*", default=[27.8, 35.2, 74.0, 124.0])
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    L = importlib.import_module("dg-mesh_amd._lib").lib()
    for mb in args.mb:
        n = int(mb * 1e6 / 4)
        flat = torch.zeros(n, device=dev)
        # write the bucket with a ctypes-launched kernel on the current stream: p -= lr * m/(sqrt(v)+eps) with g = 1
        m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        g = torch.ones(n, device=dev)
        VP = ctypes.c_void_p * 1
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.dgm_adam_step(1, VP(flat.data_ptr()), VP(g.data_ptr()), VP(m.data_ptr()), VP(v.data_ptr()),
                             (ctypes.c_longlong * 1)(n), (ctypes.c_float * 1)(1.0), (ctypes.c_int * 1)(1), 0.9, 0.999, 1e-15, st)
        assert rc == 0
        dist.all_reduce(flat)  # first step of Adam with g = 1 moves every element by -lr = -1  ->  sum = -world
        torch.cuda.synchronize()
        want = -float(world)
        assert abs(flat[0].item() - want) < 1e-5 and abs(flat[-1].item() - want) < 1e-5 and \
            abs(flat.double().mean().item() - want) < 1e-5, "all-reduce after a ctypes-launched kernel returned wrong data"
        for _ in range(3):
            dist.all_reduce(flat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        dist.barrier()
        e0.record()
        for _ in range(args.iters):
            dist.all_reduce(flat)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        t = torch.tensor([ms], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            busbw = (2.0 * (world - 1) / world) * n * 4 / (t.item() * 1e-3) / 1e9 if world > 1 else 0.0
            print(json.dumps({"allreduce_mb": mb, "world": world, "ms": round(t.item(), 4), "busbw_GBps": round(busbw, 1),
                              "backend": "nccl (RCCL)"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
