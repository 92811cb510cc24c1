"""RCCL on the GPU box: `nccl` process group (world_size 1: one GPU is visible to gpurun) all-reducing the train step's
real gradient bucket after a kernel launched through the C ABI on the same stream.  Proves librccl loads and
initialises and that ctypes-launched kernels and RCCL order correctly on torch's current stream; the multi-rank
semantics of the data-parallel step are covered by the gloo world-2 tests (test_trainer_dp*.py)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_rccl_world1_allreduce_of_the_gradient_bucket():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "allreduce_bench.py"), "--mb", "27.8", "--iters", "5"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    print(rec)
    assert rec["world"] == 1 and rec["ms"] > 0


def _rccl_worker(rank, world, port, out_dir, overlap):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import test_trainer_dp_gpu as tdp
    real = torch.device
    tdp.torch.device = lambda *a, **k: real("cuda", rank) if a and str(a[0]).startswith("cuda") else real(*a, **k)  # one GPU per rank
    try:
        tr = tdp.make_trainer(rank, world, overlap=overlap)
    finally:
        tdp.torch.device = real
    assert tr.pack and (tr._early is not None) == overlap and dist.get_backend() == "nccl" and dist.get_world_size() == world
    it = tr.opt.warm_up + 10
    for s in range(3):
        tr.step(it + s)
    torch.cuda.synchronize()
    torch.save(tdp.snapshot(tr), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_world2_step(overlap):
    """Two ranks, one GPU each, backend nccl (= RCCL over xGMI): three data-parallel train steps, with the single bucket after
    backward (the default) and with the early Gaussian-bucket all-reduce (overlap=True); the replicas must stay bit-identical.  Skips on boxes with fewer than two GPUs (gpurun's have one; the
    multi-rank logic is also covered over gloo by test_trainer_dp*.py)."""
    import tempfile

    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    with tempfile.TemporaryDirectory() as d:
        port = 29400 + (os.getpid() % 400)
        mp.start_processes(_rccl_worker, args=(2, port + (1 if overlap else 0), d, overlap), nprocs=2, join=True, start_method="spawn")
        r0, r1 = torch.load(os.path.join(d, "rank0.pt")), torch.load(os.path.join(d, "rank1.pt"))
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,steps,sync_free", [("cfg1", 6, False), ("cfg2", 8, False), ("cfg1", 12, True)])
def test_bench_gpus_flag_launches_its_own_ranks(workload, steps, sync_free):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself and report n_gpus = 2 -- exercised on a
    one-GPU box through DGM_BENCH_SHARE_GPU=1 (both ranks on cuda:0, gloo), which runs the same self-launch, rendezvous,
    barrier / max-over-ranks timing and reporting code as the RCCL configuration; at cfg1 (short) and at cfg2, the metric's own
    workload (27.8 MB bucket; the record is kept in gpurun_out/dp2_shared_gpu_<workload>.json -- its it/s is two ranks time-slicing
    ONE GPU over gloo, not a scaling number)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DGM_BENCH_SHARE_GPU="1", DGM_BENCH_STEADY_STEPS="0")
    if sync_free:  # the rasterizer forward that never waits for the device, its capacity check deferred behind the backward
        env.update(DGM_SYNC_FREE="1")
    else:
        env.pop("DGM_SYNC_FREE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "2", "--workload",
                          workload, "--no-cpu-baseline", "--no-extras"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"dp2_shared_gpu_{workload}{'_sync_free' if sync_free else ''}.json"), "w") as fh:
        json.dump({k: rec.get(k) for k in ("metric", "value", "n_gpus", "steps", "ms_per_step", "rccl_world_size", "replicas_identical",
                                          "data_parallel", "allreduce", "config", "host_ms_per_step")}, fh, indent=1)
    assert rec["host_ms_per_step"]["sync_free_forward"] is sync_free
    assert rec["n_gpus"] == 2 and rec["rccl_world_size"] == 2 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["allreduce"]["world_size_observed"] == 2 and len(rec["allreduce"]["bucket_bytes"]) == 1  # (default: one flat bucket)
    # the line vouches for itself: parameter + Adam-moment hashes of all ranks compared after the timed region, both exchange forms
    assert rec["replicas_identical"] is True
    dp = rec["data_parallel"]
    assert dp["overlap"] is False and dp["exchange_ms_per_step"] > 0
    assert dp["overlap_on"]["replicas_identical"] is True and dp["overlap_on"]["value"] > 0


def test_bench_gpus_flag_refuses_too_few_devices():
    """No silent fallback: asking for more ranks than there are GPUs must fail loudly (here: a box without any GPU)."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("box has 8 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DGM_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode != 0 and "refusing" in out.stdout
