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
