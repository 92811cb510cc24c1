"""cProfile of the host side of the train step (bench.py's scene): python tools/host_profile.py [steps]"""
import cProfile
import importlib
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bench.WORKLOAD = os.environ.get("DGM_BENCH_WORKLOAD", "cfg2")
tr, _ = bench.build_scene(dev, 0, 1, "hip")
it0 = tr.opt.warm_up + 2000
for i in range(15):
    tr.step(it0 + i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    tr.step(it0 + 20 + i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(32)
