"""Attribute the small kernels of one train step to PyTorch ops: python tools/profile_step.py  (needs a GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
tr, cfg = bench.build_scene(dev, 0, 1, "hip")
it0 = 5000
for i in range(5):
    tr.step(it0 + i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for i in range(3):
        tr.step(it0 + 5 + i)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=220, max_name_column_width=45, max_shapes_column_width=60))
print(prof.key_averages().table(sort_by="count", row_limit=80, max_name_column_width=50))
