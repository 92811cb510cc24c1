"""A/B timing of Trainer options inside one process: python tools/ab_step.py  (GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
tr, _ = bench.build_scene(dev, 0, 1, "hip")
it0 = 5000
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        tr.step(it0 + i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(10):
    tr.step(it0)
for rep in range(3):
    for fused in (False, True):
        tr.fused_glue = fused
        run(5)
        print(f"fused_glue={fused}: {run(30):.3f} ms/step", flush=True)
