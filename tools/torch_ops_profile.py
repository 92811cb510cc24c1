"""Which torch (non-dgm) device kernels and copies remain in the train step, by the ATen op that launched them:
python tools/torch_ops_profile.py [steps]   (GPU; torch.profiler)"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bench.WORKLOAD = os.environ.get("DGM_BENCH_WORKLOAD", "cfg2")
tr, _ = bench.build_scene(dev, 0, 1, "hip")
it0 = tr.opt.warm_up + 2000
for i in range(15):
    tr.step(it0 + i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(steps):
        tr.step(it0 + 20 + i)
    torch.cuda.synchronize()
ev = prof.events()
agg = defaultdict(lambda: [0, 0.0, ""])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for k in e.kernels:
            if k.name.startswith("dgm::") or "dgm" in k.name:
                continue
            stack = [s for s in (e.stack or []) if "dg-mesh_amd" in s or "bench" in s]
            key = (e.name, k.name[:70], stack[0][-70:] if stack else "")
            agg[key][0] += 1
            agg[key][1] += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = 0.0
for (op, kern, where), (n, us, _) in rows[:40]:
    tot += us
    print(f"{n/steps:6.1f}/step {us/steps:8.1f} us/step  {op:34s} {kern:70s} {where}")
print("total non-dgm device time per step (us):", sum(v[1] for v in agg.values()) / steps)
