"""Host-timed duration of each of the first train steps after set-up (sync per step): does the step settle within bench.py's
priming?   python tools/step_times.py [steps=80]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
dev = torch.device("cuda", 0)
tr, _ = bench.build_scene(dev, 0, 1, "hip")
it0 = tr.opt.warm_up + 2000
ts = []
for i in range(steps):
    if i == 10:
        tr.freeze_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(it0 + i)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
for a in range(0, steps, 10):
    print(f"steps {a:3d}-{a+9:3d}: " + " ".join(f"{t:.2f}" for t in ts[a:a + 10]))
