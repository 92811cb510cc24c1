"""Which train step makes torch's caching allocator go to hipMalloc, and for how many bytes?  (bench.py's
`device_allocations_in_timed_region`)   python tools/alloc_probe.py [steps=40]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")
RZ = importlib.import_module("dg-mesh_amd.rasterizer")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
tr, _ = bench.build_scene(dev, 0, 1, "hip")
it0 = tr.opt.warm_up + 2000
L = importlib.import_module("dg-mesh_amd._lib")
mode = os.environ.get("PROBE_MODE", "")  # "bench": freeze the GC and switch the deferred stage timers on after 15 steps, as bench.py does
st = lambda k: torch.cuda.memory_stats(dev).get(k, 0)
a0, r0 = st("num_device_alloc"), st("reserved_bytes.all.current")
for i in range(steps):
    if mode == "bench" and i == 10:
        tr.freeze_gc()
    if mode == "bench" and i == 15:
        L.lib().dgm_set_profiling_sampling(8)
        L.lib().dgm_set_profiling(2)
        print("profiling on at step 15")
    tr.step(it0 + i)
    torch.cuda.synchronize()
    a1, r1 = st("num_device_alloc"), st("reserved_bytes.all.current")
    if a1 != a0:
        print(f"step {i}: +{a1 - a0} device allocation(s), reserved {r0 / 2**20:.0f} -> {r1 / 2**20:.0f} MiB (+{(r1 - r0) / 2**20:.1f}); frame {tr.last_frame} "
              f"R {RZ.LAST_NUM_RENDERED} binning capacity {RZ._BIN_CAPACITY}")
    a0, r0 = a1, r1
print("done; reserved", r0 / 2**20, "MiB; allocated peak", st("allocated_bytes.all.peak") / 2**20)
