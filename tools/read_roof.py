"""How fast does a pure READ stream go on this GPU?  torch.sum over fp32 tensors of the sizes the library's read-only launches move,
against a copy of the same bytes (read + write).  python tools/read_roof.py  (GPU)"""
import json
import torch

dev = torch.device("cuda", 0)
out = {}
for mb in (102, 141, 320, 1024):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("sum", lambda: x.sum(), 4 * n), ("copy", lambda: y.copy_(x), 8 * n)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 50
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / iters
        out[f"{name}_{mb}MB"] = {"us": round(us, 2), "TBps": round(nbytes / us / 1e6, 3)}
print(json.dumps(out))
