"""Does a train step read memory it has not written?  A one-stream trainer is stepped twice from the same start: plainly, and with
every free block of the allocator filled with 0xFF bytes (NaN patterns) before each step.  All parameters must stay finite and
bit-identical (python tools/poison_steps.py [steps] [P W H])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_trainer_dp_gpu as H  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P, W, Hh = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (60000, 640, 512)


def poison(fill):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 30,), fill, dtype=torch.uint8, device="cuda") for _ in range(8)]
    junk += [torch.full((8 << 20,), fill, dtype=torch.uint8, device="cuda") for _ in range(48)]
    junk += [torch.full((1 << 20,), fill, dtype=torch.uint8, device="cuda") for _ in range(64)]
    del junk


def run(fill):
    tr = H.make_trainer(0, 1, P=P, W=W, H=Hh)
    it = tr.opt.warm_up + 10
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    for s in range(steps):
        if fill is not None:
            poison(fill)
        tr.step(it + s)
        torch.cuda.synchronize()
        bad = [n for n, p in zip(names, tr.g.parameters()[:6]) if not torch.isfinite(p).all()]
        bad += [f"deform.{n}" for n, p in tr.deform.net.named_parameters() if not torch.isfinite(p).all()]
        bad += [f"deform_back.{n}" for n, p in tr.deform_back.net.named_parameters() if not torch.isfinite(p).all()]
        if bad:
            print(f"fill={fill}: non-finite parameters after step {s}: {bad[:8]}")
            break
    return H.snapshot(tr)


a = run(None)
b = run(255)
c = run(0)
print("poisoned (0xFF) run equals the plain run:", all(torch.equal(x, y) for x, y in zip(a, b)))
print("zero-filled run equals the plain run:", all(torch.equal(x, y) for x, y in zip(a, c)))
