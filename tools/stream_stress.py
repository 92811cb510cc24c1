"""Stress check of the two-stream step: N trainers with two streams against one with one stream, many steps, parameters and
losses must be bit-identical every time (python tools/stream_stress.py [repeats] [steps])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DGM_SIDE_STREAM"] = "1"
import test_trainer_dp_gpu as H  # noqa: E402

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def poison(fill):
    """every block the allocator will hand out next holds `fill` bytes (0xFF: NaN patterns)"""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 30,), fill, dtype=torch.uint8, device="cuda") for _ in range(6)]
    junk += [torch.full((8 << 20,), fill, dtype=torch.uint8, device="cuda") for _ in range(32)]
    del junk


def run(side, P, W, Hh, fill=255):
    poison(fill)
    tr = H.make_trainer(0, 1, P=P, W=W, H=Hh, side_stream=side)
    it = tr.opt.warm_up + 10
    ls = [tr.step(it + s)[0] for s in range(steps)]
    torch.cuda.synchronize()
    return H.snapshot(tr), [float(x) for x in ls]


bad = 0
sizes = ((3000, 160, 128), (20000, 320, 256), (60000, 640, 512))
if os.environ.get("STRESS_SIZE"):
    sizes = tuple(s for s in sizes if s[0] == int(os.environ["STRESS_SIZE"]))
for P, W, Hh in sizes:
    ref, ref_l = run(False, P, W, Hh)
    ref2, ref2_l = run(False, P, W, Hh, fill=0)
    print(f"P={P}: one-stream run repeats itself: {all(torch.equal(a, b) for a, b in zip(ref, ref2))}, losses {ref_l == ref2_l}")
    for r in range(repeats):
        got, got_l = run(os.environ.get("STRESS_SIDE", "1") == "1", P, W, Hh)
        same = all(torch.equal(a, b) for a, b in zip(ref, got))
        same_l = ref_l == got_l
        if not (same and same_l):
            bad += 1
            k = [i for i, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a, b)]
            first = next((i for i, (x, y) in enumerate(zip(ref_l, got_l)) if x != y), None)
            print(f"MISMATCH P={P} repeat {r}: tensors {k[:6]}, first differing loss at step {first}: {ref_l[first] if first is not None else None} vs {got_l[first] if first is not None else None}")
    print(f"P={P}: {repeats} two-stream runs of {steps} steps compared", flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
