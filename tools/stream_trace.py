"""Which tensor deviates first?  One-stream reference vs two-stream runs; after every step the parameters and gradients of the
three groups (Gaussians, deform, deform_back) are hashed ON THE GPU into a log (no host synchronisation, each group on the stream
that owns it) and compared at the end (python tools/stream_trace.py [repeats] [steps] [P W H])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DGM_SIDE_STREAM"] = "1"
import test_trainer_dp_gpu as H  # noqa: E402

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P, W, Hh = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (60000, 640, 512)
KEYS = ["grad.gauss", "param.gauss", "grad.deform", "param.deform", "grad.deform_back", "param.deform_back", "image", "radii",
        "g.xyz", "g.f_dc", "g.f_rest", "g.opacity", "g.scaling", "g.rotation", "loss"]


def h(ts):
    acc = None
    for t in ts:
        if t is None:
            continue
        v = t.detach().contiguous().view(torch.int32).to(torch.int64).sum()
        acc = v if acc is None else acc * 31 + v
    return acc


NF = int(os.environ.get("TRACE_FRAMES", "6"))
if os.environ.get("TRACE_NOGC") == "1":
    import gc
    gc.disable()


def run(side):
    tr = H.make_trainer(0, 1, P=P, W=W, H=Hh, side_stream=side, n_frames=NF)
    groups = {"gauss": tr.g.parameters()[:6], "deform": list(tr.deform.net.parameters()), "deform_back": list(tr.deform_back.net.parameters())}
    it = tr.opt.warm_up + 10
    log = torch.zeros(steps, len(KEYS), dtype=torch.int64, device="cuda")
    side_log = torch.zeros(steps, len(KEYS), dtype=torch.int64, device="cuda")
    for s in range(steps):
        loss_s, pkg = tr.step(it + s)
        log[s, KEYS.index("image")] = h([pkg["render"]])
        log[s, KEYS.index("radii")] = h([pkg["radii"].to(torch.int32)])
        log[s, KEYS.index("loss")] = h([loss_s.reshape(1)])
        for i_, n_ in enumerate(("g.xyz", "g.f_dc", "g.f_rest", "g.opacity", "g.scaling", "g.rotation")):
            log[s, KEYS.index(n_)] = h([groups["gauss"][i_].grad])
        for gname in ("gauss", "deform"):
            log[s, KEYS.index("grad." + gname)] = h([p.grad for p in groups[gname]])
            log[s, KEYS.index("param." + gname)] = h(groups[gname])
        ctx = torch.cuda.stream(tr.side_stream) if tr.side_stream is not None and tr.side_defer else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            side_log[s, KEYS.index("grad.deform_back")] = h([p.grad for p in groups["deform_back"]])
            side_log[s, KEYS.index("param.deform_back")] = h(groups["deform_back"])
    torch.cuda.synchronize()
    return (log + side_log).cpu()


ref = run(False)
ref2 = run(False)
print("reference repeats itself:", bool((ref == ref2).all()))
hits = {}
for r in range(repeats):
    got = run(True)
    bad = (got != ref)
    if bad.any():
        s = int(bad.any(dim=1).nonzero()[0])
        first = [KEYS[i] for i in bad[s].nonzero().flatten().tolist()]
        hits[tuple(first)] = hits.get(tuple(first), 0) + 1
        print(f"repeat {r}: first deviation at step {s}: {first}", flush=True)
print("deviating runs:", sum(hits.values()), "of", repeats, hits)
