"""Is one train step's arithmetic bit-reproducible?  Same parameters, same frame, K evaluations of loss_terms() + backward:
every loss term, the image and every gradient must repeat bit for bit (python tools/determinism_check.py [P] [W] [H] [K])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DGM_SIDE_STREAM"] = "0"
import test_trainer_dp_gpu as H  # noqa: E402

P, W, Hh, K = (int(x) for x in (sys.argv[1:5] + ["60000", "640", "512", "6"][len(sys.argv) - 1:]))
tr = H.make_trainer(0, 1, P=P, W=W, H=Hh, side_stream=False)
it = tr.opt.warm_up + 10
names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"] + [f"deform.{n}" for n, _ in tr.deform.net.named_parameters()] + \
        [f"deform_back.{n}" for n, _ in tr.deform_back.net.named_parameters()]
params = tr.g.parameters()[:6] + list(tr.deform.net.parameters()) + list(tr.deform_back.net.parameters())
ref = None
bad = {}
for k in range(K):
    for p in tr.params:
        p.grad = None
    losses, pkg = tr.loss_terms(tr.cameras[1], it)
    total = None
    for v in losses.values():
        total = v if total is None else total + v
    total.backward()
    torch.cuda.synchronize()
    cur = {"image": pkg["render"].detach().clone(), "radii": pkg["radii"].clone()}
    cur.update({"loss." + n: v.detach().clone() for n, v in losses.items()})
    cur.update({"grad." + n: (p.grad.clone() if p.grad is not None else torch.zeros(1)) for n, p in zip(names, params)})
    if ref is None:
        ref = cur
        continue
    for n in cur:
        if not torch.equal(cur[n], ref[n]):
            d = (cur[n].float() - ref[n].float()).abs()
            bad.setdefault(n, []).append((k, float(d.max()), int((d > 0).sum())))
print(f"P={P} {W}x{Hh}, {K} evaluations: {'all bit-identical' if not bad else 'DIFFERENCES'}")
for n, v in list(bad.items())[:20]:
    print("  ", n, v[:3])
