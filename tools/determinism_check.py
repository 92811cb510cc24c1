"""Is one train step's arithmetic bit-reproducible?  Same parameters, same frame, K evaluations of loss_terms() + backward:
every loss term, the image and every gradient must repeat bit for bit (python tools/determinism_check.py [P] [W] [H] [K])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_trainer_dp_gpu as H  # noqa: E402

P, W, Hh, K = (int(x) for x in (sys.argv[1:5] + ["60000", "640", "512", "6"][len(sys.argv) - 1:]))
tr = H.make_trainer(0, 1, P=P, W=W, H=Hh)
it = tr.opt.warm_up + 10
names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"] + [f"deform.{n}" for n, _ in tr.deform.net.named_parameters()] + \
        [f"deform_back.{n}" for n, _ in tr.deform_back.net.named_parameters()]
params = tr.g.parameters()[:6] + list(tr.deform.net.parameters()) + list(tr.deform_back.net.parameters())
from conftest import pkg as _pkg0
_G = _pkg0("glue")
_orig_apply = _G.gaussian_apply
_cap = {}


def _capturing_apply(*a):
    outs = _orig_apply(*a)
    for n_, o_ in zip(("means3D", "scales", "rotations", "opacity"), outs):
        o_.retain_grad()
        _cap[n_] = o_
    return outs


_G.gaussian_apply = _capturing_apply   # (scene.render imports it at call time): the rasterizer's own gradient outputs
ADV = int(os.environ.get("CHECK_ADVANCE", "0"))      # train this many steps first (the state the stress runs deviate in)
for s_ in range(ADV):
    tr.step(it + s_)
it += ADV
MLPNOISE = os.environ.get("CHECK_NOISE") == "mlp"   # the second stream runs another network's forward + backward passes
NOISE = os.environ.get("CHECK_NOISE") == "1"          # a second stream kept busy with unrelated GEMMs of varying size
noise_stream = torch.cuda.Stream() if NOISE else None
na = torch.randn(4096, 4096, device="cuda") if NOISE else None
cam_i = ADV % len(tr.cameras)
if MLPNOISE:
    from conftest import pkg as _pkg
    D = _pkg("deform")
    noise_stream = torch.cuda.Stream()
    noise_net = D.DeformModelNormal(is_blender=True, model_name="noise", device=torch.device("cuda:0"), trunk_impl="hip")
    noise_x = torch.randn(P, 3, device="cuda")
    noise_t = torch.tensor([[0.3]], device="cuda").expand(P, -1)
ref = None
bad = {}
for k in range(K):
    if NOISE:
        with torch.cuda.stream(noise_stream):
            for j in range(6 + k % 5):
                n = 512 * (1 + (j + k) % 7)
                nb = na[:n, :n] @ na[:n, :n]
                nb = torch.relu(nb) * 0.5 + na[:n, :n]          # (GEMMs and streaming elementwise kernels)
    for p in tr.params:
        p.grad = None
    if os.environ.get("CHECK_POISON") == "1":  # every free block holds NaN patterns: a stale read shows as NaN, not as 1e-8
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        junk = [torch.full((1 << 30,), 255, dtype=torch.uint8, device="cuda") for _ in range(6)]
        junk += [torch.full((8 << 20,), 255, dtype=torch.uint8, device="cuda") for _ in range(48)]
        junk += [torch.full((1 << 20,), 255, dtype=torch.uint8, device="cuda") for _ in range(64)]
        del junk
    if MLPNOISE:
        torch.cuda.synchronize()
        noise_stream.wait_stream(torch.cuda.current_stream())
    NOISE_FWD = os.environ.get("CHECK_NOISE_AT") == "fwd"
    if MLPNOISE and NOISE_FWD:  # the other stream's passes run beside the FORWARD pass only (joined before the backward pass)
        with torch.cuda.stream(noise_stream):
            for j in range(1 + k % 3):
                o = noise_net.step_raw(noise_x, noise_t)
                o.sum().backward()
    losses, pkg = tr.loss_terms(tr.cameras[cam_i], it)
    if MLPNOISE and NOISE_FWD:
        torch.cuda.current_stream().wait_stream(noise_stream)
    total = None
    for v in losses.values():
        total = v if total is None else total + v
    if MLPNOISE and not NOISE_FWD:  # queued now: runs beside the backward pass below
        with torch.cuda.stream(noise_stream):
            for j in range(1 + k % 3):
                o = noise_net.step_raw(noise_x, noise_t)
                o.sum().backward()
    total.backward()
    torch.cuda.synchronize()
    cur = {"image": pkg["render"].detach().clone(), "radii": pkg["radii"].clone()}
    cur.update({"loss." + n: v.detach().clone() for n, v in losses.items()})
    cur.update({"rast.d_" + n: v.grad.clone() for n, v in _cap.items() if v.grad is not None})
    if pkg.get("viewspace_points") is not None and pkg["viewspace_points"].grad is not None:
        cur["rast.d_mean2D"] = pkg["viewspace_points"].grad.clone()
    cur.update({"grad." + n: (p.grad.clone() if p.grad is not None else torch.zeros(1)) for n, p in zip(names, params)})
    if ref is None:
        ref = cur
        continue
    for n in cur:
        if not torch.equal(cur[n], ref[n]):
            d = (cur[n].float() - ref[n].float()).abs()
            if not torch.isfinite(cur[n].float()).all():
                bad.setdefault(n + " NON-FINITE", []).append((k, int((~torch.isfinite(cur[n].float())).sum())))
            idx = (d > 0).nonzero()[:8].tolist()
            rel = float((d / (ref[n].float().abs() + 1e-30))[d > 0].max())
            bad.setdefault(n, []).append((k, float(d.max()), int((d > 0).sum()), "rel %.2e" % rel, idx))
            if n.startswith("rast.") and len(bad[n]) == 1:
                gids = sorted({i[0] for i in (d > 0).nonzero().tolist()})[:6]
                for gi in gids:
                    print(f"      [{n}] gaussian {gi}: ref {ref[n][gi].tolist()} now {cur[n][gi].tolist()} | radii {int(cur['radii'][gi])}"
                          f" d_mean2D {ref['rast.d_mean2D'][gi].tolist()} d_opacity {ref['rast.d_opacity'][gi].tolist()}")
print(f"P={P} {W}x{Hh}, {K} evaluations: {'all bit-identical' if not bad else 'DIFFERENCES'}")
for n, v in list(bad.items())[:20]:
    print("  ", n, v[:2])
