"""Fused per-Gaussian glue (dgm_gaussian_apply_*, dgm_cycle_loss_*) against the PyTorch expressions of the reference's
render() prologue (gaussian_renderer/__init__.py:77-95) and cycle loss (train.py:221-238): values and every gradient,
1e-6 relative to each tensor's max; then the whole train step with and without the fused glue."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import pkg


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize("P,ld", [(1, 10), (777, 13), (100_003, 13)])
def test_gaussian_apply_matches_torch(P, ld):
    G = pkg("glue")
    dev = "cuda"
    rng = np.random.RandomState(P)
    mk = lambda *s, scale=1.0: torch.tensor((rng.randn(*s) * scale).astype(np.float32), device=dev, requires_grad=True)
    xyz, scaling, rotation, opacity, delta = mk(P, 3), mk(P, 3, scale=0.7), mk(P, 4), mk(P, 1, scale=2.0), mk(P, ld, scale=0.1)
    with torch.no_grad():
        rotation[0] = 0.0                                       # degenerate quaternion: the clamped-denominator branch
    ref_in = [t.detach().clone().requires_grad_(True) for t in (xyz, scaling, rotation, opacity, delta)]
    rx, rs, rr, ro, rd = ref_in
    ref_out = (rx + rd[:, 0:3], torch.exp(rs) + rd[:, 7:10], F.normalize(rr) + rd[:, 3:7], torch.sigmoid(ro))
    out = G.gaussian_apply(xyz, scaling, rotation, opacity, delta)
    w = [torch.tensor(rng.randn(*o.shape).astype(np.float32), device=dev) for o in out]
    sum((o * wi).sum() for o, wi in zip(out, w)).backward()
    sum((o * wi).sum() for o, wi in zip(ref_out, w)).backward()
    for a, b in zip(out, ref_out):
        assert _rel(a, b) < 1e-6
    for name, a, b in zip(("xyz", "scaling", "rotation", "opacity", "delta"), (xyz, scaling, rotation, opacity, delta), ref_in):
        if name == "rotation":                                  # row 0: both divide by 1e-12
            assert torch.allclose(a.grad[0], b.grad[0], rtol=1e-5)
            if P > 1:
                assert _rel(a.grad[1:], b.grad[1:]) < 1e-5
        else:
            assert _rel(a.grad, b.grad) < 1e-6, name


@pytest.mark.gpu
@pytest.mark.parametrize("N,ld", [(5, 10), (4099, 13), (100_000, 13)])
def test_cycle_loss_matches_torch(N, ld):
    G, S = pkg("glue"), pkg("scene")
    dev = "cuda"
    rng = np.random.RandomState(N)
    a = torch.tensor(rng.randn(N, ld).astype(np.float32), device=dev, requires_grad=True)
    b = torch.tensor(rng.randn(N, ld).astype(np.float32), device=dev, requires_grad=True)
    with torch.no_grad():
        b[0, :10] = -a[0, :10]                                  # exact zeros: sgn(0) = 0
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = (S.l1_loss(-b2[:, 0:3], a2[:, 0:3]) + S.l1_loss(-b2[:, 3:7], a2[:, 3:7]) + S.l1_loss(-b2[:, 7:10], a2[:, 7:10])) / 3.0
    out = G.cycle_loss(a, b)
    assert abs(out.item() - ref.item()) < 2e-6 * abs(ref.item())
    (out * 1.7).backward()
    (ref * 1.7).backward()
    assert _rel(a.grad, a2.grad) < 1e-6 and _rel(b.grad, b2.grad) < 1e-6
    assert float(a.grad[:, 10:].abs().sum()) == 0.0
    out2 = G.cycle_loss(a.detach(), b.detach())
    assert out2.item() == out.item()                            # fixed-order reduction: bit-reproducible


@pytest.mark.gpu
def test_train_step_same_with_and_without_fused_glue():
    """One iteration's losses and every parameter gradient, fused glue vs the PyTorch expressions.  (Parameters after
    several Adam steps are not compared: with eps = 1e-15 Adam turns a gradient of 1e-30 into a full-size update, so
    rounding-level differences in irrelevant gradients show up at the 1e-4 level.)"""
    import test_trainer_dp_gpu as H
    res = []
    for fused in (False, True):
        tr = H.make_trainer(0, 1)
        tr.fused_glue = fused
        it = tr.opt.warm_up + 10
        for p in tr.params:
            p.grad = None
        losses, _ = tr.loss_terms(tr.cameras[2], it)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        res.append(({k: float(v.detach()) for k, v in losses.items()}, [p.grad.detach().clone() for p in tr.params]))
    for k in res[0][0]:
        assert abs(res[0][0][k] - res[1][0][k]) < 1e-5 * abs(res[0][0][k]), k
    for k, (a, b) in enumerate(zip(res[0][1], res[1][1])):
        assert _rel(b, a) < 1e-4, (k, _rel(b, a))


def test_glue_refuses_cpu_tensors():
    """No CPU fallback for the fused glue either (the trainer only selects it for device tensors)."""
    G = pkg("glue")
    z = lambda *s: torch.zeros(*s)
    with pytest.raises(RuntimeError):
        G.gaussian_apply(z(5, 3), z(5, 3), z(5, 4), z(5, 1), z(5, 13))
    with pytest.raises(RuntimeError):
        G.cycle_loss(z(5, 13), z(5, 13))


# ---- 6-DoF heads: screw motion -> rigid transform -> moved centres (dgm_se3_*) ------------------------------------------------------
def _reference_rigid_utils():
    """The reference's own utils/rigid_utils.py, byte-compiled by oracle/build_ref.sh (binaries only; travels to the GPU box)."""
    import importlib.machinery
    import importlib.util
    import os
    from conftest import ROOT
    path = os.path.join(ROOT, "oracle", "_ref", "pyref", "rigid_utils.pyc")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/pyref/rigid_utils.pyc not built (oracle/build_ref.sh needs /root/reference)")
    loader = importlib.machinery.SourcelessFileLoader("ref_rigid_utils", path)
    mod = importlib.util.module_from_spec(importlib.util.spec_from_loader("ref_rigid_utils", loader))
    loader.exec_module(mod)
    return mod


@pytest.mark.gpu
@pytest.mark.parametrize("N,ld", [(1, 6), (1000, 6), (100_003, 16)])
def test_se3_kernels_match_the_reference_functions(N, ld):
    """dgm_se3_exp_* / dgm_se3_transform_* against the REFERENCE's exp_se3 (R/utils/rigid_utils.py:60-83) behind the normalisation of
    R/utils/time_utils.py:116-123 and the 6-DoF branch of render() (R/gaussian_renderer/__init__.py:68-75), evaluated by PyTorch in
    fp64 with autograd: transforms, moved centres and the gradients w.r.t. the raw head outputs and the positions, 2e-5 of each
    tensor's maximum (fp32 kernels vs an fp64 reference; angles from 0.01 to ~6 rad)."""
    G = pkg("glue")
    ref = _reference_rigid_utils()
    dev = "cuda"
    g = torch.Generator().manual_seed(N)
    o = torch.randn(N, ld, generator=g) * torch.pow(10.0, torch.rand(N, 1, generator=g) * 2.5 - 2.0)  # |w| from ~0.01 to ~5
    xyz = torch.randn(N, 3, generator=g)
    wT, wX = torch.randn(N, 4, 4, generator=g), torch.randn(N, 3, generator=g)
    o32, x32 = o.to(dev).requires_grad_(True), xyz.to(dev).requires_grad_(True)
    T = G.se3_exp(o32)
    m = G.se3_transform(T, x32)
    ((T * wT.to(dev)).sum() + (m * wX.to(dev)).sum()).backward()
    o64, x64 = o.double().to(dev).requires_grad_(True), xyz.double().to(dev).requires_grad_(True)
    w, v = o64[:, 0:3], o64[:, 3:6]
    theta = torch.norm(w, dim=-1, keepdim=True)
    T64 = ref.exp_se3(torch.cat([w / theta + 1e-5, v / theta + 1e-5], dim=-1), theta)
    hom = torch.cat([x64, torch.ones_like(x64[:, :1])], -1)
    out = torch.bmm(T64, hom.unsqueeze(-1)).squeeze(-1)
    m64 = out[..., :3] / out[..., 3:]
    ((T64 * wT.double().to(dev)).sum() + (m64 * wX.double().to(dev)).sum()).backward()
    # the same reference functions evaluated by PyTorch in fp32: the yardstick for the gradient w.r.t. (w, v), which is ill-conditioned
    # at small angles (terms ~ 1 / theta^2 cancel) in ANY fp32 evaluation
    of, xf = o.to(dev).requires_grad_(True), xyz.to(dev).requires_grad_(True)
    wf, vf = of[:, 0:3], of[:, 3:6]
    thf = torch.norm(wf, dim=-1, keepdim=True)
    Tf = ref.exp_se3(torch.cat([wf / thf + 1e-5, vf / thf + 1e-5], dim=-1), thf)
    outf = torch.bmm(Tf, torch.cat([xf, torch.ones_like(xf[:, :1])], -1).unsqueeze(-1)).squeeze(-1)
    ((Tf * wT.to(dev)).sum() + (outf[..., :3] / outf[..., 3:] * wX.to(dev)).sum()).backward()
    e_ref = _rel(of.grad.double()[:, :6], o64.grad[:, :6])
    assert _rel(T.detach().double(), T64.detach()) < 2e-5 and _rel(m.detach().double(), m64.detach()) < 2e-5
    e_hip = _rel(o32.grad.double()[:, :6], o64.grad[:, :6])
    print(f"se3 N={N}: d(w, v) error vs fp64, of the tensor's maximum: kernels {e_hip:.2e}, the reference's functions in fp32 {e_ref:.2e}")
    assert e_hip < 2e-5 + 2.0 * e_ref and _rel(x32.grad.double(), x64.grad) < 2e-5
    if ld > 6:
        assert float(o32.grad[:, 6:].abs().max()) == 0.0


@pytest.mark.gpu
def test_render_six_dof_branch_on_the_kernels():
    """scene.render(..., is_6dof=True) with (N, 4, 4) transforms: the fused centre transform equals the reference's cat / bmm / divide
    (image and every gradient), through the rasterizer."""
    S, syn, G = pkg("scene"), pkg("synthetic"), pkg("glue")
    dev = torch.device("cuda")
    P, W, H = 3000, 128, 96
    g_np = syn.make_gaussians(P, seed=3, kind="aniso", extent=0.7)
    grads = {}
    for fused in (True, False):
        g = S.GaussianModel(sh_degree=3, device=dev)
        g.load_raw(g_np["xyz"], g_np["features_dc"], g_np["features_rest"], g_np["scaling"] + 0.5, g_np["rotation"], g_np["opacity"] + 2.0)
        g.active_sh_degree = 3
        cam = S.TorchCamera(syn.make_camera(W, H, azimuth=0.4, elevation=0.3), dev)
        gen = torch.Generator().manual_seed(5)
        o = (0.05 * torch.randn(P, 6, generator=gen)).to(dev).requires_grad_(True)
        T = G.se3_exp(o)
        bg = torch.ones(3, device=dev)
        if fused:
            pkg_ = S.render(cam, g, S.PipelineParams(), bg, T, 0.0, 0.0, is_6dof=True)
        else:
            hom = torch.cat([g.get_xyz, torch.ones_like(g.get_xyz[:, :1])], -1)
            out = torch.bmm(T, hom.unsqueeze(-1)).squeeze(-1)
            means = out[..., :3] / out[..., 3:]
            pkg_ = S.render(cam, g, S.PipelineParams(), bg, means - g.get_xyz, 0.0, 0.0, is_6dof=False)  # means3D = xyz + d_xyz = means
        w = torch.randn(3, H, W, generator=torch.Generator().manual_seed(9)).to(dev)
        (pkg_["render"] * w).sum().backward()
        grads[fused] = (pkg_["render"].detach(), o.grad.clone(), g._xyz.grad.clone())
    for a, b in zip(grads[True], grads[False]):
        assert _rel(a, b) < 1e-5
