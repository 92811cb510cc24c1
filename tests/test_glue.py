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
