"""DPSR on the device (dg-mesh_amd/dpsr.py, csrc/dpsr.hip + rocFFT through torch.fft) against goldens produced by the
REFERENCE's own code (tests/golden/make_golden.py::dpsr_golden executes point_rasterize / DPSR.forward from
/root/reference/dgmesh/nvdiffrast_utils/{dpsr_utils,dpsr}.py on the CPU): the rasterised normal field, the indicator
grid phi and the gradients of a weighted sum of phi w.r.t. points and normals.  Tolerance 2e-4 of each tensor's maximum:
fp32 FFTs of two libraries and atomic summation order are the only differences."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

GOLD = os.path.join(ROOT, "tests", "golden", "dpsr_small.npz")


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.gpu
def test_dpsr_matches_the_reference_code():
    D = pkg("dpsr")
    g = np.load(GOLD)
    res, sig = int(g["res"]), float(g["sig"])
    V = torch.tensor(g["V"], device="cuda", requires_grad=True)
    N = torch.tensor(g["N"], device="cuda", requires_grad=True)
    ras = D.point_rasterize(V.detach().unsqueeze(0), N.detach().unsqueeze(0), (res, res, res))[0].cpu().numpy()
    assert rel(ras[:, ::2, ::2, ::2], g["raster_sub"]) < 1e-5
    assert abs(np.abs(ras).sum() - float(g["raster_abs_sum"])) < 1e-4 * float(g["raster_abs_sum"])
    phi = D.DPSR(res=(res, res, res), sig=sig)(V.unsqueeze(0), N.unsqueeze(0))
    assert phi.shape == (1, res, res, res)
    assert rel(phi[0].detach().cpu().numpy(), g["phi"]) < 2e-4
    w = torch.tensor(np.random.RandomState(int(g["weight_seed"])).randn(1, res, res, res).astype(np.float32), device="cuda")
    (phi * w).sum().backward()
    assert rel(N.grad.cpu().numpy(), g["dN"]) < 2e-4
    assert rel(V.grad.cpu().numpy(), g["dV"]) < 2e-4


@pytest.mark.gpu
def test_dpsr_properties_at_training_resolution():
    """res 128 (the reference's default grid): the field of an oriented sphere is negative inside / positive outside after
    the reference's normalisation (|phi(0,0,0)| -> 0.5 at the far corner), zero on average at the samples, and the autograd
    gradient w.r.t. the normals matches a finite difference of the (linear-in-N before normalisation) pipeline."""
    D = pkg("dpsr")
    rng = np.random.RandomState(0)
    n, res = 50000, 128
    d = rng.randn(n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    V = torch.tensor((0.5 + 0.3 * d).astype(np.float32), device="cuda")
    N = torch.tensor(d.astype(np.float32), device="cuda", requires_grad=True)
    dpsr = D.DPSR(res=(res, res, res), sig=2.0)
    phi = dpsr(V.unsqueeze(0), N.unsqueeze(0))[0]
    c = res // 2
    pd = phi.detach()
    assert float(pd[c, c, c]) * float(pd[0, 0, 0]) < 0 and abs(abs(float(pd[0, 0, 0])) - 0.5) < 1e-5
    fv = D.grid_interp(phi.detach().unsqueeze(0).unsqueeze(-1), V.unsqueeze(0))[0, :, 0]
    assert abs(float(fv.detach().mean())) < 2e-3 * float(phi.detach().abs().max())
    w = torch.randn(res, res, res, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    (phi * w).sum().backward()
    dirn = torch.randn(n, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    eps = 1e-2
    with torch.no_grad():
        lp = (dpsr(V.unsqueeze(0), (N + eps * dirn).unsqueeze(0))[0] * w).sum().double()
        lm = (dpsr(V.unsqueeze(0), (N - eps * dirn).unsqueeze(0))[0] * w).sum().double()
    fd = float((lp - lm) / (2 * eps))
    an = float((N.grad * dirn).sum())
    assert abs(fd - an) < 2e-2 * max(abs(fd), abs(an)) + 1e-3, (fd, an)


def test_laplace_regularizer_matches_reference_formula():
    D = pkg("dpsr")
    v = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], requires_grad=True)
    f = torch.tensor([[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]])
    loss = D.laplace_regularizer_const(v, f)
    # umbrella operator by hand: every vertex of the tetrahedron has 3 neighbours, each counted twice (norm = 6)
    nb = {0: [1, 2, 3], 1: [0, 2, 3], 2: [0, 1, 3], 3: [0, 1, 2]}
    want = torch.stack([sum(2 * (v[j] - v[i]) for j in nb[i]) / 6.0 for i in range(4)]).pow(2).mean()
    assert abs(float(loss) - float(want)) < 1e-7
    loss.backward()
    assert v.grad is not None and torch.isfinite(v.grad).all()


@pytest.mark.gpu
def test_laplace_regularizer_hip_kernels():
    """dgm_laplace_forward / backward against the reference's index-op formulation (regularizer.py:40-60) evaluated in fp64 on the same
    mesh: a closed triangulated grid-sphere with 20 k vertices plus a few isolated vertices (norm clamps at 1) -- loss and the
    gradient w.r.t. every vertex."""
    D = pkg("dpsr")
    dev = "cuda"
    n = 100
    th, ph = np.meshgrid(np.linspace(0.1, np.pi - 0.1, n), np.linspace(0, 2 * np.pi, 2 * n, endpoint=False), indexing="ij")
    v = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], -1).reshape(-1, 3)
    v = v + 0.01 * np.random.RandomState(0).randn(*v.shape)
    idx = np.arange(n * 2 * n).reshape(n, 2 * n)
    a, b, c, d = idx[:-1, :], np.roll(idx, -1, 1)[:-1, :], idx[1:, :], np.roll(idx, -1, 1)[1:, :]
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([b, d, c], -1).reshape(-1, 3)], 0)
    v = np.concatenate([v, np.random.RandomState(1).randn(5, 3)], 0)  # isolated vertices: no face, term 0
    vt = torch.tensor(v.astype(np.float32), device=dev, requires_grad=True)
    ft = torch.tensor(faces.astype(np.int64), device=dev)
    loss = D.laplace_regularizer_const(vt, ft)
    (loss * 3.0).backward()
    v64 = torch.tensor(v, dtype=torch.float64, device=dev, requires_grad=True)
    want = D._laplace_regularizer_torch(v64, ft)
    (want * 3.0).backward()
    assert abs(float(loss.detach()) - float(want.detach())) <= 1e-5 * abs(float(want.detach()))
    err = (vt.grad.double() - v64.grad).abs().max().item() / v64.grad.abs().max().item()
    assert err < 1e-5, err
    assert float(vt.grad[-5:].abs().max()) == 0.0
