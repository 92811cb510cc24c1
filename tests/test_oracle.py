"""Pins the CPU oracle (oracle/dgr_oracle.c, oracle/knn_oracle.c).

The reference ships no tests or golden vectors for this path, so the oracle
is pinned here by: analytic known-answer tests, structural invariants of the binning, a
torch.autograd cross-check of every hand-written gradient, and the committed golden fixtures
(tests/golden/, produced by tests/golden/make_golden.py).
"""
import math

import numpy as np
import pytest
import torch

from conftest import oracle_backward, oracle_forward, raster_args
import _torch_ref as tr


def _single(syn, W=64, H=64, scale=0.05, opacity=0.8, pos=(0.0, 0.0, 0.0), quat=(1, 0, 0, 0)):
    cam = syn.make_camera(W, H, azimuth=0.0, elevation=0.0, radius=4.0)
    shs = np.zeros((1, 16, 3), np.float32)
    shs[0, 0] = (np.array([0.2, 0.5, 0.9]) - 0.5) / syn.C0
    return dict(
        bg=np.zeros(3, np.float32), means3D=np.array([pos], np.float32), colors_precomp=None,
        opacities=np.array([[opacity]], np.float32), scales=np.full((1, 3), scale, np.float32),
        rotations=np.array([quat], np.float32), scale_modifier=1.0, cov3D_precomp=None,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), H=H, W=W, sh=shs, degree=3,
        campos=cam.camera_center), cam


def test_kat_single_isotropic_gaussian(orc, syn):
    """Closed form: sigma_px^2 = (f*s/z)^2 + 0.3, radius = ceil(3*sqrt(lambda_max)), centre pixel alpha."""
    a, cam = _single(syn)
    f = oracle_forward(orc, a)
    W = H = 64
    z = 4.0
    focal = W / (2 * a["tanfovx"])
    var = (focal * 0.05 / z) ** 2 + 0.3
    g = f["geom"]
    assert g["radii"][0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))  # lambda = mid + sqrt(max(.1, 0))
    np.testing.assert_allclose(g["depths"][0], z, rtol=1e-6)
    np.testing.assert_allclose(g["means2D"][0], [(W - 1) / 2, (H - 1) / 2], atol=1e-4)
    np.testing.assert_allclose(g["conic_opacity"][0], [1 / var, 0, 1 / var, 0.8], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(g["cov3D"][0], [0.0025, 0, 0, 0.0025, 0, 0.0025], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(g["rgb"][0], [0.2, 0.5, 0.9], rtol=1e-6)
    # pixel (31,31): d = (0.5, 0.5)
    alpha = 0.8 * math.exp(-0.5 * (0.25 + 0.25) / var)
    np.testing.assert_allclose(f["color"][:, 31, 31], np.array([0.2, 0.5, 0.9]) * alpha, rtol=1e-5)
    np.testing.assert_allclose(f["img"]["final_T"][31, 31], 1 - alpha, rtol=1e-5)
    # rect: centre 31.5 +- r, tiles of 16
    r = g["radii"][0]
    tmin, tmax = int((31.5 - r) / 16), int((31.5 + r + 15) / 16)
    assert g["tiles_touched"][0] == (tmax - tmin) ** 2
    assert f["num_rendered"] == (tmax - tmin) ** 2


def test_kat_sh_one_hot(orc, syn):
    """One-hot SH coefficient k -> rgb = basis_k(dir) + 0.5 with the constants of auxiliary.h:22-39."""
    a, cam = _single(syn, pos=(0.3, -0.2, 0.4))
    d = a["means3D"][0].astype(np.float64) - cam.camera_center.astype(np.float64)
    d /= np.linalg.norm(d)
    B = tr.sh_basis(3, torch.tensor(d[None])).numpy()[0]
    for k in range(16):
        shs = np.zeros((1, 16, 3), np.float32)
        shs[0, k, 1] = 0.5
        a["sh"] = shs
        f = oracle_forward(orc, a)
        exp = 0.5 * B[k] + 0.5
        np.testing.assert_allclose(f["geom"]["rgb"][0], [0.5, max(exp, 0), 0.5], rtol=2e-6, atol=1e-7)
        assert f["geom"]["clamped"][0, 1] == (exp < 0)


def test_kat_culling_and_empty(orc, syn):
    a, cam = _single(syn)
    # behind the near plane (view z <= 0.2): camera at x=4 looking towards -x
    a["means3D"] = np.array([[3.9, 0, 0]], np.float32)
    f = oracle_forward(orc, a)
    assert f["radii"][0] == 0 and f["num_rendered"] == 0
    assert np.all(f["color"] == 0)  # bg = 0, T = 1
    assert not orc.mark_visible(a["means3D"], a["viewmatrix"], a["projmatrix"])[0]
    # P == 0 (DGR/rasterize_points.cu:81)
    a["means3D"] = np.zeros((0, 3), np.float32)
    f = oracle_forward(orc, a)
    assert f["num_rendered"] == 0 and f["color"].shape == (3, 64, 64)


def test_get_higher_msb(orc):
    for n, want in [(1, 1), (2, 2), (625, 10), (2500, 12), (4096, 13), (8160, 13)]:
        assert orc.lib().orc_get_higher_msb(n) == want


@pytest.mark.parametrize("kind,W,H", [("init", 200, 136), ("aniso", 123, 77), ("trained", 160, 160)])
def test_binning_invariants(orc, syn, kind, W, H):
    a = raster_args(syn, 3000, W, H, seed=3, kind=kind)
    f = oracle_forward(orc, a)
    g, b, img = f["geom"], f["binning"], f["img"]
    R = f["num_rendered"]
    assert R == int(g["tiles_touched"].sum()) == len(b["point_list"])
    assert np.all(np.diff(b["keys"].astype(np.uint64)) >= 0)  # sorted
    tiles = b["ranges"].shape[0]
    # ranges partition [0,R) in tile order; empty tiles are (0,0)
    pos = 0
    for t in range(tiles):
        r0, r1 = b["ranges"][t]
        if r0 == r1 == 0 and not np.any((b["keys"] >> np.uint64(32)) == t):
            continue
        assert r0 == pos and r1 > r0
        assert np.all((b["keys"][r0:r1] >> np.uint64(32)) == t)
        # ties on depth resolve by ascending Gaussian index (stable sort of idx-ordered emission)
        k = b["keys"][r0:r1]
        same = k[1:] == k[:-1]
        assert np.all(b["point_list"][r0:r1][1:][same] > b["point_list"][r0:r1][:-1][same])
        pos = r1
    assert pos == R
    # depth bits of each instance equal the Gaussian's depth
    dbits = g["depths"].view(np.uint32)[b["point_list"]]
    assert np.all((b["keys"] & np.uint64(0xFFFFFFFF)) == dbits)
    # per-pixel: n_contrib <= tile range length; out = C + T*bg consistency via bg linearity
    gx = (W + 15) // 16
    yy, xx = np.mgrid[0:H, 0:W]
    tl = (yy // 16) * gx + xx // 16
    assert np.all(img["n_contrib"] <= (b["ranges"][tl, 1] - b["ranges"][tl, 0]))
    a2 = dict(a)
    a2["bg"] = np.zeros(3, np.float32)
    f2 = oracle_forward(orc, a2)
    np.testing.assert_allclose(f["color"], f2["color"] + img["final_T"][None] * a["bg"][:, None, None], atol=2e-7)
    assert np.all((img["final_T"] >= 0) & (img["final_T"] <= 1))


def test_gradients_match_autograd(orc, syn):
    """Every hand-written gradient of the oracle vs torch.autograd through the dense fp64 restatement."""
    W = H = 48
    P = 96
    cam = syn.make_camera(W, H, azimuth=0.7, elevation=0.4, radius=4.0, fovx=0.5)
    a = raster_args(syn, P, W, H, seed=5, kind="aniso", cam=cam, extent=0.6, bg=(0.3, 0.6, 0.1))
    a["scales"] = (a["scales"] * 1.5).astype(np.float32)
    f = oracle_forward(orc, a)
    assert (f["radii"] > 0).sum() > 50
    rng = np.random.RandomState(0)
    dL = rng.randn(3, H, W).astype(np.float32)
    g = oracle_backward(orc, f, a, dL)

    T = lambda v: torch.tensor(np.asarray(v, np.float64), requires_grad=True)
    means3D, scales, rots, opac, shs = map(T, (a["means3D"], a["scales"], a["rotations"], a["opacities"], a["sh"]))
    vm = torch.tensor(a["viewmatrix"].astype(np.float64))
    pm = torch.tensor(a["projmatrix"].astype(np.float64))
    campos = torch.tensor(a["campos"].astype(np.float64))
    pix, ndc, conic, rgb, _ = tr.preprocess(means3D, scales, rots, opac, shs, vm, pm, campos, W, H, a["tanfovx"],
                                            a["tanfovy"], 3)
    # fov clamp must be inactive for the autograd comparison (see module docstring of _torch_ref)
    tv = (torch.cat([means3D, torch.ones(P, 1, dtype=torch.float64)], 1) @ vm).detach().numpy()
    assert np.all(np.abs(tv[:, 0] / tv[:, 2]) < 1.3 * a["tanfovx"]) and np.all(np.abs(tv[:, 1] / tv[:, 2]) < 1.3 * a["tanfovy"])
    b = f["binning"]
    img, ncon = tr.render(pix, conic, opac[:, 0], rgb, torch.tensor(a["bg"].astype(np.float64)), b["ranges"],
                          b["point_list"], W, H)
    np.testing.assert_allclose(img.detach().numpy(), f["color"], atol=2e-5)
    ok = f["img"]["fragile"] == 0
    assert np.all(ncon.numpy()[ok] == f["img"]["n_contrib"][ok])
    (img * torch.tensor(dL.astype(np.float64))).sum().backward()
    vis = f["radii"] > 0

    def close(mine, ref, name):
        ref = ref.numpy()
        scale = np.abs(ref).max() + 1e-30
        err = np.abs(mine - ref).max() / scale
        assert err < 2e-4, f"{name}: rel-to-max err {err:.3e}"

    close(g["dL_dmeans3D"], means3D.grad, "means3D")
    close(g["dL_dscales"], scales.grad, "scales")
    close(g["dL_drotations"], rots.grad, "rotations")
    close(g["dL_dopacity"], opac.grad, "opacity")
    close(g["dL_dsh"], shs.grad, "sh")
    close(g["dL_dmeans2D"][:, :2], ndc.grad, "means2D (NDC units)")
    assert np.all(g["dL_dmeans3D"][~vis] == 0) and np.all(g["dL_dsh"][~vis] == 0)


def test_fov_clamp_gradient_gate(orc, syn):
    """backward.cu:175-176, 262-263: outside +-1.3*tan(fov) the x/y gradient of the covariance path is
    gated to zero while dL_dtz keeps using the clamped t (not an autograd identity -> KAT)."""
    W = H = 64
    a, cam = _single(syn, W, H, scale=0.3, opacity=0.9, pos=(0.0, 2.2, 0.0))
    a["sh"][0, 1:] = 0
    f = oracle_forward(orc, a)
    assert f["radii"][0] > 0 and f["num_rendered"] > 0
    tv = np.append(a["means3D"][0], 1) @ a["viewmatrix"]
    assert abs(tv[0] / tv[2]) > 1.3 * a["tanfovx"]
    dL = np.random.RandomState(1).randn(3, H, W).astype(np.float32)
    g = oracle_backward(orc, f, a, dL)
    # recompute the covariance-path mean gradient alone: set dL_dmean2D = 0 and SH off
    geom = f["geom"]
    P = 1
    dmeans = np.zeros((P, 3), np.float32)
    dcov = np.zeros((P, 6), np.float32)
    import ctypes

    fx = np.float32(W) / (np.float32(2) * np.float32(a["tanfovx"]))
    orc.lib().orc_cov2d_bwd(ctypes.c_int(P), orc._ptr(a["means3D"]), orc._ptr(geom["radii"]), orc._ptr(geom["cov3D"]),
                            ctypes.c_float(fx), ctypes.c_float(fx), ctypes.c_float(a["tanfovx"]),
                            ctypes.c_float(a["tanfovy"]), orc._ptr(a["viewmatrix"]), orc._ptr(g["dL_dconic"]),
                            orc._ptr(dmeans), orc._ptr(dcov))
    # dL_dt = Rwc * dL_dmean ; its x component must be exactly zero (gated)
    Rwc = a["viewmatrix"][:3, :3].T
    dt = Rwc @ dmeans[0]
    assert abs(dt[0]) <= 1e-6 * (np.abs(dt).max() + 1e-30)
    assert np.abs(dt[2]) > 0


def test_knn_matches_definition(orc, syn):
    rng = np.random.RandomState(0)
    for P in (1500, 5000):
        pts = ((rng.rand(P, 3) * 2 - 1) * 1.3).astype(np.float32)
        pts[: P // 10] = pts[P // 10: 2 * (P // 10)]  # exact duplicates -> zero distances
        a = orc.knn(pts)
        b = orc.knn(pts, brute=True)
        assert np.array_equal(a, b)
    # degenerate axis (all x equal) must not crash and still be exact
    pts = ((rng.rand(2000, 3) * 2 - 1)).astype(np.float32)
    pts[:, 0] = 0.25
    assert np.array_equal(orc.knn(pts), orc.knn(pts, brute=True))
    # against an independent fp64 KD-tree (tolerance: fp32 rounding of d2)
    ref = syn.brute_knn_dist2(pts)
    np.testing.assert_allclose(orc.knn(pts), ref, rtol=1e-5)
