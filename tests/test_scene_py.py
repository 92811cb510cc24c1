"""The Python-side options of the reference's render() (R/gaussian_renderer/__init__.py:82-100): precomputed 3D
covariance (pipe.compute_cov3D_python -> get_covariance) and SH -> RGB in Python (pipe.convert_SHs_python -> eval_sh).
CPU: both against the oracle's preprocess (cov3D, rgb).  GPU: a render with each flag equals the default render."""
import math

import numpy as np
import pytest
import torch

from conftest import pkg


def _scene(P=600, W=64, H=48, seed=3):
    syn = pkg("synthetic")
    g = syn.make_gaussians(P, seed=seed, kind="aniso")
    cam = syn.make_camera(W, H, azimuth=0.4, elevation=0.2)
    return syn, g, cam


def test_eval_sh_and_covariance_match_oracle(orc):
    S = pkg("scene")
    syn, g, cam = _scene()
    a = syn.activate(g, 0, 0, 0)
    P = a["means3D"].shape[0]
    f = orc.preprocess_fwd(P, 3, 16, a["means3D"], a["scales"], 1.0, a["rotations"], a["opacities"], a["shs"], None, None,
                           cam.world_view_transform, cam.full_proj_transform, cam.camera_center, 64, 48,
                           math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2))
    vis = f["radii"] > 0
    assert vis.sum() > 100
    sh = torch.tensor(a["shs"]).transpose(1, 2)
    d = torch.tensor(a["means3D"]) - torch.tensor(np.asarray(cam.camera_center)).reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    for deg in (0, 1, 2, 3):
        fd = orc.preprocess_fwd(P, deg, 16, a["means3D"], a["scales"], 1.0, a["rotations"], a["opacities"], a["shs"], None,
                                None, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, 64, 48,
                                math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2))
        rgb = torch.clamp_min(S.eval_sh(deg, sh, d) + 0.5, 0.0).numpy()
        assert np.abs(rgb[vis] - fd["rgb"][vis]).max() < 1e-5, deg
    with pytest.raises(ValueError):
        S.sh_basis(4, d)
    cov = S.covariance_from_scaling_rotation(torch.tensor(a["scales"]), 1.0, torch.tensor(a["rotations"])).numpy()
    assert np.abs(cov[vis] - f["cov3D"][vis]).max() < 1e-5 * np.abs(f["cov3D"][vis]).max()
    cov2 = S.covariance_from_scaling_rotation(torch.tensor(a["scales"]), 1.7, 3.0 * torch.tensor(a["rotations"])).numpy()
    assert np.allclose(cov2, cov * 1.7 ** 2, rtol=1e-4, atol=1e-6)      # modifier scales S, the quaternion is re-normalised


@pytest.mark.gpu
def test_render_python_options_match_default():
    S, syn = pkg("scene"), pkg("synthetic")
    dev = "cuda"
    _, g_np, _ = _scene(P=3000, seed=5)
    pc = S.GaussianModel(sh_degree=3, device=dev)
    pc.load_raw(g_np["xyz"], g_np["features_dc"], g_np["features_rest"], g_np["scaling"], g_np["rotation"], g_np["opacity"])
    pc.active_sh_degree = 3
    cam = S.TorchCamera(syn.make_camera(160, 128, azimuth=0.4, elevation=0.2), dev, torch.zeros(3, 128, 160, device=dev))
    bg = torch.tensor([1.0, 1.0, 1.0], device=dev)
    zero3, zero4 = torch.zeros(3000, 3, device=dev), torch.zeros(3000, 4, device=dev)
    base = S.render(cam, pc, S.PipelineParams(), bg, zero3, zero4, zero3)["render"]
    for flag in ("convert_SHs_python", "compute_cov3D_python"):
        pipe = S.PipelineParams()
        setattr(pipe, flag, True)
        out = S.render(cam, pc, pipe, bg, zero3, zero4, zero3)
        assert out["radii"].shape == (3000,)
        assert (out["render"] - base).abs().max().item() < 2e-5, flag
        out["render"].sum().backward()                                    # gradients flow through the Python path
        assert pc._features_dc.grad is not None and torch.isfinite(pc._scaling.grad).all()
        for p in pc.parameters():
            p.grad = None
