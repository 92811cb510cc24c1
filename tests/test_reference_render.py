"""The REFERENCE's own render() (/root/reference/dgmesh/gaussian_renderer/__init__.py:32-119), executed unmodified -- its
byte-compiled module from oracle/_ref/pyref (oracle/build_ref.sh; binaries only, built where /root/reference exists) -- over
this repository's drop-in packages: `diff_gaussian_rasterization` resolves to the HIP rasterizer, `pc` is this repo's
GaussianModel, the camera and pipeline objects are this repo's.  What "configs/ run unchanged" means for the hot path:
the reference's host code runs as it is and produces what scene.render() produces, gradients included."""
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

PYREF = os.path.join(ROOT, "oracle", "_ref", "pyref")


def _load_pyc(name, fname):
    loader = importlib.machinery.SourcelessFileLoader(name, os.path.join(PYREF, fname))
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    loader.exec_module(mod)
    return mod


@pytest.fixture
def reference_render():
    if not os.path.exists(os.path.join(PYREF, "gaussian_renderer.pyc")):
        pytest.skip("oracle/_ref/pyref not built (needs /root/reference at build time)")
    saved = {k: sys.modules.get(k) for k in ("scene", "scene.gaussian_model", "utils", "utils.sh_utils", "utils.rigid_utils",
                                              "ref_gaussian_renderer")}
    try:
        # the modules the reference file imports: its own two pure-torch helpers, and a name-only stand-in for the class used
        # as a type annotation (scene.gaussian_model imports the whole mesh stack)
        sys.modules["scene"] = types.ModuleType("scene")
        gm = types.ModuleType("scene.gaussian_model")
        gm.GaussianModel = object
        sys.modules["scene.gaussian_model"] = gm
        sys.modules["utils"] = types.ModuleType("utils")
        _load_pyc("utils.sh_utils", "sh_utils.pyc")
        _load_pyc("utils.rigid_utils", "rigid_utils.pyc")
        mod = _load_pyc("ref_gaussian_renderer", "gaussian_renderer.pyc")
        import diff_gaussian_rasterization
        assert mod.GaussianRasterizer is diff_gaussian_rasterization.GaussianRasterizer  # the drop-in package was picked up
        yield mod.render
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _scene(P=20000, W=320, H=240, seed=3):
    syn, S = pkg("synthetic"), pkg("scene")
    dev = torch.device("cuda")
    g = syn.make_gaussians(P, seed=seed, kind="aniso")
    pc = S.GaussianModel(sh_degree=3, device=dev)
    rng = np.random.RandomState(seed)
    pc.load_raw(g["xyz"], g["features_dc"], g["features_rest"], g["scaling"], g["rotation"], g["opacity"], rng.randn(P, 3))
    pc.active_sh_degree = 3
    cam = S.TorchCamera(syn.make_camera(W, H, azimuth=0.4, elevation=0.2), dev)
    d = [torch.tensor((rng.randn(P, k) * 0.01).astype(np.float32), device=dev, requires_grad=True) for k in (3, 4, 3)]
    return pc, cam, d


@pytest.mark.gpu
@pytest.mark.parametrize("convert_shs,cov_python", [(False, False), (True, False), (False, True)])
def test_reference_render_runs_unmodified_over_the_drop_in_packages(reference_render, convert_shs, cov_python):
    S = pkg("scene")
    pc, cam, (d_xyz, d_rot, d_scale) = _scene()
    pipe = S.PipelineParams()
    pipe.convert_SHs_python, pipe.compute_cov3D_python, pipe.debug = convert_shs, cov_python, False
    bg = torch.tensor([1.0, 1.0, 1.0], device="cuda")
    dL = torch.randn(3, cam.image_height, cam.image_width, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    outs = []
    for fn in (reference_render, S.render):
        for p in pc.parameters() + [d_xyz, d_rot, d_scale]:
            p.grad = None
        pkg_ = fn(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
        (pkg_["render"] * dL).sum().backward()
        outs.append((pkg_["render"].detach().clone(), pkg_["radii"].clone(), pkg_["visibility_filter"].clone(),
                     pkg_["viewspace_points"].grad.clone(), [p.grad.clone() if p.grad is not None else None
                                                             for p in pc.parameters() + [d_xyz, d_rot, d_scale]]))
    (img_r, rad_r, vis_r, vsp_r, gr_r), (img_o, rad_o, vis_o, vsp_o, gr_o) = outs
    assert img_r.shape == (3, cam.image_height, cam.image_width) and float(img_r.std()) > 0.01
    assert torch.equal(rad_r, rad_o) and torch.equal(vis_r, vis_o) and int(vis_r.sum()) > 1000
    # same kernels, same inputs: equal up to the order of the torch ops in front of the rasterizer (identical here)
    assert float((img_r - img_o).abs().max()) <= 1e-6
    assert float((vsp_r - vsp_o).abs().max()) <= 1e-6 * (1 + float(vsp_r.abs().max()))
    for a, b in zip(gr_r, gr_o):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 1e-5 * (1e-12 + float(a.abs().max()))
