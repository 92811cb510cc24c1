"""'configs/ run unchanged' (BASELINE.json north_star): the REFERENCE's own configuration objects drive this repo's Trainer.

The reference builds its per-run parameters in train.py:858-906 (R/ = /root/reference/dgmesh/): argparse groups
`ModelParams`, `OptimizationParams`, `PipelineParams` (R/arguments/__init__.py:48-154), the YAML file merged over the
parsed defaults (`load_config_from_file`, `merge_config`, R/utils/system_utils.py:33-51), then `extract()`.  This test
executes exactly that -- the two reference modules byte-compiled by oracle/build_ref.sh into oracle/_ref/pyref (binaries
only), the shipped YAML files copied next to them -- and hands the resulting objects, untouched, to `Trainer`: a Gaussian-phase
step (warm_up <= it < dpsr_iter) and a mesh-phase step (it >= dpsr_iter + NORMAL_WARMUP_ITER, R/train.py:127) must run and
move every network.  CPU path: oracle-backed test render, PyTorch trunks (same harness as tests/test_trainer_dp.py).
"""
import glob
import importlib.machinery
import importlib.util
import os
import sys
from argparse import ArgumentParser, Namespace

import pytest
import torch

from conftest import ROOT, pkg

REF = os.path.join(ROOT, "oracle", "_ref")
PYREF = os.path.join(REF, "pyref")
NAMED = ["d-nerf/jumpingjacks.yaml", "dg-mesh/beagle.yaml", "nerfies/toby-sit.yaml", "neural-actor/D2_vlad.yaml"]


def _load_pyc(name, fname):
    loader = importlib.machinery.SourcelessFileLoader(name, os.path.join(PYREF, fname))
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def _all_configs():
    found = sorted(os.path.relpath(p, os.path.join(REF, "configs")) for p in glob.glob(os.path.join(REF, "configs", "*", "*.yaml")))
    return found or NAMED


def reference_params(config):
    """lp, op, pp exactly as R/train.py:858-906 makes them (no command-line flags besides --config)."""
    if not os.path.exists(os.path.join(PYREF, "arguments.pyc")) or not os.path.isdir(os.path.join(REF, "configs")):
        pytest.skip("oracle/_ref/pyref/arguments.pyc / oracle/_ref/configs not built (needs /root/reference at build time)")
    A = _load_pyc("ref_arguments", "arguments.pyc")
    U = _load_pyc("ref_system_utils", "system_utils.pyc")
    parser = ArgumentParser(description="Training script parameters")
    lp, op, pp = A.ModelParams(parser), A.OptimizationParams(parser), A.PipelineParams(parser)
    args = parser.parse_args([])
    path = os.path.join(REF, "configs", config)
    assert os.path.exists(path), path
    args = Namespace(**U.merge_config(U.load_config_from_file(path), args))
    return lp.extract(args), op.extract(args), pp.extract(args)


def make_trainer(lp, op, pp, mesh, seed=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle_render
    import test_trainer_dp as H

    D, T = pkg("deform"), pkg("trainer")
    base = H.make_trainer(0, 1, seed=seed, n_frames=3, P=96, W=32, H=32)  # scene, cameras, the two position networks
    nets = {}
    torch.manual_seed(seed + 7)
    for name in ("deform", "deform_back"):
        nets[name] = D.DeformModelNormal(is_blender=lp.is_blender, is_6dof=lp.is_6dof, model_name=name, device="cpu", trunk_impl="torch")
    ms = None
    if mesh:
        extra = [D.DeformModelNormalSep(is_blender=lp.is_blender, model_name="deform_normal", device="cpu", trunk_impl="torch"),
                 D.DeformModelNormalSep(is_blender=lp.is_blender, model_name="deform_back_normal", device="cpu", trunk_impl="torch"),
                 D.AppearanceModel(is_blender=lp.is_blender, device="cpu", trunk_impl="torch")]
        with torch.no_grad():  # (zero-initialised heads, time_utils.py:248-249: give the cycle loss something to act on)
            for m in extra[:2]:
                torch.nn.init.normal_(m.net.gaussian_normal.weight, std=0.02)
        ms = T.MeshPhase(*extra, dpsr=None, n_verts=32, device="cpu", seed=seed, density_thres=op.init_density_threshold,
                         center=lp.gaussian_center)
    bg = torch.tensor([1.0, 1.0, 1.0] if lp.white_background else [0.0, 0.0, 0.0])
    return T.Trainer(base.g, nets["deform"], nets["deform_back"], base.cameras, opt=op, pipe=pp, background=bg,
                     is_blender=lp.is_blender, is_6dof=lp.is_6dof, render_fn=_oracle_render.render, fused_adam=False,
                     prune_threshold=lp.prune_threshold, white_background=lp.white_background, mesh=ms)


def _moved(before, after):
    return [not torch.equal(a, b) for a, b in zip(before, after)]


@pytest.mark.parametrize("config", _all_configs())
def test_reference_config_drives_both_phases(config):
    lp, op, pp = reference_params(config)
    # the object is the reference's: none of this repo's additions on it
    assert type(op).__name__ == "GroupParams" and not hasattr(op, "normal_deform_delay")
    T = pkg("trainer")
    # ---- Gaussian phase: deformation MLP on, before the mesh phase
    tr = make_trainer(lp, op, pp, mesh=False)
    assert tr.opt is op and tr.pipe is pp
    before = [p.detach().clone() for p in tr.params]
    it = op.warm_up + 10
    assert it < op.dpsr_iter
    loss, _ = tr.step(it)
    assert torch.isfinite(loss)
    mv = _moved(before, [p.detach() for p in tr.params])
    n_g = 6
    assert all(mv[:n_g]), "a Gaussian tensor did not step"
    assert any(mv[n_g:]), "no network weight stepped"
    # the schedules read the config's numbers (exponential decay to position_lr_final at position_lr_max_steps, scaled by
    # the scene extent -- gaussian_model_dpsr_dynamic_anchor.py:186-229)
    tr.g.update_learning_rate(op.position_lr_max_steps)
    lr_xyz = [g["lr"] for g in tr.g.optimizer.param_groups if g["name"] == "xyz"][0]
    assert abs(lr_xyz - op.position_lr_final * tr.g.spatial_lr_scale) <= 1e-12 + 1e-6 * lr_xyz

    # ---- mesh co-training phase with every network on
    tr = make_trainer(lp, op, pp, mesh=True)
    before = [p.detach().clone() for p in tr.params]
    it = op.dpsr_iter + T.NORMAL_WARMUP_ITER + max(op.normal_warm_up, 0) + 10
    loss, _ = tr.step(it)
    assert torch.isfinite(loss)
    mv = _moved(before, [p.detach() for p in tr.params])
    off = n_g
    for m in [tr.deform, tr.deform_back, tr.mesh.deform_normal, tr.mesh.deform_back_normal]:
        n = len([p for p in m.net.parameters() if p.requires_grad])
        assert any(mv[off:off + n]), f"{m.model_name} did not step"
        off += n
    assert len(tr.optimizers) == 6


def test_named_baseline_configs_are_covered():
    """BASELINE.json's configs name these four files; the parametrised test above must have seen them."""
    if not os.path.isdir(os.path.join(REF, "configs")):
        pytest.skip("oracle/_ref/configs not built")
    have = set(_all_configs())
    for n in NAMED:
        assert n in have, n
    lp, op, _ = reference_params("d-nerf/jumpingjacks.yaml")
    assert (op.warm_up, op.dpsr_iter, op.iterations, lp.grid_res, lp.is_blender) == (3000, 10000, 25000, 288, True)
