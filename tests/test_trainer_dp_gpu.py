"""Data-parallel step on the GPU code path ("pack" mode: fresh gradients -> one multi-tensor copy into the flat bucket ->
all-reduce -> one-launch Adam on the bucket views).  Two ranks share cuda:0 here (the test box has one GPU), so the
collective runs over gloo; bench.py uses the same Trainer with backend nccl (= RCCL), one rank per GPU.
Property: replicas stay bit-identical, and DP-2 equals one rank that sums the gradients of the same two frames."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def make_trainer(rank, world, P=3000, W=160, H=128, n_frames=6, seed=0):
    syn, S, D, T = pkg("synthetic"), pkg("scene"), pkg("deform"), pkg("trainer")
    dev = torch.device("cuda:0")
    g_np = syn.make_gaussians(P, seed=seed, kind="aniso", extent=0.7)
    g = S.GaussianModel(sh_degree=3, device=dev)
    g.load_raw(g_np["xyz"], g_np["features_dc"], g_np["features_rest"], g_np["scaling"] + 0.5, g_np["rotation"],
               g_np["opacity"] + 2.0)
    g.active_sh_degree = 3
    cams = [S.TorchCamera(syn.make_camera(W, H, azimuth=0.5 * f, elevation=0.3, fid=f / n_frames), dev,
                          torch.tensor(syn.gt_image(W, H, seed=f), device=dev)) for f in range(n_frames)]
    torch.manual_seed(seed)
    deform = D.DeformModelNormal(is_blender=True, model_name="deform", device=dev, trunk_impl="hip")
    deform_back = D.DeformModelNormal(is_blender=True, model_name="deform_back", device=dev, trunk_impl="hip")
    with torch.no_grad():
        for m in (deform.net, deform_back.net):
            for h in m.head_modules():
                h.weight.mul_(0.05)
                h.bias.mul_(0.05)
    bg = torch.tensor([1.0, 1.0, 1.0], device=dev)
    return T.Trainer(g, deform, deform_back, cams, background=bg, rank=rank, world=world, seed=seed)


def snapshot(tr):
    ps = tr.g.parameters()[:6] + list(tr.deform.net.parameters()) + list(tr.deform_back.net.parameters())
    return [p.detach().cpu().clone() for p in ps]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = make_trainer(rank, world)
    assert tr.pack and tr.multi_adam is not None and tr.bucket is not None
    it = tr.opt.warm_up + 10
    for s in range(2):
        tr.step(it + s)
    torch.cuda.synchronize()
    torch.save(snapshot(tr), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_dp2_pack_mode_on_gpu():
    world = 2
    T = pkg("trainer")
    with tempfile.TemporaryDirectory() as d:
        port = 29700 + (os.getpid() % 2000)
        mp.start_processes(_worker, args=(world, port, d), nprocs=world, join=True, start_method="spawn")
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)                               # replicas stay bit-identical
    # one rank, gradients of the same two frames summed by hand, then the same one-launch Adam
    tr = make_trainer(0, 1)
    it = tr.opt.warm_up + 10
    n = len(tr.cameras)
    for s in range(2):
        tr.g.update_learning_rate(it + s)
        tr.deform.update_learning_rate(it + s)
        tr.deform_back.update_learning_rate(it + s)
        total = None
        for r in range(world):
            for p in tr.params:
                p.grad = None
            cam = tr.cameras[T.frame_schedule(n, s, r, world, 0)]
            losses, _ = tr.loss_terms(cam, it + s)
            sum(losses.values()).backward()
            gs = {id(p): (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in tr.params}
            total = gs if total is None else {k: total[k] + v for k, v in gs.items()}
        tr.multi_adam.step(total)
    for k, (a, b) in enumerate(zip(r0, snapshot(tr))):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), (k, (a - b).abs().max())
    fresh = snapshot(make_trainer(0, 1))
    assert any(not torch.equal(a, b) for a, b in zip(r0, fresh))


@pytest.mark.gpu
def test_lean_step_equals_the_reference_shaped_step():
    """The training loop's lean render (SH tensors unconcatenated through dgm_rasterize_*_split_sh, uninitialised leaf for
    the screen-space gradient, no visibility mask; head weights read in place) against the same Trainer driving render() the
    way the reference calls it (get_features' torch.cat, zeros + 0, radii > 0): after three steps every parameter and the
    densification statistics are bit-equal, and the lean step launches no torch.cat at all.  (The statistics kernel itself is
    checked against the reference's indexing form in test_densify.py.)"""
    S = pkg("scene")
    a = make_trainer(0, 1)
    b = make_trainer(0, 1)
    b.render_fn = lambda *args, **kw: S.render(*args, **kw)  # not `S.render` itself -> Trainer does not ask for lean
    calls = {"cat": 0}
    cat = torch.cat

    def counting_cat(*args, **kw):
        calls["cat"] += 1
        return cat(*args, **kw)

    it = a.opt.warm_up + 10
    for s in range(3):
        torch.cat = counting_cat
        try:
            a.step(it + s)
        finally:
            torch.cat = cat
        b.step(it + s)
    assert calls["cat"] == 0, "the lean step still concatenates something"
    torch.cuda.synchronize()
    for x, y in zip(snapshot(a), snapshot(b)):
        assert torch.equal(x, y)
    assert torch.equal(a.g.max_radii2D, b.g.max_radii2D) and torch.equal(a.g.denom, b.g.denom)
    assert float(a.g.denom.sum()) > 0
    assert torch.equal(a.g.xyz_gradient_accum, b.g.xyz_gradient_accum) and float(a.g.xyz_gradient_accum.sum()) > 0
