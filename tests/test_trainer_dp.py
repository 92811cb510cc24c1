"""Frame-parallel data parallelism of the train step, on CPU with the gloo backend (world_size 2).

Property (dg-mesh_amd/trainer.py): W ranks x 1 frame per step, flat-bucket all-reduce(SUM), identical Adam update
== 1 rank that accumulates the gradients of the same W frames and then steps.  The rasterizer used here is the
oracle-backed test render (tests/_oracle_render.py); the MLPs run on PyTorch-CPU.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def make_trainer(rank, world, seed=0, n_frames=6, P=160, W=48, H=32):
    syn, S, D, T = pkg("synthetic"), pkg("scene"), pkg("deform"), pkg("trainer")
    import _oracle_render

    dev = "cpu"
    g_np = syn.make_gaussians(P, seed=seed, kind="aniso", extent=0.7)
    g = S.GaussianModel(sh_degree=3, device=dev)
    g.load_raw(g_np["xyz"], g_np["features_dc"], g_np["features_rest"], g_np["scaling"] + 0.5, g_np["rotation"],
               g_np["opacity"] + 2.0)
    g.active_sh_degree = 3
    cams = [S.TorchCamera(syn.make_camera(W, H, azimuth=0.5 * f, elevation=0.3, fid=f / n_frames), dev,
                          syn.gt_image(W, H, seed=f)) for f in range(n_frames)]
    torch.manual_seed(seed)
    deform = D.DeformModelNormal(is_blender=True, model_name="deform", device=dev, trunk_impl="torch")
    deform_back = D.DeformModelNormal(is_blender=True, model_name="deform_back", device=dev, trunk_impl="torch")
    with torch.no_grad():
        for m in (deform.net, deform_back.net):
            for h in m.head_modules():
                h.weight.mul_(0.05)
                h.bias.mul_(0.05)
    bg = torch.tensor([1.0, 1.0, 1.0])
    return T.Trainer(g, deform, deform_back, cams, background=bg, rank=rank, world=world, seed=seed,
                     render_fn=_oracle_render.render, fused_adam=False)


def snapshot(tr):
    ps = tr.g.parameters()[:6] + list(tr.deform.net.parameters()) + list(tr.deform_back.net.parameters())
    return [p.detach().clone() for p in ps]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = make_trainer(rank, world)
    it = tr.opt.warm_up + 10
    for s in range(2):
        tr.step(it + s)
    torch.save(snapshot(tr), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_schedule_partitions_epoch():
    T = pkg("trainer")
    n, W = 12, 4
    seen = [T.frame_schedule(n, s, r, W, seed=3) for s in range(n // W) for r in range(W)]
    assert sorted(seen) == list(range(n))                      # one epoch covers every frame exactly once
    assert [T.frame_schedule(n, s, 1, W, seed=3) for s in range(3)] == seen[1::W]
    assert T.frame_schedule(n, 0, 0, W, seed=3) == T.frame_schedule(n, 0, 0, W, seed=3)
    # the per-epoch permutation is cached (one entry): alternating epochs / seeds / frame counts must give the same answers
    # as asking for them in order
    want = {(e, sd, m): T.frame_schedule(m, e * (m // W), 0, W, seed=sd) for e in (0, 1, 2) for sd in (3, 4) for m in (n, n + 8)}
    for _ in range(2):
        for key in reversed(sorted(want)):
            e, sd, m = key
            assert T.frame_schedule(m, e * (m // W), 0, W, seed=sd) == want[key]


def test_flat_bucket_views():
    T = pkg("trainer")
    a, b = torch.nn.Parameter(torch.zeros(3, 2)), torch.nn.Parameter(torch.zeros(5))
    bk = T.FlatGradBucket([a, b])
    (a.sum() * 2 + (b * torch.arange(5.0)).sum()).backward()
    assert bk.flat.tolist() == [2.0] * 6 + [0.0, 1.0, 2.0, 3.0, 4.0]
    assert a.grad.data_ptr() == bk.flat.data_ptr()            # gradients accumulate in place inside the bucket
    bk.zero()
    assert float(b.grad.abs().sum()) == 0.0


def test_dp2_equals_single_rank_accumulation():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000)
        mp.start_processes(_worker, args=(world, port, d), nprocs=world, join=True, start_method="spawn")
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)                               # replicas stay bit-identical
    # single process: accumulate the same two frames per step, then step
    T = pkg("trainer")
    tr = make_trainer(0, 1)
    it = tr.opt.warm_up + 10
    n = len(tr.cameras)
    for s in range(2):
        tr.g.update_learning_rate(it + s)
        tr.deform.update_learning_rate(it + s)
        tr.deform_back.update_learning_rate(it + s)
        tr.bucket.zero()
        for r in range(world):
            cam = tr.cameras[T.frame_schedule(n, s, r, world, 0)]
            losses, _ = tr.loss_terms(cam, it + s)
            sum(losses.values()).backward()
        for o in tr.optimizers:
            o.step()
    for a, b in zip(r0, snapshot(tr)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (a - b).abs().max()
    # and the step really moved the parameters
    fresh = snapshot(make_trainer(0, 1))
    assert any(not torch.equal(a, b) for a, b in zip(r0, fresh))


# ---- mesh co-training phase: five networks in the bucket (2 607 545 floats = 10.43 MB) --------------
def make_mesh_trainer(rank, world, seed=0, **kw):
    D, T = pkg("deform"), pkg("trainer")
    tr = make_trainer(rank, world, seed=seed, **kw)
    torch.manual_seed(seed + 100)
    extra = [D.DeformModelNormalSep(is_blender=True, model_name="deform_normal", device="cpu", trunk_impl="torch"),
             D.DeformModelNormalSep(is_blender=True, model_name="deform_back_normal", device="cpu", trunk_impl="torch"),
             D.AppearanceModel(is_blender=True, device="cpu", trunk_impl="torch")]
    with torch.no_grad():  # the Sep networks' head is zero-initialised (time_utils.py:248-249): give it something to cycle on
        for m in extra[:2]:
            torch.nn.init.normal_(m.net.gaussian_normal.weight, std=0.02)
    mesh = T.MeshPhase(*extra, dpsr=None, n_verts=64, device="cpu", seed=seed)
    return T.Trainer(tr.g, tr.deform, tr.deform_back, tr.cameras, background=tr.bg, rank=rank, world=world, seed=seed,
                     render_fn=tr.render_fn, fused_adam=False, mesh=mesh)


def _mesh_snapshot(tr):
    ps = list(tr.params)
    return [p.detach().clone() for p in ps]


def _mesh_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = make_mesh_trainer(rank, world)
    it = tr.opt.dpsr_iter + tr.opt.normal_deform_delay + 10
    for s in range(2):
        tr.step(it + s)
    torch.save({"params": _mesh_snapshot(tr), "bytes": tr.grad_bytes()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_mesh_phase_bucket_holds_the_five_networks():
    """The data-parallel bucket of the mesh co-training phase: deform + deform_back (523 051 parameters each) and
    deform_normal, deform_back_normal, appearance (520 481 each) = 2 607 545 floats = 10.43 MB of MLP gradients -- the figure
    SURVEY.md section 8(e) / BASELINE.md name -- next to the Gaussian tensors (now including the normals) and the density
    threshold."""
    tr = make_mesh_trainer(0, 1)
    count = lambda m: sum(p.numel() for p in m.net.parameters())
    nets = [tr.deform, tr.deform_back] + tr.mesh.networks()
    assert [count(m) for m in nets] == [523051, 523051, 520481, 520481, 520481]
    P = tr.g._xyz.shape[0]
    assert tr.grad_bytes() == 4 * (2607545 + 62 * P + 1)  # 62 floats per Gaussian: 59 of the splat + 3 of the normal
    assert len(tr.optimizers) == 6  # R/train.py:517-524's six; the density threshold is the Gaussian optimizer's 8th group,
    names = [g["name"] for g in tr.g.optimizer.param_groups]  # scheduled 0.01 -> 1e-4 like the reference's (gaussian_model_*.py:201-229)
    assert names[-1] == "density_thres" and len(names) == 8
    tr.g.update_learning_rate(0)
    lr0 = tr.g.optimizer.param_groups[-1]["lr"]
    tr.g.update_learning_rate(tr.opt.position_lr_max_steps)
    assert abs(tr.g.optimizer.param_groups[-1]["lr"] - 1e-4) < 1e-9 and abs(lr0 - 0.01) < 1e-9


def test_mesh_phase_dp2_replicas_identical_and_all_networks_step():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        port = 29300 + (os.getpid() % 150)
        mp.start_processes(_mesh_worker, args=(world, port, d), nprocs=world, join=True, start_method="spawn")
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    fresh = _mesh_snapshot(make_mesh_trainer(0, 1))
    moved = [not torch.equal(a, b) for a, b in zip(r0["params"], fresh)]
    tr = make_mesh_trainer(0, 1)
    # every network of the phase received gradients and stepped (a weight of each changed), and so did the normals
    off = 6
    for m in [tr.deform, tr.deform_back, tr.mesh.deform_normal, tr.mesh.deform_back_normal]:
        n = len(list(m.net.parameters()))
        assert any(moved[off:off + n]), m.model_name
        off += n
    n_app = len(list(tr.mesh.appearance.net.parameters()))
    idx_normal = off + n_app
    assert moved[idx_normal], "the Gaussian normals did not move"
