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


def make_trainer(rank, world, P=3000, W=160, H=128, n_frames=6, seed=0, **kw):
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
    return T.Trainer(g, deform, deform_back, cams, background=bg, rank=rank, world=world, seed=seed, **kw)


def snapshot(tr):
    ps = tr.g.parameters()[:6] + list(tr.deform.net.parameters()) + list(tr.deform_back.net.parameters())
    return [p.detach().cpu().clone() for p in ps]


def _worker(rank, world, port, out_dir, overlap=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = make_trainer(rank, world, overlap=overlap)
    assert tr.pack and tr.multi_adam is not None and tr.bucket is not None and (tr._early is not None) == overlap
    it = tr.opt.warm_up + 10
    for s in range(2):
        tr.step(it + s)
    torch.cuda.synchronize()
    torch.save(snapshot(tr), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_dp2_pack_mode_on_gpu(overlap):
    """overlap=False (the default): one flat bucket after backward.  overlap=True: the Gaussian bucket's all-reduce is launched
    from an autograd hook under the MLP backward passes, the MLP bucket follows.  Either way the replicas are bit-identical."""
    world = 2
    T = pkg("trainer")
    with tempfile.TemporaryDirectory() as d:
        port = 29700 + (os.getpid() % 2000) + (7 if overlap else 0)
        mp.start_processes(_worker, args=(world, port, d, overlap), nprocs=world, join=True, start_method="spawn")
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)                               # replicas stay bit-identical
    # one rank, gradients of the same two frames summed by hand, then the same one-launch Adam
    tr = make_trainer(0, 1)  # (driven by hand below)
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


def make_mesh_trainer(rank, world, res=48, n_verts=4000, **kw):
    D, T, DP = pkg("deform"), pkg("trainer"), pkg("dpsr")
    tr = make_trainer(rank, world, **kw)
    dev = tr.g.get_xyz.device
    torch.manual_seed(7)
    dn = D.DeformModelNormalSep(is_blender=True, model_name="deform_normal", device=dev, trunk_impl="hip")
    dbn = D.DeformModelNormalSep(is_blender=True, model_name="deform_back_normal", device=dev, trunk_impl="hip")
    app = D.AppearanceModel(is_blender=True, device=dev, trunk_impl="hip")  # differentiates w.r.t. its input (dgm_mlp_backward_dx)
    with torch.no_grad():
        for m in (dn, dbn):
            torch.nn.init.normal_(m.net.gaussian_normal.weight, std=0.02)
        tr.g._normal.copy_(torch.nn.functional.normalize(tr.g.get_xyz.detach(), dim=1))
    mesh = T.MeshPhase(dn, dbn, app, dpsr=DP.DPSR(res=(res,) * 3, sig=2.0), n_verts=n_verts, scale=1.0, device=dev, stand_in_weight=1e-3)
    return T.Trainer(tr.g, tr.deform, tr.deform_back, tr.cameras, background=tr.bg, rank=rank, world=world, seed=0, mesh=mesh)


@pytest.mark.gpu
def test_mesh_phase_step_runs_the_dpsr_chain_and_moves_every_network():
    """One process, the mesh co-training phase on the GPU code path: four fused-trunk networks on P, the DPSR chain (HIP splat ->
    rocFFT -> HIP spectral solve -> HIP read-back, all with their adjoints), deform_back + appearance on the probe vertices, the
    one-launch Adam over seven optimizers.  Every network, the Gaussian normals and positions and the density threshold must
    receive a finite gradient and move; two identically seeded trainers agree to rounding (the DPSR splat accumulates with
    atomics, like the reference's scatter_add: not bit-reproducible)."""
    a, b = make_mesh_trainer(0, 1), make_mesh_trainer(0, 1)
    before = [p.detach().clone() for p in a.params]
    it = a.opt.dpsr_iter + a.opt.normal_deform_delay + 1000
    for s in range(2):
        la, _ = a.step(it + s)
        lb, _ = b.step(it + s)
    torch.cuda.synchronize()
    assert torch.isfinite(la) and abs(float(la) - float(lb)) < 1e-5 * abs(float(la))
    for x, y in zip(a.params, b.params):  # (Adam turns a rounding-level gradient difference into at most +-lr per step)
        assert float((x.detach() - y.detach()).abs().max()) <= 5e-3
    moved = [not torch.equal(x.detach(), y) for x, y in zip(a.params, before)]
    off = 6
    for m in [a.deform, a.deform_back] + a.mesh.networks():
        n = len(list(m.net.parameters()))
        assert any(moved[off:off + n]), m.model_name
        off += n
    assert moved[0] and moved[off] and moved[off + 1], "positions / normals / density threshold did not move"
    assert all(bool(torch.isfinite(p).all()) for p in a.params)


@pytest.mark.gpu
def test_sync_free_steps_equal_the_synchronous_ones_and_an_overflowed_frame_is_redone():
    """Trainer.step over the sync-free rasterizer forward (rasterizer.SYNC_FREE: capacity-sized binning buffer, {R, flags} looked at
    after the backward has been enqueued): six steps leave bit-identical parameters to six steps over the synchronous forward -- with
    one frame's capacity forced too small on the way, which must be noticed before the optimizer runs and rendered again."""
    RZ = pkg("rasterizer")
    it = None
    snaps = []
    prev = (RZ.SYNC_FREE, dict(RZ._SF))
    try:
        for mode in (False, True):
            RZ.SYNC_FREE = mode
            RZ._SF.clear()
            tr = make_trainer(0, 1)
            it = tr.opt.warm_up + 10
            t_fwd0 = RZ.FORWARD_CALL_SECONDS
            for s in range(6):
                if mode and s == 3:
                    RZ.INJECT_CAPACITY = 4096
                tr.step(it + s)
            torch.cuda.synchronize()
            snaps.append(snapshot(tr))
            if mode:
                assert getattr(tr, "redone_frames", 0) >= 1 and RZ.INJECT_CAPACITY == 0
                assert RZ.LAST_NUM_RENDERED > 4096
    finally:
        RZ.SYNC_FREE = prev[0]
        RZ._SF.clear()
        RZ._SF.update(prev[1])
        RZ.INJECT_CAPACITY = 0
    for a, b in zip(*snaps):
        assert torch.equal(a, b)


# ---- the 8-rank path hardened on the hardware there is: N ranks sharing cuda:0 over gloo, many steps, a densification event -------
def _long_worker(rank, world, port, out_dir, overlap, sync_free, steps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import importlib
    import json
    import time
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench = importlib.import_module("bench")
    RZ = pkg("rasterizer")
    RZ.SYNC_FREE = bool(sync_free)
    bench.WORKLOAD, bench.TARGETS = "cfg1", "teacher"
    dev = torch.device("cuda:0")
    tr, (P0, W, H) = bench.build_scene(dev, rank, world, "hip", densify=True)
    tr.overlap = bool(overlap)
    tr._bind_parameters()
    assert (tr._early is not None) == bool(overlap)
    tr.opt.densify_grad_threshold = 1e-5  # (synthetic targets give small view-space gradients: the reference's 2e-4 selects nothing)
    first = tr.opt.warm_up + 2000 + 100 - steps // 2 + 1  # the densification iteration (a multiple of 100) falls mid-run
    checks, P_path = [], [int(tr.g.get_xyz.shape[0])]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        it = first + s
        tr.step(it)
        if it % tr.opt.densification_interval == 0:
            P_path.append(int(tr.g.get_xyz.shape[0]))
        if (s + 1) % 10 == 0:
            checks.append(bool(tr.replicas_identical()))  # 64-bit hash of every parameter and Adam moment, all-gathered
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        with open(os.path.join(out_dir, "rec.json"), "w") as fh:
            json.dump({"workload": "cfg1 (400x400, P=20000), densify=True, one frame per rank per step", "world": world,
                       "ranks_share": "cuda:0 (one-GPU box), gloo", "overlap": bool(overlap), "sync_free_forward": bool(sync_free),
                       "steps": steps, "first_iteration": first, "P": P_path, "replicas_identical_every_10_steps": checks,
                       "frames_redone_for_capacity": int(getattr(tr, "redone_frames", 0)) + int(RZ.OVERFLOW_REDOS),
                       "ms_per_step_time_sliced": round(1e3 * dt / steps, 3)}, fh, indent=1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,overlap,sync_free", [(4, True, False), (2, False, True)])
def test_many_ranks_many_steps_with_a_densification_event(world, overlap, sync_free):
    """What the first 8-GPU run will do, as far as one GPU can show it: `world` ranks (sharing cuda:0, gloo) train cfg1 for 60 steps
    with densification on; the parameters and Adam moments of all ranks are hashed and compared every 10 steps -- the
    non-reproducibility round 3 saw with a second queue was ~1 event in 60-100 evaluations, which 8 steps cannot see.
    (4, overlap): the Gaussian bucket's all-reduce launched from the autograd hook under the MLP backward passes.
    (2, sync-free): the rasterizer forward that never waits for the device, its capacity check deferred behind the backward."""
    import json
    steps = 60
    with tempfile.TemporaryDirectory() as d:
        port = 29900 + (os.getpid() % 1000) + 13 * world
        mp.start_processes(_long_worker, args=(world, port, d, overlap, sync_free, steps), nprocs=world, join=True, start_method="spawn")
        rec = json.load(open(os.path.join(d, "rec.json")))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = f"dp{world}_shared_gpu{'_overlap' if overlap else ''}{'_sync_free' if sync_free else ''}.json"
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
    print(rec)
    assert len(rec["replicas_identical_every_10_steps"]) == steps // 10 and all(rec["replicas_identical_every_10_steps"])
    assert len(rec["P"]) == 2  # one densification event inside the run
