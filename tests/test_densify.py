"""Densification / pruning / opacity reset on the device (dg-mesh_amd/densify.py, csrc/densify.hip) against (1) goldens
produced by the REFERENCE's own methods executed from source (tests/golden/densify_surgery.npz, make_golden.py::densify_golden)
and (2), at other sizes, a PyTorch restatement of the reference's optimizer surgery -- itself checked against the goldens --
(/root/reference/dgmesh/scene/gaussian_model_dpsr_dynamic_anchor.py:291-294, 364-551), fed the same statistics, the same
Adam state and the same standard-normal samples: same decisions (the new P), bit-equal gathered parameters and moments,
split children to fp32 rounding.  Plus: the optimizer keeps working on the new set, and two data-parallel replicas stay
bit-identical through a densify step."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import ROOT, pkg

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "normal")


def build_rotation(r):
    """R/utils/general_utils.py:130-149."""
    q = r / torch.sqrt((r * r).sum(1))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    R[:, 0, 0], R[:, 0, 1], R[:, 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)
    R[:, 1, 0], R[:, 1, 1], R[:, 1, 2] = 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)
    R[:, 2, 0], R[:, 2, 1], R[:, 2, 2] = 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)
    return R


class RefSurgery:
    """The reference's sequence on plain tensors: params / exp_avg / exp_avg_sq dicts keyed by group name."""

    def __init__(self, params, m, v, accum, denom, max_radii, percent_dense):
        self.p, self.m, self.v = dict(params), dict(m), dict(v)
        self.accum, self.denom, self.max_radii, self.percent_dense = accum, denom, max_radii, percent_dense

    def scaling(self):
        return torch.exp(self.p["scaling"])

    def _cat(self, new):  # cat_tensors_to_optimizer + densification_postfix (:421-460)
        for k in NAMES:
            self.p[k] = torch.cat((self.p[k], new[k]), 0)
            self.m[k] = torch.cat((self.m[k], torch.zeros_like(new[k])), 0)
            self.v[k] = torch.cat((self.v[k], torch.zeros_like(new[k])), 0)
        P = self.p["xyz"].shape[0]
        dev = self.p["xyz"].device
        self.accum, self.denom, self.max_radii = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros(P, device=dev)

    def prune_points(self, mask):  # :383-419
        keep = ~mask
        for k in NAMES:
            self.p[k], self.m[k], self.v[k] = self.p[k][keep], self.m[k][keep], self.v[k][keep]
        self.accum, self.denom, self.max_radii = self.accum[keep], self.denom[keep], self.max_radii[keep]

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, z):
        grads = self.accum / self.denom
        grads[grads.isnan()] = 0.0
        # densify_and_clone (:487-506)
        sel = (torch.norm(grads, dim=-1) >= max_grad) & (self.scaling().max(1).values <= self.percent_dense * extent)
        self._cat({k: self.p[k][sel] for k in NAMES})
        # densify_and_split (:462-485), N = 2; z[c][i] is the sample of copy c of source row i
        n0 = self.p["xyz"].shape[0]
        padded = torch.zeros(n0, device=grads.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= max_grad) & (self.scaling().max(1).values > self.percent_dense * extent)
        idx = sel.nonzero().squeeze(1)
        stds = self.scaling()[sel].repeat(2, 1)
        samples = stds * torch.cat((z[0][idx], z[1][idx]), 0)
        rots = build_rotation(self.p["rotation"][sel]).repeat(2, 1, 1)
        new = {k: self.p[k][sel].repeat(2, *([1] * (self.p[k].dim() - 1))) for k in NAMES}
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.p["xyz"][sel].repeat(2, 1)
        new["scaling"] = torch.log(self.scaling()[sel].repeat(2, 1) / (0.8 * 2))
        self._cat(new)
        self.prune_points(torch.cat((sel, torch.zeros(2 * int(sel.sum()), device=sel.device, dtype=torch.bool))))
        # prune (:532-540)
        mask = (torch.sigmoid(self.p["opacity"]) < min_opacity).squeeze()
        if max_screen_size:
            mask = mask | (self.max_radii > max_screen_size) | (self.scaling().max(1).values > 0.1 * extent)
        self.prune_points(mask)


def make_model(P, seed=0, steps=2):
    syn, S = pkg("synthetic"), pkg("scene")
    dev = torch.device("cuda")
    rng = np.random.RandomState(seed)
    gn = syn.make_gaussians(P, seed=seed, kind="aniso", extent=1.0, dist2=np.full(P, 2e-3, np.float32))
    g = S.GaussianModel(sh_degree=3, device=dev)
    g.load_raw(gn["xyz"], gn["features_dc"], gn["features_rest"], gn["scaling"], gn["rotation"],
               (rng.randn(P, 1) * 3).astype(np.float32), rng.randn(P, 3).astype(np.float32))
    g.training_setup(S.OptimizationParams())
    for grp in g.optimizer.param_groups:
        grp["lr"] = 1e-3
    ma = pkg("optim").MultiAdam([g.optimizer])
    gen = torch.Generator(device=dev).manual_seed(seed + 7)
    for _ in range(steps):  # give every tensor non-trivial Adam moments
        for p in g.parameters():
            p.grad = torch.randn(p.shape, device=dev, generator=gen)
        ma.step()
    for p in g.parameters():
        p.grad = None
    g.xyz_gradient_accum = torch.rand((P, 1), device=dev, generator=gen) * 6e-4 * 3
    g.denom = torch.randint(0, 4, (P, 1), device=dev, generator=gen).float()   # zeros -> NaN -> 0
    g.max_radii2D = torch.rand(P, device=dev, generator=gen) * 40
    return g, ma


def ref_of(g, S):
    grp = {x["name"]: x["params"][0] for x in g.optimizer.param_groups}
    st = g.optimizer.state
    return RefSurgery({k: grp[k].detach().clone() for k in NAMES}, {k: st[grp[k]]["exp_avg"].clone() for k in NAMES},
                      {k: st[grp[k]]["exp_avg_sq"].clone() for k in NAMES}, g.xyz_gradient_accum.clone(), g.denom.clone(),
                      g.max_radii2D.clone(), g.percent_dense)


def compare(g, ref, n_before):
    grp = {x["name"]: x["params"][0] for x in g.optimizer.param_groups}
    Pn = ref.p["xyz"].shape[0]
    assert g._xyz.shape[0] == Pn
    for k in NAMES:
        attr = getattr(g, pkg("densify").ATTR[k])
        assert attr is grp[k] and attr.requires_grad and attr.shape == ref.p[k].shape
        st = g.optimizer.state[grp[k]]
        if k in ("xyz", "scaling"):  # split children are computed (R s z + xyz, log(s / 1.6)): fp32 rounding only
            assert torch.allclose(attr, ref.p[k], rtol=1e-5, atol=1e-6), k
        else:
            assert torch.equal(attr.detach(), ref.p[k]), k
        assert torch.equal(st["exp_avg"], ref.m[k]) and torch.equal(st["exp_avg_sq"], ref.v[k]), k
        assert int(st["step"]) == 2
    assert g.xyz_gradient_accum.shape == (Pn, 1) and g.denom.shape == (Pn, 1) and g.max_radii2D.shape == (Pn,)
    assert torch.equal(g.xyz_gradient_accum, ref.accum) and torch.equal(g.max_radii2D, ref.max_radii)


@pytest.mark.gpu
@pytest.mark.parametrize("P,size_limit", [(20000, None), (20000, 20), (257, 20), (100000, 20)])
def test_densify_and_prune_matches_reference_surgery(P, size_limit):
    S = pkg("scene")
    g, ma = make_model(P, seed=P % 7)
    ref = ref_of(g, S)
    extent = 0.9
    gen = torch.Generator(device="cuda").manual_seed(99)
    state = gen.get_state()
    Pn = g.densify_and_prune(0.0002, 0.005, extent, size_limit, generator=gen)
    gen.set_state(state)
    z = torch.randn((2, P, 3), device="cuda", generator=gen)
    ref.densify_and_prune(0.0002, 0.005, extent, size_limit, z)
    assert Pn == ref.p["xyz"].shape[0] and Pn != P
    compare(g, ref, P)
    # the optimizer keeps working on the new set: moments of survivors continue, new rows start from zero
    for p in g.parameters():
        p.grad = torch.ones_like(p)
    ma.step()
    st = g.optimizer.state[g._xyz]
    assert int(st["step"]) == 3 and st["exp_avg"].shape == g._xyz.shape


@pytest.mark.gpu
def test_prune_points_and_reset_opacity():
    S = pkg("scene")
    g, ma = make_model(5000, seed=3)
    ref = ref_of(g, S)
    mask = torch.rand(5000, device="cuda") < 0.3
    g.prune_points(mask)
    ref.prune_points(mask)
    compare(g, ref, 5000)
    assert torch.equal(g.denom, ref.denom)
    # reset_opacity (:291-294): opacity <- inverse_sigmoid(min(sigmoid(opacity), 0.01)), moments zeroed, other groups untouched
    before = g._opacity.detach().clone()
    m_xyz = g.optimizer.state[g._xyz]["exp_avg"].clone()
    g.reset_opacity()
    want = torch.log(torch.clamp_max(torch.sigmoid(before), 0.01) / (1 - torch.clamp_max(torch.sigmoid(before), 0.01)))
    assert torch.allclose(g._opacity, want, rtol=1e-6, atol=1e-6)
    grp = {x["name"]: x["params"][0] for x in g.optimizer.param_groups}
    assert grp["opacity"] is g._opacity
    assert float(g.optimizer.state[g._opacity]["exp_avg"].abs().sum()) == 0.0
    assert torch.equal(g.optimizer.state[g._xyz]["exp_avg"], m_xyz)


GOLD = os.path.join(ROOT, "tests", "golden", "densify_surgery.npz")
ATTRS = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
             rotation="_rotation", normal="_normal")


def _load_case(gold, tag, dev):
    """A GaussianModel (sh_degree 1) + Adam state exactly as the golden generator left the reference host before surgery."""
    S = pkg("scene")
    g = S.GaussianModel(sh_degree=1, device=dev)
    t = lambda k: torch.tensor(gold[k], device=dev)
    g.load_raw(*[t(f"{tag}/p/{k}") for k in ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "normal")])
    g.training_setup(S.OptimizationParams())
    grp = {x["name"]: x["params"][0] for x in g.optimizer.param_groups}
    for k in NAMES:
        g.optimizer.state[grp[k]] = {"step": torch.tensor(float(gold[f"{tag}/step/{k}"])), "exp_avg": t(f"{tag}/m/{k}"),
                                     "exp_avg_sq": t(f"{tag}/v/{k}")}
    g.xyz_gradient_accum, g.denom, g.max_radii2D = t(f"{tag}/accum"), t(f"{tag}/denom"), t(f"{tag}/max_radii")
    return g


def _check_case(g, gold, tag, computed=("xyz", "scaling")):
    grp = {x["name"]: x["params"][0] for x in g.optimizer.param_groups}
    for k in NAMES:
        attr = getattr(g, ATTRS[k])
        want = torch.tensor(gold[f"{tag}/p/{k}"], device=attr.device)
        assert attr is grp[k] and attr.shape == want.shape, k
        if k in computed:  # split children / reset opacities are computed: fp32 rounding only
            assert torch.allclose(attr.detach(), want, rtol=1e-5, atol=1e-6), k
        else:
            assert torch.equal(attr.detach(), want), k
        st = g.optimizer.state[grp[k]]
        assert torch.equal(st["exp_avg"].cpu(), torch.tensor(gold[f"{tag}/m/{k}"])), k
        assert torch.equal(st["exp_avg_sq"].cpu(), torch.tensor(gold[f"{tag}/v/{k}"])), k
        assert float(st["step"]) == float(gold[f"{tag}/step/{k}"])
    assert torch.equal(g.xyz_gradient_accum.cpu(), torch.tensor(gold[f"{tag}/accum"]))
    assert torch.equal(g.denom.cpu(), torch.tensor(gold[f"{tag}/denom"]))
    assert torch.equal(g.max_radii2D.cpu(), torch.tensor(gold[f"{tag}/max_radii"]))


@pytest.mark.gpu
def test_surgery_matches_the_reference_code_goldens():
    """tests/golden/densify_surgery.npz holds what the REFERENCE's own methods (densify_and_prune -> prune_points ->
    reset_opacity of gaussian_model_dpsr_dynamic_anchor.py, executed from source by make_golden.py) did to a model, its Adam
    state and its statistics; the device implementation is fed the same model and the same normal draws."""
    gold = np.load(GOLD)
    dev = torch.device("cuda")
    for case in (0, 1):
        g = _load_case(gold, f"c{case}/in", dev)
        max_grad, min_opacity, extent, size_limit, pd = gold[f"c{case}/args"]
        assert abs(g.percent_dense - pd) < 1e-12
        Pn = g.densify_and_prune(max_grad, min_opacity, extent, None if size_limit < 0 else size_limit, samples=gold[f"c{case}/z"])
        assert Pn == gold[f"c{case}/out/p/xyz"].shape[0]
        _check_case(g, gold, f"c{case}/out")
        if case == 0:
            g.prune_points(torch.tensor(gold["c0/prune_mask"], device=dev))
            _check_case(g, gold, "c0/pruned")
            g.reset_opacity()
            _check_case(g, gold, "c0/reset", computed=("xyz", "scaling", "opacity"))


def test_restatement_matches_the_reference_code_goldens():
    """The PyTorch restatement the other tests compare with (RefSurgery, used at sizes and seeds the golden does not cover)
    reproduces the reference's own result on the golden inputs -- on the CPU, bit for bit where nothing is computed."""
    gold = np.load(GOLD)
    t = lambda k: torch.tensor(gold[k])
    for case in (0, 1):
        tag = f"c{case}/in"
        max_grad, min_opacity, extent, size_limit, pd = gold[f"c{case}/args"]
        ref = RefSurgery({k: t(f"{tag}/p/{k}") for k in NAMES}, {k: t(f"{tag}/m/{k}") for k in NAMES},
                         {k: t(f"{tag}/v/{k}") for k in NAMES}, t(f"{tag}/accum"), t(f"{tag}/denom"), t(f"{tag}/max_radii"), float(pd))
        ref.densify_and_prune(max_grad, min_opacity, extent, None if size_limit < 0 else size_limit, t(f"c{case}/z"))
        for k in NAMES:
            want = t(f"c{case}/out/p/{k}")
            assert ref.p[k].shape == want.shape
            if k in ("xyz", "scaling"):
                assert torch.allclose(ref.p[k], want, rtol=1e-6, atol=1e-7), k
            else:
                assert torch.equal(ref.p[k], want), k
            assert torch.equal(ref.m[k], t(f"c{case}/out/m/{k}")) and torch.equal(ref.v[k], t(f"c{case}/out/v/{k}"))
        assert torch.equal(ref.max_radii, t(f"c{case}/out/max_radii"))


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_trainer_dp_gpu import make_trainer, snapshot
    tr = make_trainer(rank, world)
    tr.densify = True
    tr.cameras_extent = 1.0
    tr.opt.densify_from_iter = 0
    tr.opt.densification_interval = 3
    tr.opt.densify_grad_threshold = 1e-7   # make sure clones AND splits happen on this tiny scene
    it0 = tr.opt.warm_up + 1
    sizes = []
    for s in range(5):                      # iterations it0 .. it0+4 contain one multiple of 3
        tr.step(it0 + s)
        sizes.append(tr.g._xyz.shape[0])
    torch.cuda.synchronize()
    torch.save({"params": snapshot(tr), "sizes": sizes,
                "moments": [tr.g.optimizer.state[p]["exp_avg"].cpu() for p in tr.g.parameters()[:6]]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_dp2_replicas_stay_identical_through_a_densify_step():
    import torch.multiprocessing as mp
    world = 2
    with tempfile.TemporaryDirectory() as d:
        port = 29900 + (os.getpid() % 2000)
        mp.start_processes(_worker, args=(world, port, d), nprocs=world, join=True, start_method="spawn")
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    assert r0["sizes"] == r1["sizes"] and len(set(r0["sizes"])) > 1     # P changed, identically on both ranks
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    for a, b in zip(r0["moments"], r1["moments"]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_stats_kernel_equals_reference_indexing():
    """dgm_densify_stats (one launch; what GaussianModel.track_densification_stats uses on the GPU) == the reference's
    boolean-index bookkeeping (train.py:489-496 + add_densification_stats, gaussian_model_dpsr_dynamic_anchor.py:679-682):
    max_radii2D and denom bit-equal, the gradient-norm accumulator to fp32 rounding.  Also the lean form: no filter given,
    no gradient (max_radii2D only)."""
    S = pkg("scene")
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(3)
    for P in (1, 255, 1000):
        g1, g2 = S.GaussianModel(sh_degree=3, device=dev), S.GaussianModel(sh_degree=3, device=dev)
        raw = [rng.randn(P, 3), rng.randn(P, 1, 3), rng.randn(P, 15, 3), rng.randn(P, 3), rng.randn(P, 4), rng.randn(P, 1)]
        for g in (g1, g2):
            g.load_raw(*raw)
            g.training_setup(S.OptimizationParams())
        for step in range(3):
            vp = torch.zeros(P, 3, device=dev, requires_grad=True)
            vp.grad = torch.tensor(rng.randn(P, 3).astype(np.float32), device=dev)
            radii = torch.tensor(rng.randint(0, 40, P).astype(np.int32), device=dev)
            vis = radii > 0
            g1.max_radii2D[vis] = torch.max(g1.max_radii2D[vis], radii[vis].to(g1.max_radii2D.dtype))
            g1.add_densification_stats(vp, vis)
            g2.track_densification_stats(vp, None, radii)
        assert torch.equal(g1.max_radii2D, g2.max_radii2D) and torch.equal(g1.denom, g2.denom)
        assert torch.allclose(g1.xyz_gradient_accum, g2.xyz_gradient_accum, rtol=1e-6, atol=0)
        before = (g2.xyz_gradient_accum.clone(), g2.denom.clone())
        radii = torch.full((P,), 77, dtype=torch.int32, device=dev)
        g2.track_densification_stats(None, None, radii)
        assert bool((g2.max_radii2D == 77).all())
        assert torch.equal(before[0], g2.xyz_gradient_accum) and torch.equal(before[1], g2.denom)
