"""MultiAdam (dgm_adam_step): the three torch.optim.Adam(eps=1e-15) steps of a train iteration as one kernel.
GPU: identical trajectory to torch.optim.Adam over several steps with per-group learning rates that change every step
(the reference rewrites param_groups[*]["lr"] each iteration) -- tolerance 1e-6 relative (same formula, fp32).
CPU: the flat-bucket packing used by the data-parallel path."""
import numpy as np
import pytest
import torch

from conftest import pkg


def test_bucket_pack_cpu():
    T = pkg("trainer")
    a, b, c = (torch.nn.Parameter(torch.zeros(3, 2)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2)))
    bk = T.FlatGradBucket([a, b, c], attach=False)
    assert a.grad is None
    (a.sum() * 2 + (b * torch.arange(5.0)).sum()).backward()     # c receives no gradient
    bk.flat.fill_(7.0)
    views = bk.pack()
    assert bk.flat.tolist() == [2.0] * 6 + [0.0, 1.0, 2.0, 3.0, 4.0] + [0.0, 0.0]
    assert views[id(b)].data_ptr() == bk.flat.data_ptr() + 6 * 4 and views[id(b)].shape == b.shape


@pytest.mark.gpu
def test_multi_adam_matches_torch_adam():
    O = pkg("optim")
    dev = "cuda"
    rng = np.random.RandomState(0)
    shapes = [(100000, 3), (100000, 15, 3), (100000, 1), (256, 352), (256,), (3, 256), (7,), (1,), (4099,)]
    ref = [torch.nn.Parameter(torch.tensor(rng.randn(*s).astype(np.float32), device=dev)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]

    def make(ps):
        o1 = torch.optim.Adam([{"params": [ps[0]], "lr": 1.6e-4, "name": "xyz"}, {"params": [ps[1]], "lr": 1.25e-4},
                               {"params": [ps[2]], "lr": 0.05}], lr=0.0, eps=1e-15)
        o2 = torch.optim.Adam([{"params": ps[3:7], "lr": 8e-4}], eps=1e-15)
        o3 = torch.optim.Adam([{"params": ps[7:], "lr": 1e-3}], eps=1e-15, betas=(0.8, 0.99))
        return [o1, o2, o3]

    o_ref, o_mine = make(ref), make(mine)
    ma = O.MultiAdam(o_mine)
    for it in range(6):
        scale = 10.0 ** (-it)                                   # gradients spanning many magnitudes
        for k, (p, q) in enumerate(zip(ref, mine)):
            g = torch.tensor(rng.randn(*p.shape).astype(np.float32), device=dev) * scale
            if it == 2 and k == 4:
                p.grad, q.grad = None, None                     # a tensor that skips a step keeps its own step count
            else:
                p.grad, q.grad = g, g.clone()
        for oa, ob in zip(o_ref, o_mine):                       # lr schedule: rewritten every iteration
            for ga, gb in zip(oa.param_groups, ob.param_groups):
                ga["lr"] = gb["lr"] = ga["lr"] * 0.9
        for o in o_ref:
            o.step()
        ma.step()
        for k, (p, q) in enumerate(zip(ref, mine)):
            err = (p - q).abs().max().item() / (p.abs().max().item() + 1e-30)
            assert err < 1e-6, f"step {it} tensor {k}: {err:.2e}"
    # gradients passed explicitly (views of a reduced bucket) behave the same as .grad
    g = {id(q): torch.ones_like(q) for q in mine}
    for p in ref:
        p.grad = torch.ones_like(p)
    for q in mine:
        q.grad = None
    for o in o_ref:
        o.step()
    ma.step(g)
    for p, q in zip(ref, mine):
        assert (p - q).abs().max().item() / (p.abs().max().item() + 1e-30) < 1e-6


@pytest.mark.gpu
def test_multi_adam_rejects_unsupported():
    O = pkg("optim")
    p = torch.nn.Parameter(torch.zeros(4, device="cuda"))
    with pytest.raises(ValueError):
        O.MultiAdam([torch.optim.Adam([p], amsgrad=True)])
    with pytest.raises(ValueError):
        O.MultiAdam([torch.optim.Adam([p], weight_decay=0.1)])


def test_multi_adam_refuses_cpu_tensors():
    """No CPU fallback: the one-launch Adam only accepts device tensors (and plain-Adam hyper parameters)."""
    O = pkg("optim")
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    ma = O.MultiAdam([torch.optim.Adam([p], lr=1e-3, eps=1e-15)])
    with pytest.raises(ValueError):
        ma.step()
    assert torch.equal(p.detach(), torch.zeros(4))
    with pytest.raises(ValueError):
        O.MultiAdam([torch.optim.Adam([p], amsgrad=True)])


@pytest.mark.gpu
def test_multi_adam_state_lives_in_the_optimizer_and_survives_surgery():
    """The moments are torch.optim.Adam-layout entries of optimizer.state, so reference-style surgery works on them:
    pruning rows of exp_avg / exp_avg_sq (_prune_optimizer), replacing a Parameter with zeroed moments
    (replace_tensor_to_optimizer, used by reset_opacity) -- and a replaced Parameter never inherits stale state."""
    O = pkg("optim")
    dev = "cuda"
    p = torch.nn.Parameter(torch.ones(10, 3, device=dev))
    ref = torch.nn.Parameter(p.detach().clone())
    opt = torch.optim.Adam([{"params": [p], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)
    opt_ref = torch.optim.Adam([{"params": [ref], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)
    ma = O.MultiAdam([opt])
    for _ in range(3):
        p.grad = torch.full_like(p, 0.5)
        ref.grad = torch.full_like(ref, 0.5)
        ma.step()
        opt_ref.step()
    st = opt.state[p]
    assert set(st) >= {"step", "exp_avg", "exp_avg_sq"} and int(st["step"]) == 3
    assert torch.allclose(st["exp_avg"], opt_ref.state[ref]["exp_avg"], rtol=1e-6)
    sd = opt.state_dict()
    assert sd["state"][0]["exp_avg"].shape == (10, 3)
    # _prune_optimizer (gaussian_model_dpsr_dynamic_anchor.py:383-401): keep rows 0..5, carry the moments along
    mask = torch.arange(10, device=dev) < 6
    for o, q in ((opt, p), (opt_ref, ref)):
        g = o.param_groups[0]
        s = o.state.pop(q)
        s["exp_avg"], s["exp_avg_sq"] = s["exp_avg"][mask], s["exp_avg_sq"][mask]
        g["params"][0] = torch.nn.Parameter(q.detach()[mask].clone().requires_grad_(True))
        o.state[g["params"][0]] = s
    p2, r2 = opt.param_groups[0]["params"][0], opt_ref.param_groups[0]["params"][0]
    p2.grad, r2.grad = torch.full_like(p2, -0.25), torch.full_like(r2, -0.25)
    ma.step()
    opt_ref.step()
    assert torch.allclose(p2, r2, rtol=1e-6, atol=1e-7) and int(opt.state[p2]["step"]) == 4
    # a replaced Parameter without state starts from zero moments and step 0, whatever id() it got
    del opt.state[p2]
    p3 = torch.nn.Parameter(torch.ones(6, 3, device=dev))
    opt.param_groups[0]["params"][0] = p3
    p3.grad = torch.ones_like(p3)
    ma.step()
    assert int(opt.state[p3]["step"]) == 1
    assert torch.allclose(p3, torch.full_like(p3, 1.0 - 0.1), rtol=1e-6)   # first Adam step moves by exactly lr


@pytest.mark.gpu
def test_multi_adam_fast_path_sees_in_place_surgery():
    """Between two repeated calls (the cached-pointer fast path) the state of an UNCHANGED Parameter is edited in place: a moment
    tensor replaced, the step tensor replaced, the parameter's storage re-pointed.  Every edit must reach the next step, as it
    would with torch.optim.Adam."""
    O = pkg("optim")
    dev = "cuda"
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(50, 3, device=dev)), torch.nn.Parameter(torch.randn(7, device=dev))]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = torch.optim.Adam([{"params": [ps[0]], "lr": 0.01}, {"params": [ps[1]], "lr": 0.02}], eps=1e-15)
    ref = torch.optim.Adam([{"params": [rs[0]], "lr": 0.01}, {"params": [rs[1]], "lr": 0.02}], eps=1e-15)
    ma = O.MultiAdam([opt])

    def both(k):
        for p, r in zip(ps, rs):
            g = torch.full_like(p, 0.1 * (k + 1))
            p.grad, r.grad = g, g.clone()
        ma.step()
        ref.step()

    for k in range(3):      # the third call runs on cached pointers
        both(k)
    # (1) reset one moment by replacing the tensor (what replace_tensor_to_optimizer does to a kept Parameter)
    opt.state[ps[0]]["exp_avg"] = torch.zeros_like(ps[0])
    ref.state[rs[0]]["exp_avg"] = torch.zeros_like(rs[0])
    both(3)
    # (2) rewind the step count by replacing the step tensor
    opt.state[ps[1]]["step"] = torch.tensor(1.0)
    ref.state[rs[1]]["step"] = torch.tensor(1.0)
    both(4)
    both(5)
    # (3) re-point the parameter's storage
    ps[0].data = ps[0].data.clone() * 0.5
    rs[0].data = rs[0].data.clone() * 0.5
    both(6)
    # (4) edit the step count in place
    opt.state[ps[1]]["step"].fill_(10.0)
    ref.state[rs[1]]["step"].fill_(10.0)
    for k in range(7, 7 + 70):  # past the 64-call revalidation of the general path
        both(k)
    for p, r in zip(ps, rs):
        assert torch.allclose(p, r, rtol=2e-6, atol=1e-7), (p - r).abs().max()
    assert int(opt.state[ps[1]]["step"]) == int(ref.state[rs[1]]["step"])


@pytest.mark.gpu
def test_adam_step_more_than_64_tensors_with_empty_ones():
    """dgm_adam_step batches 64 tensors per launch; empty tensors are skipped without disturbing the batching (each
    tensor must be updated exactly once)."""
    import ctypes
    L = pkg("_lib").lib()
    dev = "cuda"
    sizes = [0 if i % 7 == 3 else 5 + i for i in range(150)]
    ps = [torch.zeros(max(n, 1), device=dev)[:n] for n in sizes]
    gs = [torch.ones(max(n, 1), device=dev)[:n] for n in sizes]
    ms = [torch.zeros(max(n, 1), device=dev)[:n] for n in sizes]
    vs = [torch.zeros(max(n, 1), device=dev)[:n] for n in sizes]
    n = len(sizes)
    VP = ctypes.c_void_p * n
    ptr = lambda ts: VP(*[t.data_ptr() if t.numel() else None for t in ts])
    rc = L.dgm_adam_step(n, ptr(ps), ptr(gs), ptr(ms), ptr(vs), (ctypes.c_longlong * n)(*sizes),
                         (ctypes.c_float * n)(*([0.5] * n)), (ctypes.c_int * n)(*([1] * n)), 0.9, 0.999, 1e-15,
                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    for i, (p, m) in enumerate(zip(ps, ms)):
        if p.numel():
            assert torch.allclose(p, torch.full_like(p, -0.5), rtol=1e-6), f"tensor {i} updated {p[0].item() / -0.5:.2f} times"
            assert torch.allclose(m, torch.full_like(m, 0.1), rtol=1e-6)
