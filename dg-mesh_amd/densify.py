"""Densification / pruning / opacity reset of the Gaussian set, on the device.

Same decisions and resulting tensors as the reference's optimizer surgery
  R/scene/gaussian_model_dpsr_dynamic_anchor.py:291-294 (reset_opacity), :364-381 (replace_tensor_to_optimizer),
  :383-419 (_prune_optimizer, prune_points), :421-460 (cat_tensors_to_optimizer, densification_postfix),
  :462-506 (densify_and_split, densify_and_clone), :521-551 (prune, densify_and_prune)
but as ONE pass: libdgmesh_hip decides per Gaussian (keep / clone / split / prune), scans, and gathers every parameter and
both Adam moments of every parameter group into their final size with a single multi-tensor launch
(csrc/densify.hip) instead of ~60 boolean-index / cat kernels and three re-allocations per tensor.  The host reads
back three integers (kept / cloned / split) once per call to size the new tensors.

The normal samples of densify_and_split come from `generator` (a device torch.Generator).  Data-parallel replicas pass
generators seeded alike, so every rank builds the identical new set.  No CPU fallback.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "normal")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation", "normal": "_normal"}


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return _lib.stream_ptr()


def _decide(g, P, keep_mask=None, args=None):
    L = _lib.lib()
    dev = g._xyz.device
    scratch = torch.empty(L.dgm_densify_scratch_bytes(P), dtype=torch.uint8, device=dev)
    with _lib.device_guard(dev):
        if keep_mask is not None:
            km = keep_mask.to(torch.uint8).contiguous()
            _lib.check(L.dgm_densify_decide(P, None, None, None, None, 0.0, 0.0, 0.0, 0.0, _vp(km), _vp(scratch), _stream()))
        else:
            ga, dn, thr, dense, min_op, big = args
            _lib.check(L.dgm_densify_decide(P, _vp(ga), _vp(dn), _vp(g._scaling.detach().contiguous()),
                                            _vp(g._opacity.detach().contiguous()), thr, dense, min_op, big, None,
                                            _vp(scratch), _stream()))
    off = L.dgm_densify_totals_offset(P)
    K, C, S = (int(v) for v in scratch[off:off + 12].view(torch.int32).tolist())  # the call's one read-back
    return scratch, K, C, S


def _apply(g, P, scratch, K, C, S, z):
    """Gather all parameters + Adam moments into the new set and re-seat them in the optimizer / model."""
    L = _lib.lib()
    dev = g._xyz.device
    Pn = K + C + 2 * S
    opt = g.optimizer
    by_name = {grp["name"]: grp for grp in opt.param_groups} if opt is not None else {}
    entries = []  # (name, kind, old tensor, new tensor)
    empty = {}    # zero-width tensors (max_sh_degree = 0: _features_rest is (P, 0, 3)): nothing to gather, new ones made directly
    for name in GROUPS:
        old = getattr(g, ATTR[name])
        new = torch.empty((Pn,) + tuple(old.shape[1:]), dtype=torch.float32, device=dev)
        st = opt.state.get(by_name[name]["params"][0]) if name in by_name else None
        if old.numel() == 0 and P > 0:
            empty[name] = {"param": new}
            if st is not None and "exp_avg" in st:
                empty[name].update(exp_avg=torch.empty_like(new), exp_avg_sq=torch.empty_like(new))
            continue
        entries.append((name, "param", old.detach().contiguous(), new))
        if st is not None and "exp_avg" in st:
            for key in ("exp_avg", "exp_avg_sq"):
                entries.append((name, key, st[key].contiguous(), torch.empty_like(new)))
    n = len(entries)
    idx = {nm: i for i, (nm, kind, _, _) in enumerate(entries) if kind == "param"}
    VP = ctypes.c_void_p * n
    if Pn > 0:
        src = torch.empty(Pn, dtype=torch.int32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(L.dgm_densify_apply(
                P, K, C, S, _vp(scratch), _vp(src), n, VP(*[e[2].data_ptr() for e in entries]),
                VP(*[e[3].data_ptr() for e in entries]),
                (ctypes.c_int * n)(*[max(e[2].numel() // max(P, 1), 1) for e in entries]),
                (ctypes.c_int * n)(*[0 if e[1] == "param" else 1 for e in entries]), idx["xyz"], idx["scaling"],
                idx["rotation"], _vp(z) if z is not None else None, _stream()))
    new_state = dict(empty)
    for name, kind, _, new in entries:
        new_state.setdefault(name, {})[kind] = new
    for name in GROUPS:
        p_new = nn.Parameter(new_state[name]["param"].requires_grad_(True))
        if name in by_name:
            grp = by_name[name]
            st = opt.state.pop(grp["params"][0], None)
            grp["params"][0] = p_new
            if st is not None:
                if "exp_avg" in new_state[name]:
                    st["exp_avg"], st["exp_avg_sq"] = new_state[name]["exp_avg"], new_state[name]["exp_avg_sq"]
                opt.state[p_new] = st
        setattr(g, ATTR[name], p_new)
    return Pn


@torch.no_grad()
def densify_and_prune(g, max_grad, min_opacity, extent, max_screen_size, generator=None, samples=None):
    """densify_and_prune (:542-551).  Returns the new number of Gaussians.  `samples`: optional (2, P, 3) standard-normal
    draws for the split children (copy, source row) instead of drawing them from `generator` -- what lets a test feed the
    draws the reference made."""
    P = g._xyz.shape[0]
    if P == 0:
        return 0
    dev = g._xyz.device
    big = 0.1 * extent if max_screen_size else float("inf")
    args = (g.xyz_gradient_accum.contiguous(), g.denom.contiguous(), float(max_grad), float(g.percent_dense * extent),
            float(min_opacity), float(big))
    scratch, K, C, S = _decide(g, P, args=args)
    # one standard-normal triple per (copy, Gaussian): indexed by the SOURCE row, so the samples a split child receives
    # do not depend on how many other Gaussians were selected
    if samples is not None:
        z = torch.as_tensor(samples, dtype=torch.float32, device=dev).contiguous()
        if tuple(z.shape) != (2, P, 3):
            raise RuntimeError("densify_and_prune: samples must be (2, P, 3)")
    else:
        z = torch.randn((2, P, 3), device=dev, generator=generator) if S > 0 else None
    Pn = _apply(g, P, scratch, K, C, S, z)
    g.xyz_gradient_accum = torch.zeros((Pn, 1), device=dev)   # densification_postfix (:458-460)
    g.denom = torch.zeros((Pn, 1), device=dev)
    g.max_radii2D = torch.zeros((Pn,), device=dev)
    return Pn


@torch.no_grad()
def prune_points(g, mask):
    """prune_points (:403-419): remove the Gaussians where `mask` is True, statistics travel with the survivors."""
    P = g._xyz.shape[0]
    if P == 0:
        return 0
    keep = ~mask.to(torch.bool).reshape(-1)
    scratch, K, C, S = _decide(g, P, keep_mask=keep)
    Pn = _apply(g, P, scratch, K, 0, 0, None)
    g.xyz_gradient_accum = g.xyz_gradient_accum[keep]
    g.denom = g.denom[keep]
    g.max_radii2D = g.max_radii2D[keep]
    return Pn


@torch.no_grad()
def reset_opacity(g):
    """reset_opacity (:291-294) through replace_tensor_to_optimizer (:364-381): opacity <- min(opacity, 0.01), moments zeroed."""
    op = torch.sigmoid(g._opacity)
    new = torch.log(torch.clamp_max(op, 0.01) / (1.0 - torch.clamp_max(op, 0.01)))
    p_new = nn.Parameter(new.contiguous().requires_grad_(True))
    opt = g.optimizer
    if opt is not None:
        for grp in opt.param_groups:
            if grp.get("name") == "opacity":
                st = opt.state.pop(grp["params"][0], None)
                grp["params"][0] = p_new
                if st is not None:
                    st["exp_avg"] = torch.zeros_like(new)
                    st["exp_avg_sq"] = torch.zeros_like(new)
                    opt.state[p_new] = st
    g._opacity = p_new
