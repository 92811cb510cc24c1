"""The per-iteration training step of DG-Mesh (Gaussian branch) and its frame-parallel data-parallel form.

Single-rank semantics follow R/train.py:129-321 and :517-530 (R/ = /root/reference/dgmesh/):
  lr update -> pick camera -> deform MLP (iteration >= warm_up) -> render -> deform_back MLP + cycle losses ->
  0.8*L1 + 0.2*(1-SSIM) -> backward -> Adam steps (eps 1e-15) -> zero grads.
Densification / pruning / opacity reset run inside the loop when `densify=True` (densify.py).

Mesh co-training phase (`mesh=MeshPhase(...)`, iteration >= dpsr_iter; R/train.py:165-176, 225-235, 243-285, 517-530 and
R/utils/renderer.py:150-183): the two normal networks (deform_normal, deform_back_normal: DeformNetworkNormalSep on the P
Gaussians, cycle loss / 4), the DPSR chain on the deformed points and normals (normalise to the unit cube -> trilinear splat ->
spectral Poisson solve -> sign fix -> minus density threshold), deform_back + appearance on the V mesh vertices, six Adam
steps (+ the density threshold).  What sits between phi and the image losses in the reference -- DiffMC marching cubes and
nvdiffrast, third-party packages outside /root/reference (parity unpinned) -- is NOT rebuilt; in its
place phi is read back trilinearly (grid_interp, with its adjoint) at V fixed probe points and those probes act as the mesh
vertices, with L1 losses against fixed targets standing in for the mask / mesh-image losses.  Every kernel chain of the phase
that the reference owns therefore runs and is differentiated; the numbers of the stand-in losses mean nothing.
Host synchronisations of the reference's loop that do not change results are dropped
(torch.cuda.empty_cache() every iteration, R/train.py:130; get_psnr's .item(), :315).

Data parallelism (absent in the reference; BASELINE.json north_star): one process per GPU, frames shard across
ranks, every rank holds a full replica of the Gaussians and MLPs.
  * all ranks derive the same shuffled camera order from a shared seed; rank r takes entries r, r+W, ... so one
    step consumes W frames (effective batch W);
  * gradients are exchanged as flat fp32 buckets over RCCL/xGMI.  Default: ONE bucket (Gaussian + MLP gradients, 27.8 MB at
    P=100k) all-reduced after backward on the current stream.  `overlap=True` (opt-in): two buckets -- the Gaussian gradients
    (23.6 MB), whose all-reduce is launched from an autograd hook as soon as they are final and runs (on RCCL's own queue) under
    the two MLP backward passes, and the MLP gradients after backward.  It is opt-in until a multi-GPU run has shown the
    replicas bit-identical with it on (bench.py prints `replicas_identical` for both forms); DESIGN.md section 6.
    On the GPU the step's fresh gradients are packed into the bucket by one multi-tensor copy ("pack" mode; no per-tensor
    accumulate kernels, no zero-fill), on the CPU test path every .grad is a view into the bucket ("views" mode, which also
    supports accumulating several frames);
  * every rank then applies the identical Adam update, so replicas stay bit-identical without broadcasting.  On the
    GPU that is one kernel for all three optimizers (optim.MultiAdam).
  W ranks x 1 frame is therefore equivalent to 1 rank accumulating the same W frames before stepping.
"""
import os
import random

import torch
import torch.distributed as dist

from . import rasterizer as _RZ
from . import scene as S


class FlatGradBucket:
    """All gradients of `params` as views into one contiguous buffer (DDP's gradient_as_bucket_view idea)."""

    def __init__(self, params, attach=True):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            if attach:
                p.grad = self.views[-1]
            off += n

    def pack(self):
        """Copy the parameters' current .grad tensors into the bucket (one multi-tensor kernel); absent gradients
        count as zero in the exchange.  Returns {id(param): view} for the parameters that HAD a gradient (a parameter
        without one is not stepped, like torch.optim.Adam)."""
        src, dst, out = [], [], {}
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                src.append(p.grad)
                dst.append(v)
                out[id(p)] = v
        if src:
            torch._foreach_copy_(dst, src)
        return out

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


_PERM_CACHE = {}

# R/train.py:127: deform_normal / deform_back_normal start this many iterations after dpsr_iter.  A module constant in the
# reference, not a field of its OptimizationParams -- so a reference config object does not carry it.
NORMAL_WARMUP_ITER = 2000


def normal_deform_delay(opt):
    """`opt.normal_deform_delay` when the config object has one (this repo's OptimizationParams), else the reference's constant."""
    return getattr(opt, "normal_deform_delay", NORMAL_WARMUP_ITER)


def frame_schedule(n_frames, step, rank, world, seed=0):
    """Index of the camera rank `rank` renders at `step`: a shared-seed shuffle per epoch, strided by rank."""
    per_epoch = max(n_frames // world, 1)
    epoch, k = divmod(step, per_epoch)
    key = (n_frames, seed, epoch)
    perm = _PERM_CACHE.get(key)
    if perm is None:  # one shuffle per epoch, not per step
        perm = list(range(n_frames))
        random.Random(seed * 1000003 + epoch).shuffle(perm)
        _PERM_CACHE.clear()
        _PERM_CACHE[key] = perm
    return perm[(k * world + rank) % n_frames]


class MeshPhase:
    """The mesh co-training phase's own state: the three extra networks, the DPSR module and the V probe points that stand in for
    the DiffMC vertices.  The density threshold (the 8th Adam group of the Gaussian model, exponential schedule 0.01 -> 1e-4)
    and the normalisation gaussian_center / gaussian_scale live on the GaussianModel, as in the reference
    (R/scene/gaussian_model_dpsr_dynamic_anchor.py:76-86, 201-229), so checkpoints carry the trained values; `density_thres`,
    `center`, `scale` given here initialise them (R/train.py: normal_initialization sets init_density_threshold)."""

    def __init__(self, deform_normal, deform_back_normal, appearance, dpsr=None, n_verts=20000, density_thres=None,
                 center=None, scale=None, seed=0, device="cuda", stand_in_weight=1e-6):
        self.stand_in_weight = stand_in_weight
        self.deform_normal, self.deform_back_normal, self.appearance, self.dpsr = deform_normal, deform_back_normal, appearance, dpsr
        dev = torch.device(device)
        gen = torch.Generator().manual_seed(4242 + seed)
        self.init = {"density_thres": density_thres, "center": center, "scale": scale}
        # probe points in the unit cube (a shell around the centre, where an iso-surface would lie) and fixed targets for the
        # stand-in losses; their world positions follow from the Gaussian model's normalisation (bind())
        d = torch.randn(n_verts, 3, generator=gen)
        d = d / d.norm(dim=1, keepdim=True)
        self.probes = (0.5 + 0.25 * d * (1 + 0.1 * torch.randn(n_verts, 1, generator=gen))).clamp(0.02, 0.98).to(dev)
        self.verts = None
        self.phi_target = torch.zeros(n_verts, device=dev)
        self.color_target = torch.rand(n_verts, 3, generator=gen).to(dev)

    def bind(self, g):
        """Applies the initial values to the Gaussian model and places the probe vertices in world space."""
        with torch.no_grad():
            if self.init["density_thres"] is not None:
                g.density_thres_param.fill_(float(self.init["density_thres"]))
            if self.init["center"] is not None:
                g.gaussian_center = torch.tensor(self.init["center"], dtype=torch.float32, device=g.gaussian_center.device)
            if self.init["scale"] is not None:
                g.gaussian_scale = torch.tensor([float(self.init["scale"])], dtype=torch.float32, device=g.gaussian_scale.device)
        self.verts = ((self.probes * 2.0 - 1.0) * g.gaussian_scale + g.gaussian_center).contiguous()

    def networks(self):
        return [self.deform_normal, self.deform_back_normal, self.appearance]


class Trainer:
    def __init__(self, gaussians, deform, deform_back, cameras, opt=None, pipe=None, background=None,
                 is_blender=True, is_6dof=False, rank=0, world=1, seed=0, render_fn=None, fused_adam=None,
                 process_group=None, fused_loss=True, fused_glue=None, track_stats=True, densify=False,
                 cameras_extent=1.0, prune_threshold=0.005, white_background=True, overlap=False, mesh=None):
        self.g, self.deform, self.deform_back = gaussians, deform, deform_back
        self.mesh = mesh
        self.cameras = cameras
        self.opt = opt or S.OptimizationParams()
        self.pipe = pipe or S.PipelineParams()
        self.bg = background
        self.is_blender, self.is_6dof = is_blender, is_6dof
        self.rank, self.world, self.seed = rank, world, seed
        self.render_fn = render_fn or S.render
        self.group = process_group
        self.fused_loss = fused_loss
        self.track_stats = track_stats
        # densification / pruning / opacity reset inside the loop (R/train.py:488-515); off = fixed-P steps (bench.py)
        self.densify = densify
        self.overlap = overlap
        self.cameras_extent, self.prune_threshold, self.white_background = cameras_extent, prune_threshold, white_background
        self.step_count = 0
        # fused per-Gaussian glue (activations + deformation, cycle loss): GPU, stock render(), plain (non-6dof) networks
        self.fused_glue = bool(fused_glue) if fused_glue is not None else (
            gaussians.get_xyz.is_cuda and render_fn is None and not is_6dof)
        dev = gaussians.get_xyz.device
        fused = (dev.type == "cuda") if fused_adam is None else fused_adam
        if mesh is not None:
            mesh.bind(gaussians)
        gaussians.training_setup(self.opt)
        deform.train_setting(self.opt)
        deform_back.train_setting(self.opt)
        self.optimizers = [gaussians.optimizer, deform.optimizer, deform_back.optimizer]
        if mesh is not None:  # R/train.py:517-524: six optimizers (the density threshold is the Gaussian optimizer's 8th group)
            for m in mesh.networks():
                m.train_setting(self.opt)
                self.optimizers.append(m.optimizer)
        self.multi_adam = None
        if fused:  # same update rule, ONE kernel for every tensor of the three optimizers
            from .optim import MultiAdam
            self.multi_adam = MultiAdam(self.optimizers)
        self.pack = self.multi_adam is not None
        self._bind_parameters()
        # normal samples of densify_and_split: one generator per rank, seeded alike, advanced in lockstep
        self.densify_generator = None
        if dev.type == "cuda":
            self.densify_generator = torch.Generator(device=dev)
            self.densify_generator.manual_seed(1234567 + seed)
        self.time_interval = 1.0 / max(len(cameras), 1)
        from .deform import get_linear_noise_func
        self.smooth_term = get_linear_noise_func(lr_init=0.1, lr_final=1e-15, lr_delay_mult=0.01, max_steps=20000)

    @staticmethod
    def freeze_gc():
        """Call once after set-up, before a long run of step().  A full pass of Python's cyclic collector walks every tracked
        object of the process -- 267 k with torch imported: ~80 ms, measured -- and a training loop allocates enough containers
        to trigger one every few hundred steps (25 steps' worth of time; one landing in a 60-step timed region turned 275 it/s
        into 200).  `gc.freeze()` moves everything alive now into the permanent generation: later collections only look at
        what the loop itself creates."""
        import gc
        gc.collect()
        gc.freeze()

    def _bind_parameters(self):
        """(Re)collect the parameters that receive gradients and (re)build the flat gradient bucket; called at start and
        after every change of the Gaussian set (densify / prune / opacity reset replace Parameter objects)."""
        gaussians, deform, deform_back = self.g, self.deform, self.deform_back
        # parameters that receive gradients in the Gaussian branch (the normal parameter is only used by the
        # mesh branch; leaving its .grad None mirrors zero_grad(set_to_none=True))
        params = [gaussians._xyz, gaussians._features_dc, gaussians._features_rest, gaussians._opacity,
                  gaussians._scaling, gaussians._rotation]
        params += list(deform.net.parameters()) + list(deform_back.net.parameters())
        if self.mesh is not None:  # the five networks of SURVEY.md section 8(e)'s bucket, the normals, the density threshold
            for m in self.mesh.networks():
                params += list(m.net.parameters())
            params += [gaussians._normal, gaussians.density_thres_param]
        self.params = [p for p in params if p.requires_grad]
        # "pack": gradients are fresh tensors every step (no accumulate kernels); "views": .grad lives in the bucket
        self.bucket = FlatGradBucket(params, attach=not self.pack) if (not self.pack or self.world > 1) else None
        # Overlap (pack mode, world > 1): the Gaussian gradients (6 tensors, 23.6 MB at P = 100k) are final as soon as the
        # rasterizer / glue backward has run, long before the two MLP backward passes finish.  A post-accumulate hook
        # counts them down and launches their all-reduce asynchronously (RCCL's own stream) while autograd is still in the
        # MLPs; the MLP bucket follows after backward, and both are awaited right before the Adam launch.
        self._early = None
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks = []
        if self.pack and self.world > 1 and self.overlap:
            gp = [p for p in params[:6] if p.requires_grad]
            mp = [p for p in params[6:] if p.requires_grad]
            self._early = {"g": FlatGradBucket(gp, attach=False), "m": FlatGradBucket(mp, attach=False), "left": 0, "work": None,
                           "views": None, "n": len(gp), "armed": False}
            for p in gp:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._gaussian_grad_ready))

    def _gaussian_grad_ready(self, _param):
        e = self._early
        if e is None or not e["armed"]:
            return
        e["left"] -= 1
        if e["left"] == 0:
            e["views"] = e["g"].pack()
            e["work"] = dist.all_reduce(e["g"].flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def maybe_densify(self, iteration):
        """R/train.py:499-515: every densification_interval iterations after densify_from_iter clone / split / prune
        (statistics of all ranks' frames summed first), and the periodic opacity reset.  Returns True if the Gaussian
        set was replaced (their gradients of this iteration are dropped, exactly like the reference, where the new
        Parameters have .grad None when optimizer.step() runs)."""
        opt, g = self.opt, self.g
        changed = False
        if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
            self.sync_densification_stats()
            size_threshold = 20 if iteration > opt.opacity_reset_interval else None
            g.densify_and_prune(opt.densify_grad_threshold, self.prune_threshold, self.cameras_extent, size_threshold,
                                generator=self.densify_generator)
            changed = True
        if iteration % opt.opacity_reset_interval == 0 or (self.white_background and iteration == opt.densify_from_iter):
            g.reset_opacity()
            changed = True
        return changed

    def sync_densification_stats(self):
        """Before a densify/prune decision every rank must see the statistics of ALL frames:
        gradient accumulators and visit counts add up, the screen-space radius bound is a maximum."""
        if self.world > 1:
            dist.all_reduce(self.g.xyz_gradient_accum, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(self.g.denom, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(self.g.max_radii2D, op=dist.ReduceOp.MAX, group=self.group)

    def state_hash(self):
        """One 64-bit word per state tensor -- every parameter in `self.params` and its two Adam moments -- computed on the
        device: the bit patterns as int32, weighted by (position-dependent odd multipliers) and summed with wrap-around.
        Equal tensors give equal words; a single differing bit changes the word (weights are odd)."""
        dev = self.g.get_xyz.device
        words = []
        for p in self.params:
            st = None
            for o in self.optimizers:
                if p in o.state:
                    st = o.state[p]
                    break
            for t in (p.detach(), st.get("exp_avg") if st else None, st.get("exp_avg_sq") if st else None):
                if t is None or t.numel() == 0:
                    words.append(torch.zeros((), dtype=torch.int64, device=dev))
                    continue
                bits = t.contiguous().view(torch.int32).reshape(-1).to(torch.int64)
                w = (torch.arange(bits.numel(), device=dev, dtype=torch.int64) * 2654435761 + 0x1E3779B97F4A7C15) | 1
                words.append((bits * w).sum())
        return torch.stack(words)

    def replicas_identical(self):
        """All ranks hold bit-identical parameters and Adam moments (all-gather of state_hash()); True on one rank."""
        h = self.state_hash()
        if self.world == 1:
            return True
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        return all(bool(torch.equal(parts[0], q)) for q in parts[1:])

    def bucket_bytes(self):
        """Byte sizes of the flat buckets one step all-reduces (Gaussian bucket, MLP bucket; or the single bucket)."""
        if self._early is not None:
            return [self._early[k].nbytes() for k in ("g", "m")]
        return [self.bucket.nbytes()] if self.bucket is not None else [self.grad_bytes()]

    def grad_bytes(self):
        """Size of the all-reduce payload (all gradients, fp32)."""
        return sum(p.numel() for p in self.params) * 4

    def time_input(self, cam, N, iteration):
        """fid (+ annealed noise for real scenes, R/train.py:158-166 and 208-216) for all N rows.  The noise is one
        sample per call, shared by every row, so it is added to the single time value BEFORE expanding: same values as
        the reference's expand-then-add, but the tensor stays a stride-0 view (the MLPs' one-row time branch)."""
        t = cam.fid.reshape(1, 1)
        if not self.is_blender:
            t = t + torch.randn(1, 1, device=t.device) * self.time_interval * self.smooth_term(iteration)
        return t.expand(N, -1)

    def loss_terms(self, cam, iteration):
        """The iteration's loss terms and the render package."""
        g, opt = self.g, self.opt
        delta = None
        if iteration < opt.warm_up:
            d_xyz, d_rotation, d_scaling = 0.0, 0.0, 0.0
        else:
            N = g.get_xyz.shape[0]
            time_input = self.time_input(cam, N, iteration)   # fid + this iteration's first noise sample (R/train.py:158-166)
            t_back = self.time_input(cam, N, iteration)       # ... and its second one, for the backward networks (:208-216)
            if self.fused_glue:  # raw (N, 13) head output straight into the fused glue kernels (glue.py)
                delta = self.deform.step_raw(g.get_xyz.detach(), time_input)
            if delta is None:
                d_xyz, d_rotation, d_scaling = self.deform.step(g.get_xyz.detach(), time_input)[:3]
        losses = {}
        if delta is not None:
            from .glue import cycle_loss
            lean = {"lean": True} if self.render_fn is S.render else {}
            if self.world > 1 and self._early is not None:
                # Data parallel with the early Gaussian-bucket all-reduce: build the cycle branch BEFORE the render branch.
                # Autograd runs later-built branches first, so the rasterizer's backward -- after which the Gaussian
                # gradients are final and their all-reduce starts -- then precedes BOTH MLP backward passes instead of only
                # the deformation network's: twice the window to hide the exchange in.  Same values (the deformed means are
                # the same fp32 sum the glue kernel forms; the two gradients of `delta` commute).
                means = (g.get_xyz.detach() + delta.detach()[:, :3]).contiguous()
                back = self.deform_back.step_raw(means, t_back)
                losses["cycle_loss"] = cycle_loss(delta, back)
                pkg = self.render_fn(cam, g, self.pipe, self.bg, None, None, None, self.is_6dof, delta=delta, **lean)
            else:
                pkg = self.render_fn(cam, g, self.pipe, self.bg, None, None, None, self.is_6dof, delta=delta, **lean)
                back = self.deform_back.step_raw(pkg["means3D"].detach(), t_back)
                losses["cycle_loss"] = cycle_loss(delta, back)
        else:
            pkg = self.render_fn(cam, g, self.pipe, self.bg, d_xyz, d_rotation, d_scaling, self.is_6dof)
            if iteration >= opt.warm_up:
                deformed_xyz = g.get_xyz + d_xyz
                back = self.deform_back.step(deformed_xyz.detach(), t_back)
                cycle = (S.l1_loss(-back[0], d_xyz) + S.l1_loss(-back[1], d_rotation) + S.l1_loss(-back[2], d_scaling)) / 3.0
                losses["cycle_loss"] = cycle
        if self.mesh is not None and iteration >= opt.dpsr_iter:
            self.mesh_terms(cam, iteration, losses, pkg, delta if delta is not None else None,
                            None if delta is not None else (d_xyz if iteration >= opt.warm_up else None), time_input, t_back)
        image = pkg["render"]
        gt = cam.original_image
        if image.is_cuda and self.fused_loss:  # same value, two HIP kernels instead of 5 convs + autograd
            from .loss import image_loss
            losses["img_loss"] = image_loss(image, gt, opt.lambda_dssim)
        else:
            Ll1 = S.l1_loss(image, gt)
            losses["img_loss"] = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - S.ssim(image, gt))
        return losses, pkg

    def mesh_terms(self, cam, iteration, losses, pkg, delta, d_xyz, t_fwd, t_back):
        """The mesh co-training additions of one iteration (see the module docstring); adds to `losses` in place.  The normal
        networks see the time inputs of their position networks -- deform_normal the deformation's noise sample, deform_back_normal
        the backward network's (R/train.py:167-175, 225-228) -- and the vertex queries the noise-free fid (R/utils/renderer.py:177)."""
        g, opt, ms = self.g, self.opt, self.mesh
        N = g.get_xyz.shape[0]
        xyz_d = g.get_xyz.detach()
        normal_nets = iteration >= opt.dpsr_iter + normal_deform_delay(opt)
        d_normal = ms.deform_normal.step(xyz_d, t_fwd) if normal_nets else None         # R/train.py:170-175
        if normal_nets:                                                                   # R/train.py:225-235: cycle / 4
            d_normal_back = ms.deform_back_normal.step(xyz_d, t_back)
            l_n = S.l1_loss(-d_normal_back, d_normal)
            if "cycle_loss" in losses:
                losses["cycle_loss"] = losses["cycle_loss"] * 0.75 + l_n * 0.25
        if ms.dpsr is None:  # (CPU test path: the networks and the bucket, no HIP DPSR)
            if normal_nets:
                losses["normal_reg"] = (g.get_normal + d_normal).abs().mean() * 1e-3
            return
        # R/utils/renderer.py:150-170: points of the deformed Gaussians in the unit cube, their normals, phi, sign, threshold
        dx = delta[:, :3] if delta is not None else d_xyz
        freeze_pos = iteration < opt.dpsr_iter + opt.normal_warm_up
        pts = (xyz_d + dx.detach()) if freeze_pos else (g.get_xyz + dx)
        pts = ((pts - g.gaussian_center) / g.gaussian_scale) / 2.0 + 0.5
        pts = torch.clamp(pts, 1e-6, 1 - 1e-6)
        normals = g.get_normal + d_normal if d_normal is not None else g.get_normal
        psr = ms.dpsr(pts.unsqueeze(0), normals.unsqueeze(0))
        sign = torch.where(psr[0, 0, 0, 0].detach() < 0, -1.0, 1.0)                       # (no host read-back of the sign)
        psr = psr * sign - g.density_thres_param
        # stand-in for DiffMC -> nvdiffrast: phi at the V probe points; the probes act as the mesh vertices
        from .dpsr import grid_interp
        phi_v = grid_interp(psr.unsqueeze(-1), ms.probes.unsqueeze(0))[0, :, 0]
        # (stand-in losses carry a tiny weight: they exist to drive the chain's backward, not to shape the scene)
        losses["mask_loss"] = S.l1_loss(phi_v, ms.phi_target) * 100 * opt.mask_loss_weight * ms.stand_in_weight
        V = ms.verts.shape[0]
        t_v = cam.fid.reshape(1, 1).expand(V, -1)
        back_v = self.deform_back.step(ms.verts, t_v)[0]                                  # R/utils/renderer.py:179-181
        vtx_color = ms.appearance.step(ms.verts + back_v, t_v)
        losses["mesh_img_loss"] = S.l1_loss(vtx_color, ms.color_target) * opt.mesh_img_loss_weight * (1e3 * ms.stand_in_weight)

    def step(self, iteration):
        g = self.g
        g.update_learning_rate(iteration)
        self.deform.update_learning_rate(iteration)
        self.deform_back.update_learning_rate(iteration)
        if self.mesh is not None:
            for m in self.mesh.networks():
                m.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            g.oneupSHdegree()
        self.last_frame = frame_schedule(len(self.cameras), self.step_count, self.rank, self.world, self.seed)
        cam = self.cameras[self.last_frame]
        # Sync-free rasterizer forward (rasterizer.SYNC_FREE / DGM_SYNC_FREE=1): the forward only enqueues; whether the frame fitted
        # the binning buffer's capacity is looked at here, AFTER the backward has been enqueued -- the event has long fired, the host
        # does not wait and the GPU never idles -- and BEFORE anything consumes the gradients: a frame that did not fit was neutralised
        # on the device, so its gradients are discarded and the frame is rendered again with the raised capacity.  (With the
        # Gaussian bucket's early all-reduce armed the check stays inside the forward call: a discarded backward must not have
        # started a collective the other ranks do not repeat.)
        defer = _RZ.SYNC_FREE and not (self.world > 1 and self._early is not None)
        for _attempt in range(4):
            if self.pack:
                for p in self.params:
                    p.grad = None
            else:
                self.bucket.zero()
            _RZ.DEFER_SETTLE = defer
            try:
                losses, pkg = self.loss_terms(cam, iteration)
                terms = list(losses.values())
                loss = terms[0]
                for t in terms[1:]:  # (not sum(): its 0 + ... start is one more launch)
                    loss = loss + t
                if self._early is not None:
                    self._early.update(left=self._early["n"], work=None, views=None, armed=True)
                unit = getattr(self, "_unit_grad", None)  # (backward()'s implicit ones_like(loss) is a fill launch per step)
                if unit is None or unit.shape != loss.shape or unit.dtype != loss.dtype or unit.device != loss.device:
                    unit = self._unit_grad = torch.ones_like(loss)
                loss.backward(unit)
                if self._early is not None:
                    self._early["armed"] = False
            finally:
                _RZ.DEFER_SETTLE = False
            if not defer or _RZ.settle():
                break
            self.redone_frames = getattr(self, "redone_frames", 0) + 1
        else:
            raise RuntimeError("rasterizer capacity: a frame overflowed four times in a row")
        rebound = False
        if self.track_stats and iteration < self.opt.densify_until_iter:  # R/train.py:488-496
            g.track_densification_stats(pkg.get("viewspace_points"), pkg.get("visibility_filter"), pkg["radii"])
            if self.densify:
                rebound = self.maybe_densify(iteration)
        grads = None
        ev = getattr(self, "exchange_events", None)  # bench.py: [(start, end), ...] hipEvents around the exchange on this stream
        if ev is not None and self.world > 1:
            ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            ev[-1][0].record()
        if self.world > 1 and self._early is not None:
            e = self._early
            if e["work"] is None:  # (a step in which some Gaussian tensor got no gradient: exchange it now)
                e["views"] = e["g"].pack()
                e["work"] = dist.all_reduce(e["g"].flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            grads = e["m"].pack()
            dist.all_reduce(e["m"].flat, op=dist.ReduceOp.SUM, group=self.group)
            e["work"].wait()
            grads.update(e["views"])  # replaced Parameters are in neither bucket's dict: no update for them
        elif self.world > 1:
            if self.pack:
                grads = self.bucket.pack()  # replaced Parameters are not in this (old) bucket: no update for them
            self.bucket.all_reduce(self.group)
        if ev is not None and self.world > 1:
            ev[-1][1].record()
        if self.multi_adam is not None:
            self.multi_adam.step(grads)
        else:
            for o in self.optimizers:
                o.step()
        if rebound:
            self._bind_parameters()
        self.step_count += 1
        return loss.detach(), pkg
