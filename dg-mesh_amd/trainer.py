"""The per-iteration training step of DG-Mesh (Gaussian branch) and its frame-parallel data-parallel form.

Single-rank semantics follow R/train.py:129-321 and :517-530 (R/ = /root/reference/dgmesh/):
  lr update -> pick camera -> deform MLP (iteration >= warm_up) -> render -> deform_back MLP + cycle losses ->
  0.8*L1 + 0.2*(1-SSIM) -> backward -> Adam steps (eps 1e-15) -> zero grads.
The mesh branch (iteration >= dpsr_iter: DPSR / DiffMC / nvdiffrast) is out of scope, and
host synchronisations of the reference's loop that do not change results are dropped
(torch.cuda.empty_cache() every iteration, R/train.py:130; get_psnr's .item(), :315).

Data parallelism (absent in the reference; BASELINE.json north_star): one process per GPU, frames shard across
ranks, every rank holds a full replica of the Gaussians and MLPs.
  * all ranks derive the same shuffled camera order from a shared seed; rank r takes entries r, r+W, ... so one
    step consumes W frames (effective batch W);
  * gradients live in ONE flat fp32 bucket (every .grad is a view into it), so the exchange is a single
    all-reduce(SUM) per step over RCCL/xGMI -- 35 MB at P=100k: latency-bound, one collective beats many;
  * every rank then applies the identical Adam update, so replicas stay bit-identical without broadcasting.
  W ranks x 1 frame is therefore equivalent to 1 rank accumulating the same W frames before stepping.
"""
import random

import torch
import torch.distributed as dist

from . import scene as S


class FlatGradBucket:
    """All gradients of `params` as views into one contiguous buffer (DDP's gradient_as_bucket_view idea)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


def frame_schedule(n_frames, step, rank, world, seed=0):
    """Index of the camera rank `rank` renders at `step`: a shared-seed shuffle per epoch, strided by rank."""
    per_epoch = max(n_frames // world, 1)
    epoch, k = divmod(step, per_epoch)
    perm = list(range(n_frames))
    random.Random(seed * 1000003 + epoch).shuffle(perm)
    return perm[(k * world + rank) % n_frames]


class Trainer:
    def __init__(self, gaussians, deform, deform_back, cameras, opt=None, pipe=None, background=None,
                 is_blender=True, is_6dof=False, rank=0, world=1, seed=0, render_fn=None, fused_adam=None,
                 process_group=None, fused_loss=True):
        self.g, self.deform, self.deform_back = gaussians, deform, deform_back
        self.cameras = cameras
        self.opt = opt or S.OptimizationParams()
        self.pipe = pipe or S.PipelineParams()
        self.bg = background
        self.is_blender, self.is_6dof = is_blender, is_6dof
        self.rank, self.world, self.seed = rank, world, seed
        self.render_fn = render_fn or S.render
        self.group = process_group
        self.fused_loss = fused_loss
        self.step_count = 0
        dev = gaussians.get_xyz.device
        fused = (dev.type == "cuda") if fused_adam is None else fused_adam
        gaussians.training_setup(self.opt)
        deform.train_setting(self.opt)
        deform_back.train_setting(self.opt)
        if fused:  # same update rule, one multi-tensor kernel per optimizer instead of several per tensor
            for o in (gaussians, deform, deform_back):
                for group in o.optimizer.param_groups:
                    group["fused"], group["foreach"] = True, False
        self.optimizers = [gaussians.optimizer, deform.optimizer, deform_back.optimizer]
        # parameters that receive gradients in the Gaussian branch (the normal parameter is only used by the
        # mesh branch; leaving its .grad None mirrors zero_grad(set_to_none=True))
        params = [gaussians._xyz, gaussians._features_dc, gaussians._features_rest, gaussians._opacity,
                  gaussians._scaling, gaussians._rotation]
        params += list(deform.net.parameters()) + list(deform_back.net.parameters())
        self.bucket = FlatGradBucket(params)
        self.time_interval = 1.0 / max(len(cameras), 1)

    def loss_terms(self, cam, iteration):
        g, opt = self.g, self.opt
        if iteration < opt.warm_up:
            d_xyz, d_rotation, d_scaling = 0.0, 0.0, 0.0
        else:
            N = g.get_xyz.shape[0]
            time_input = cam.fid.unsqueeze(0).expand(N, -1)
            d_xyz, d_rotation, d_scaling = self.deform.step(g.get_xyz.detach(), time_input)[:3]
        pkg = self.render_fn(cam, g, self.pipe, self.bg, d_xyz, d_rotation, d_scaling, self.is_6dof)
        image = pkg["render"]
        losses = {}
        if iteration >= opt.warm_up:
            deformed_xyz = g.get_xyz + d_xyz
            back = self.deform_back.step(deformed_xyz.detach(), time_input)
            cycle = (S.l1_loss(-back[0], d_xyz) + S.l1_loss(-back[1], d_rotation) + S.l1_loss(-back[2], d_scaling)) / 3.0
            losses["cycle_loss"] = cycle
        gt = cam.original_image
        if image.is_cuda and self.fused_loss:  # same value, two HIP kernels instead of 5 convs + autograd
            from .loss import image_loss
            losses["img_loss"] = image_loss(image, gt, opt.lambda_dssim)
        else:
            Ll1 = S.l1_loss(image, gt)
            losses["img_loss"] = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - S.ssim(image, gt))
        return losses, pkg

    def step(self, iteration):
        g = self.g
        g.update_learning_rate(iteration)
        self.deform.update_learning_rate(iteration)
        self.deform_back.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            g.oneupSHdegree()
        cam = self.cameras[frame_schedule(len(self.cameras), self.step_count, self.rank, self.world, self.seed)]
        self.bucket.zero()
        losses, pkg = self.loss_terms(cam, iteration)
        loss = sum(losses.values())
        loss.backward()
        if self.world > 1:
            self.bucket.all_reduce(self.group)
        for o in self.optimizers:
            o.step()
        self.step_count += 1
        return loss.detach(), pkg
