"""The slice of the reference's scene layer that the training inner loop touches, with the same names.

  GaussianModel   <- R/scene/gaussian_model_dpsr_dynamic_anchor.py:46-149 (parameters, activations, getters),
                     :155-184 (create_from_pcd), :186-236 (training_setup / update_learning_rate),
                     :679-682 (add_densification_stats)
  render()        <- R/gaussian_renderer/__init__.py:32-119 (same signature, same returned dict)
  l1_loss / ssim  <- R/utils/loss_utils.py:18-19, 32-76
(R/ = /root/reference/dgmesh/.)  Densification / pruning: densify.py.  Mesh branch (DiffMC / nvdiffrast) and dataset
readers are out of scope.
"""
import math
from math import exp

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .deform import get_expon_lr_func
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianModel:
    """Parameter store with the attribute surface render() and the optimiser step use."""

    def __init__(self, sh_degree=3, device="cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.device = device
        self._xyz = self._features_dc = self._features_rest = None
        self._scaling = self._rotation = self._opacity = self._normal = None
        self.max_radii2D = self.xyz_gradient_accum = self.denom = None
        self.optimizer = None
        self.spatial_lr_scale = 5
        # mesh branch state that lives on the Gaussian model (gaussian_model_dpsr_dynamic_anchor.py:76-86): the DPSR iso-level, a
        # trained scalar in the optimizer's 8th group, and the normalisation of the points into the unit cube
        self.density_thres_param = torch.nn.Parameter(torch.tensor(0.0, dtype=torch.float32, device=device))
        self.gaussian_center = torch.zeros(3, dtype=torch.float32, device=device)
        self.gaussian_scale = torch.ones(1, dtype=torch.float32, device=device)
        self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = F.normalize

    # -- getters (gaussian_model_dpsr_dynamic_anchor.py:122-149) --
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        """(P, 6) upper triangle [xx, xy, xz, yy, yz, zz] of R S S^T R^T with S = diag(scaling_modifier * get_scaling) and
        R from the (re-normalised) raw quaternion (gaussian_model_dpsr_dynamic_anchor.py:148-149 through
        general_utils.py:130-170)."""
        return covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    @property
    def get_normal(self):
        return self._normal

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def create_from_pcd(self, points, colors, normals=None, generator=None):
        """points (P,3), colors (P,3) in [0,1] (numpy).  Scales from simple-knn exactly as :165-166."""
        from .knn import distCUDA2

        dev = self.device
        pts = torch.tensor(np.asarray(points), dtype=torch.float32, device=dev)
        fused_color = RGB2SH(torch.tensor(np.asarray(colors), dtype=torch.float32, device=dev))
        P = pts.shape[0]
        features = torch.zeros((P, 3, (self.max_sh_degree + 1) ** 2), dtype=torch.float32, device=dev)
        features[:, :3, 0] = fused_color
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.rand((P, 4), device=dev, generator=generator)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.1 * torch.ones((P, 1), dtype=torch.float, device=dev))
        if normals is not None:
            nrm = torch.tensor(np.asarray(normals), dtype=torch.float32, device=dev)
        else:
            nrm = torch.rand((P, 3), device=dev, generator=generator)
        self.load_raw(pts, features[:, :, 0:1].transpose(1, 2).contiguous(),
                      features[:, :, 1:].transpose(1, 2).contiguous(), scales, rots, opacities, nrm)

    def load_raw(self, xyz, f_dc, f_rest, scaling, rotation, opacity, normal=None):
        dev = self.device
        mk = lambda t: nn.Parameter(torch.as_tensor(t, dtype=torch.float32, device=dev).contiguous().requires_grad_(True))
        self._xyz, self._features_dc, self._features_rest = mk(xyz), mk(f_dc), mk(f_rest)
        self._scaling, self._rotation, self._opacity = mk(scaling), mk(rotation), mk(opacity)
        P = self._xyz.shape[0]
        self._normal = mk(normal if normal is not None else torch.zeros(P, 3))
        self.max_radii2D = torch.zeros((P,), device=dev)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation,
                self._normal]

    def training_setup(self, training_args):
        """Adam groups of gaussian_model_dpsr_dynamic_anchor.py:186-212 (eps 1e-15)."""
        P = self._xyz.shape[0]
        self.percent_dense = training_args.percent_dense
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self.device)
        self.denom = torch.zeros((P, 1), device=self.device)
        self.spatial_lr_scale = 5
        lr_xyz = training_args.position_lr_init * self.spatial_lr_scale
        groups = [
            {"params": [self._xyz], "lr": lr_xyz, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr * self.spatial_lr_scale, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
            {"params": [self._normal], "lr": training_args.rotation_lr * 100, "name": "normal"},
            {"params": [self.density_thres_param], "lr": 0.01, "name": "density_thres"},
        ]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        steps = training_args.position_lr_max_steps
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=lr_xyz, lr_final=training_args.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=training_args.position_lr_delay_mult, max_steps=steps)
        self.density_thres_scheduler_args = get_expon_lr_func(lr_init=0.01, lr_final=0.0001, lr_delay_mult=0.01, max_steps=steps)
        # NB the reference applies the "normal" schedule built from rotation_lr and the "rotation" schedule built
        # from 100 * rotation_lr (gaussian_model_dpsr_dynamic_anchor.py:214-236); mirrored as is.
        self.normal_scheduler_args = get_expon_lr_func(lr_init=training_args.rotation_lr,
                                                       lr_final=training_args.rotation_lr * 0.1, lr_delay_mult=0.01,
                                                       max_steps=steps)
        self.rotation_scheduler_args = get_expon_lr_func(lr_init=training_args.rotation_lr * 100,
                                                         lr_final=training_args.rotation_lr * 100 * 0.1,
                                                         lr_delay_mult=0.01, max_steps=steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)
            elif group["name"] == "density_thres":
                group["lr"] = self.density_thres_scheduler_args(iteration)
            elif group["name"] == "normal":
                group["lr"] = self.normal_scheduler_args(iteration)
            elif group["name"] == "rotation":
                group["lr"] = self.rotation_scheduler_args(iteration)

    # -- PLY checkpoints (gaussian_model_dpsr_dynamic_anchor.py:238-289, 296-362): same elements / property names --
    def save_ply(self, path):
        from . import ply_io
        n = lambda t: t.detach().cpu().numpy().astype(np.float32)
        ply_io.save_gaussians(path, n(self._xyz), n(self._normal), n(self._features_dc), n(self._features_rest), n(self._opacity),
                              n(self._scaling), n(self._rotation), float(self.density_thres_param.detach()),
                              tuple(float(v) for v in torch.as_tensor(getattr(self, "gaussian_center", [0.0, 0.0, 0.0])).reshape(-1).tolist()),
                              float(torch.as_tensor(getattr(self, "gaussian_scale", 1.0)).reshape(-1)[0]))

    def load_ply(self, path, og_number_points=-1, iteration=-1):
        """`path` is either a .ply file or a model directory laid out like the reference
        (<path>/point_cloud/iteration_<N>/point_cloud.ply, newest iteration when iteration == -1)."""
        import os
        from . import ply_io
        if os.path.isdir(path):
            root = os.path.join(path, "point_cloud")
            if iteration == -1:  # searchForMaxIteration, R/utils/system_utils.py:29-31
                iteration = max(int(f.split("_")[-1]) for f in os.listdir(root))
            path = os.path.join(root, f"iteration_{iteration}", "point_cloud.ply")
        d = ply_io.load_gaussians(path, self.max_sh_degree)
        self.og_number_points = og_number_points
        self.load_raw(d["xyz"], d["features_dc"], d["features_rest"], d["scaling"], d["rotation"], d["opacity"], d["normal"])
        dev = self.device
        self.density_thres_param = torch.nn.Parameter(torch.tensor(float(d["density_thres"]), dtype=torch.float32, device=dev))
        self.gaussian_center = torch.tensor(d["gaussian_center"], dtype=torch.float32, device=dev)
        self.gaussian_scale = torch.tensor(float(d["gaussian_scale"]), dtype=torch.float32, device=dev)
        P = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)
        self.active_sh_degree = self.max_sh_degree

    # -- densification / pruning / opacity reset (gaussian_model_dpsr_dynamic_anchor.py:291-294, 383-551) on the device --
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None, samples=None):
        from . import densify
        return densify.densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator, samples)

    def prune_points(self, mask):
        from . import densify
        return densify.prune_points(self, mask)

    def reset_opacity(self):
        from . import densify
        densify.reset_opacity(self)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1,
                                                              keepdim=True)
        self.denom[update_filter] += 1

    @torch.no_grad()
    def track_densification_stats(self, viewspace_point_tensor, visibility_filter, radii):
        """The per-iteration bookkeeping of R/train.py:489-496 (max_radii2D of the visible Gaussians, then
        add_densification_stats) written with masks instead of boolean indexing: identical values, but no
        nonzero() and hence no host synchronisation inside the step."""
        g = None if viewspace_point_tensor is None else viewspace_point_tensor.grad
        f32c = lambda t: t.dtype == torch.float32 and t.is_contiguous()
        if radii.is_cuda and radii.dtype == torch.int32 and radii.is_contiguous() and f32c(self.max_radii2D) and \
                (g is None or (f32c(g) and g.shape[1] == 3 and f32c(self.xyz_gradient_accum) and f32c(self.denom))):
            # one launch instead of seven elementwise ones (csrc/densify.hip: densify_stats_kernel)
            from . import _lib
            from .densify import _stream, _vp
            P = radii.shape[0]
            with _lib.device_guard(radii.device):
                _lib.check(_lib.lib().dgm_densify_stats(P, None if g is None else _vp(g), _vp(radii), _vp(self.max_radii2D),
                                                        _vp(self.xyz_gradient_accum), _vp(self.denom), _stream()))
            return
        vis = visibility_filter if visibility_filter is not None else radii > 0
        r = radii.to(self.max_radii2D.dtype)
        self.max_radii2D = torch.where(vis, torch.maximum(self.max_radii2D, r), self.max_radii2D)
        if g is not None:
            self.xyz_gradient_accum += torch.norm(g[:, :2], dim=-1, keepdim=True) * vis.unsqueeze(-1)
            self.denom += vis.unsqueeze(-1).to(self.denom.dtype)


class PipelineParams:
    """R/arguments/__init__.py:95-100."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


class OptimizationParams:
    """The optimiser constants of R/arguments/__init__.py:103-154 that the Gaussian branch uses."""
    iterations = 40_000
    warm_up = 3_000
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 40_000
    apperance_lr_init = 0.00016
    apperance_lr_final = 0.0000016
    apperance_lr_delay_mult = 0.01
    apperance_lr_max_steps = 40_000
    deform_lr_max_steps = 40_000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.001
    rotation_lr = 0.001
    percent_dense = 0.01
    lambda_dssim = 0.2
    densify_from_iter = 500
    densify_until_iter = 15_000
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_grad_threshold = 0.0002
    # mesh co-training phase (R/arguments/__init__.py:109, 142, 148-149; R/train.py:124-127)
    dpsr_iter = 5000
    normal_warm_up = 1_000
    normal_deform_delay = 2000   # NORMAL_WARMUP_ITER of R/train.py:127: deform_normal / deform_back_normal start this long after dpsr_iter
    mask_loss_weight = 10.0
    mesh_img_loss_weight = 1.0


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """Sigma = (R S)(R S)^T as its six upper-triangle entries; rotation is normalised here (w, x, y, z)."""
    q = rotation / rotation.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling).unsqueeze(1)          # R @ diag(s): scales the columns
    cov = L @ L.transpose(1, 2)
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=-1)


_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def sh_basis(deg, dirs):
    """(..., (deg+1)^2) real spherical-harmonics basis at unit directions, in the coefficient order and sign convention of
    the rasterizer's computeColorFromSH (forward.cu:20-71) and of R/utils/sh_utils.py:57-112."""
    if not 0 <= deg <= 3:
        raise ValueError("sh_basis: degrees 0..3 (the Gaussian model never exceeds max_sh_degree = 3)")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, _SH_C0)]
    if deg > 0:
        b += [-_SH_C1 * y, _SH_C1 * z, -_SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [_SH_C2[0] * xy, _SH_C2[1] * yz, _SH_C2[2] * (2.0 * zz - xx - yy), _SH_C2[3] * xz, _SH_C2[4] * (xx - yy)]
    if deg > 2:
        b += [_SH_C3[0] * y * (3 * xx - yy), _SH_C3[1] * xy * z, _SH_C3[2] * y * (4 * zz - xx - yy),
              _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _SH_C3[4] * x * (4 * zz - xx - yy), _SH_C3[5] * z * (xx - yy),
              _SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def eval_sh(deg, sh, dirs):
    """sh (..., C, >= (deg+1)^2), dirs (..., 3) unit vectors -> (..., C): the Python SH path of the reference's render()
    (pipe.convert_SHs_python, R/gaussian_renderer/__init__.py:95-100)."""
    n = (deg + 1) ** 2
    basis = sh_basis(deg, dirs)
    # term by term, in coefficient order: the reference's running sum (R/utils/sh_utils.py:75-101), so that the colours handed to
    # the rasterizer are bit-identical to the reference render()'s (a reduction over the last axis rounds differently)
    out = basis[..., 0:1] * sh[..., 0]
    for k in range(1, n):
        out = out + basis[..., k:k + 1] * sh[..., k]
    return out


def render(viewpoint_camera, pc, pipe, bg_color, d_xyz, d_rotation, d_scaling, is_6dof=False, scaling_modifier=1.0,
           override_color=None, delta=None, lean=False):
    """R/gaussian_renderer/__init__.py:32-119.  viewpoint_camera needs FoVx, FoVy, image_height, image_width,
    world_view_transform, full_proj_transform, camera_center (torch tensors on the GPU).
    delta: optionally the deformation network's raw (P, >= 10) output [d_xyz | d_rotation | d_scaling | ...] instead of
    the three slices; activations + deformation then run as one fused kernel (glue.gaussian_apply).
    lean: the training loop's variant -- `viewspace_points` is an uninitialised leaf (the rasterizer never reads its
    values, only routes dL/dmeans2D into its .grad) and `visibility_filter` is None (track_densification_stats derives it
    from `radii`): three elementwise launches fewer per step."""
    if lean:
        screenspace_points = torch.empty_like(pc.get_xyz).requires_grad_(True)
    else:
        screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True,
                                              device=pc.get_xyz.device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    fused = delta is not None and not is_6dof
    if fused:
        from .glue import gaussian_apply
        means3D, scales, rotations, opacity = gaussian_apply(pc._xyz, pc._scaling, pc._rotation, pc._opacity, delta)
    elif is_6dof:
        if torch.is_tensor(d_xyz) is False:
            means3D = pc.get_xyz
        elif d_xyz.is_cuda and d_xyz.dtype == torch.float32:
            from .glue import se3_transform  # dgm_se3_transform_*: cat + bmm + divide of the reference as one kernel each way
            means3D = se3_transform(d_xyz, pc.get_xyz)
        else:
            hom = torch.cat([pc.get_xyz, torch.ones_like(pc.get_xyz[:, :1])], -1)
            out = torch.bmm(d_xyz, hom.unsqueeze(-1)).squeeze(-1)
            means3D = out[..., :3] / out[..., 3:]
    else:
        means3D = pc.get_xyz + d_xyz
    means2D = screenspace_points
    cov3D_precomp = None
    if not fused:
        opacity = pc.get_opacity
        scales = pc.get_scaling + d_scaling
        rotations = pc.get_rotation + d_rotation
    if getattr(pipe, "compute_cov3D_python", False):  # as in the reference: canonical covariance, deltas not applied
        cov3D_precomp, scales, rotations = pc.get_covariance(scaling_modifier), None, None
    shs = None
    colors_precomp = override_color
    # lean: the two SH tensors go to the kernels unconcatenated (GaussianRasterizer.forward_split_sh)
    split_sh = (lean and colors_precomp is None and not getattr(pipe, "convert_SHs_python", False) and cov3D_precomp is None
                and means3D.is_cuda and pc._features_rest.shape[1] > 0)
    if colors_precomp is None and not split_sh:
        if getattr(pipe, "convert_SHs_python", False):  # as in the reference: view directions from the CANONICAL xyz
            shs_view = pc.get_features.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.reshape(1, 3)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp / dir_pp.norm(dim=1, keepdim=True)) + 0.5, 0.0)
        else:
            shs = pc.get_features
    if split_sh:
        rendered_image, radii = rasterizer.forward_split_sh(means3D, means2D, opacity, pc._features_dc, pc._features_rest,
                                                            scales, rotations)
    else:
        rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                           opacities=opacity, scales=scales, rotations=rotations,
                                           cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": None if lean else radii > 0, "radii": radii, "means3D": means3D}


def l1_loss(network_output, gt):
    return torch.abs((network_output - gt)).mean()


_WINDOWS = {}


def _window(window_size, channel, like):
    key = (window_size, channel, like.device, like.dtype)
    if key not in _WINDOWS:
        g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
        _WINDOWS[key] = w2.expand(channel, 1, window_size, window_size).contiguous().to(like)
    return _WINDOWS[key]


def ssim(img1, img2, window_size=11, size_average=True):
    """R/utils/loss_utils.py:45-76 (the window is cached instead of rebuilt every call)."""
    channel = img1.size(-3)
    window = _window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


class TorchCamera:
    """GPU-resident view of synthetic.Camera with the attribute names of R/scene/cameras.py:18-71."""

    def __init__(self, cam, device, original_image=None):
        self.FoVx, self.FoVy = cam.FoVx, cam.FoVy
        self.image_width, self.image_height = cam.image_width, cam.image_height
        self.world_view_transform = torch.tensor(cam.world_view_transform, device=device)
        self.full_proj_transform = torch.tensor(cam.full_proj_transform, device=device)
        self.camera_center = torch.tensor(cam.camera_center, device=device)
        self.fid = torch.tensor([cam.fid], dtype=torch.float32, device=device)
        self.original_image = None if original_image is None else torch.as_tensor(original_image, device=device).clamp(0.0, 1.0)
