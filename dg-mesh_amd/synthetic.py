"""Synthetic D-NeRF-like workloads (seeded) shared by tests, bench.py and smoke().

There is no dataset on the build or GPU box, so every measured / tested input is generated
here.  The generators mirror how the reference initialises a scene:

* camera matrices: R/utils/graphics_utils.py:42-100 (getWorld2View2, getProjectionMatrix,
  getProjectionMatrix_from_K) and R/scene/cameras.py:54-71 (world_view_transform = W2C^T,
  full_proj_transform = W2C^T @ P^T, camera_center = inverse(W2C^T)[3,:3]; znear .01, zfar 100);
* Gaussians: R/scene/dataset_readers.py:332-336 (xyz ~ U[-1.3,1.3]^3) and
  R/scene/gaussian_model_dpsr_dynamic_anchor.py:155-184 (scale = sqrt(mean 3-NN d2), rot=(1,U,U,U)
  normalised, opacity 0.1, SH DC from RGB2SH, rest ~ small noise).

(R/ = /root/reference/dgmesh/.)  numpy only; no torch, no GPU.
"""
import math
from typing import NamedTuple

import numpy as np

C0 = 0.28209479177387814  # R/utils/sh_utils.py:26


class Camera(NamedTuple):
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: np.ndarray  # (4,4) float32, = W2C^T  (column-major for the kernels)
    full_proj_transform: np.ndarray  # (4,4) float32, = W2C^T @ P^T
    camera_center: np.ndarray  # (3,) float32
    fid: float


def _look_at_w2c(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """World-to-camera with +z forward, +x right, +y down (COLMAP convention used by the reference)."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    Rwc = np.stack([r, d, f], 0)  # rows = camera axes in world coords
    w2c = np.eye(4)
    w2c[:3, :3] = Rwc
    w2c[:3, 3] = -Rwc @ eye
    return w2c


def projection_matrix(znear, zfar, fovX, fovY):
    """R/utils/graphics_utils.py:56-76."""
    tanY, tanX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def projection_matrix_from_K(znear, zfar, K, W, H):
    """R/utils/graphics_utils.py:79-100 (off-centre principal point)."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    top = znear * cy / fy
    bottom = -znear * (H - cy) / fy
    right = znear * (W - cx) / fx
    left = -znear * cx / fx
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = -(right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, azimuth=0.3, elevation=0.35, radius=4.0, fovx=0.6911, fid=0.0, K=None):
    eye = radius * np.array([math.cos(elevation) * math.cos(azimuth), math.cos(elevation) * math.sin(azimuth),
                             math.sin(elevation)])
    w2c = np.float32(_look_at_w2c(eye))
    wvt = np.ascontiguousarray(w2c.T)
    if K is not None:
        P = projection_matrix_from_K(0.01, 100.0, K, W, H)
        fovx = 2 * math.atan(W / (2 * K[0, 0]))
        fovy = 2 * math.atan(H / (2 * K[1, 1]))
    else:
        fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
        P = projection_matrix(0.01, 100.0, fovx, fovy)
    full = np.ascontiguousarray((wvt @ P.T).astype(np.float32))
    center = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return Camera(W, H, fovx, fovy, wvt.astype(np.float32), full, center, float(fid))


def orbit_cameras(n, W, H, seed=0, **kw):
    rng = np.random.RandomState(seed)
    cams = []
    for i in range(n):
        az = 2 * math.pi * i / max(n, 1) + 0.1 * rng.rand()
        el = 0.15 + 0.5 * rng.rand()
        cams.append(make_camera(W, H, azimuth=az, elevation=el, fid=i / max(n, 1), **kw))
    return cams


def brute_knn_dist2(xyz):
    """Exact mean squared distance to the 3 nearest neighbours (host helper for small scenes)."""
    from scipy.spatial import cKDTree

    d, _ = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), k=4)
    return (d[:, 1:] ** 2).mean(1).astype(np.float32)


def make_gaussians(P, seed=0, kind="init", extent=1.3, dist2=None, sh_degree=3):
    """Raw (pre-activation) Gaussian parameters, laid out like the reference's nn.Parameters.

    kind = "init"    : reference initialisation (isotropic, opacity 0.1, rot (1,U,U,U))
         = "aniso"   : init + per-axis log-normal scale jitter and random rotations
         = "trained" : compact shell, small scales, opacity U[0.5,0.99] (early termination)
    Returns dict of float32 arrays: xyz (P,3), features_dc (P,1,3), features_rest (P,15,3),
    scaling (P,3) [log], rotation (P,4) [raw], opacity (P,1) [logit].
    """
    rng = np.random.RandomState(seed)
    if kind == "trained":
        v = rng.randn(P, 3)
        v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-12
        xyz = (v * (0.8 + 0.02 * rng.randn(P, 1))).astype(np.float32)
    else:
        xyz = ((rng.rand(P, 3) * 2 - 1) * extent).astype(np.float32)
    if dist2 is None:
        dist2 = brute_knn_dist2(xyz)
    dist2 = np.maximum(dist2, 1e-7).astype(np.float32)
    scaling = np.repeat(np.log(np.sqrt(dist2))[:, None], 3, 1).astype(np.float32)
    rot = rng.rand(P, 4).astype(np.float32)
    rot[:, 0] = 1
    if kind == "aniso":
        scaling = scaling + (0.5 * rng.randn(P, 3)).astype(np.float32)
        rot = rng.randn(P, 4).astype(np.float32)
    if kind == "trained":
        scaling = np.log(np.full((P, 3), 0.01, np.float32) * np.exp(0.3 * rng.randn(P, 3)).astype(np.float32))
        opacity_act = (0.5 + 0.49 * rng.rand(P, 1)).astype(np.float32)
    else:
        opacity_act = np.full((P, 1), 0.1, np.float32)
    opacity = np.log(opacity_act / (1 - opacity_act)).astype(np.float32)
    n_rest = (sh_degree + 1) ** 2 - 1
    f_dc = ((rng.rand(P, 1, 3) - 0.5) / C0).astype(np.float32)
    f_rest = (0.05 * rng.randn(P, n_rest, 3)).astype(np.float32)
    return dict(xyz=xyz, features_dc=f_dc, features_rest=f_rest, scaling=scaling, rotation=rot, opacity=opacity)


def activate(g, d_xyz=0.0, d_rot=0.0, d_scale=0.0):
    """Activations + deltas exactly as render() applies them (R/gaussian_renderer/__init__.py:75-102,
    R/scene/gaussian_model_dpsr_dynamic_anchor.py:122-149).  numpy float32."""
    rot = g["rotation"] / np.maximum(np.linalg.norm(g["rotation"], axis=1, keepdims=True), 1e-12)
    return dict(
        means3D=(g["xyz"] + d_xyz).astype(np.float32),
        scales=(np.exp(g["scaling"]) + d_scale).astype(np.float32),
        rotations=(rot + d_rot).astype(np.float32),
        opacities=(1.0 / (1.0 + np.exp(-g["opacity"]))).astype(np.float32),
        shs=np.concatenate([g["features_dc"], g["features_rest"]], 1).astype(np.float32))


def gt_image(W, H, seed=0):
    """Stand-in for viewpoint_cam.original_image: smoothed uniform noise in [0,1], (3,H,W)."""
    rng = np.random.RandomState(seed + 12345)
    img = rng.rand(3, H + 4, W + 4).astype(np.float32)
    acc = np.zeros((3, H, W), np.float32)
    for dy in range(5):
        for dx in range(5):
            acc += img[:, dy:dy + H, dx:dx + W]
    return np.clip(acc / 25.0, 0.0, 1.0)


# "Teacher" targets for a stationary benchmark workload (round 6): the ground-truth frames are renders of a PERTURBED copy of the
# scene -- centres moved by a fraction of each splat's own extent, colours, SH detail and opacity logits jittered, extents and
# orientations kept -- so that the loss pulls the student towards a scene with the same footprint statistics and the number of tile
# instances R stays where it started.  (Rounds 1-5 used smoothed noise as targets: the optimiser answered by inflating the splats,
# R drifted 3.0 M -> 4.1 M within 35 steps at cfg2 and 8 M -> 63 M at cfg4, and the CPU leg and the GPU headline timed different R.)
TEACHER = dict(xyz=0.2, dc=0.3, rest=0.05, opacity=0.5, seed=777)


def teacher_np(g, seed=None):
    """numpy raw-parameter dict of the teacher of `g` (a make_gaussians()-style dict)."""
    rng = np.random.RandomState(TEACHER["seed"] if seed is None else seed)
    t = dict(g)
    t["xyz"] = (g["xyz"] + TEACHER["xyz"] * np.exp(g["scaling"]) * rng.randn(*g["xyz"].shape)).astype(np.float32)
    t["features_dc"] = (g["features_dc"] + TEACHER["dc"] * rng.randn(*g["features_dc"].shape)).astype(np.float32)
    t["features_rest"] = (g["features_rest"] + TEACHER["rest"] * rng.randn(*g["features_rest"].shape)).astype(np.float32)
    t["opacity"] = (g["opacity"] + TEACHER["opacity"] * rng.randn(*g["opacity"].shape)).astype(np.float32)
    return t


def teacher_torch(model_cls, g, generator):
    """The same recipe on the GPU: a GaussianModel (`model_cls`) holding the perturbed copy of GaussianModel `g`."""
    import torch
    dev = g._xyz.device
    rn = lambda like: torch.randn(like.shape, device=dev, generator=generator)
    t = model_cls(sh_degree=g.max_sh_degree, device=dev)
    with torch.no_grad():
        t.load_raw(g._xyz + TEACHER["xyz"] * torch.exp(g._scaling) * rn(g._xyz), g._features_dc + TEACHER["dc"] * rn(g._features_dc),
                   g._features_rest + TEACHER["rest"] * rn(g._features_rest), g._scaling.clone(), g._rotation.clone(),
                   g._opacity + TEACHER["opacity"] * rn(g._opacity), g._normal.clone())
    t.active_sh_degree = g.active_sh_degree
    return t


# BASELINE.json configs restated as synthetic workloads
CONFIGS = {
    "cfg1": dict(W=400, H=400, P=20_000, white_bg=True, is_blender=True),
    "cfg2": dict(W=800, H=800, P=100_000, white_bg=True, is_blender=True),
    "cfg3": dict(W=800, H=800, P=100_000, white_bg=False, is_blender=True, frames=200),
    "cfg4": dict(W=1080, H=1920, P=300_000, white_bg=True, is_blender=False, off_centre=True),
    "cfg5": dict(W=1024, H=1024, P=500_000, white_bg=True, is_blender=True, extent=1.0),
}


def config_camera(name, frame=0, n_frames=200):
    c = CONFIGS[name]
    W, H = c["W"], c["H"]
    K = None
    if c.get("off_centre"):
        f = 0.5 * W / math.tan(0.6911 / 2)
        K = np.array([[f, 0, 0.47 * W], [0, f, 0.52 * H], [0, 0, 1]], np.float64)
    az = 2 * math.pi * frame / n_frames + 0.3
    return make_camera(W, H, azimuth=az, elevation=0.35, fid=frame / n_frames, K=K)
