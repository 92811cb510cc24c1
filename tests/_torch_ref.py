"""Dense differentiable PyTorch (float64) restatement of the rasterizer forward, used to cross-check
the oracle's hand-written gradients with torch.autograd.

It takes the *discrete structure* (per-tile sorted lists, which are integer outputs already checked
bit-exactly elsewhere) from the oracle and re-derives every float with autograd-tracked math:
  preprocess  DGR/cuda_rasterizer/forward.cu:74-152, 156-256 ; SH forward.cu:20-71
  blending    DGR/cuda_rasterizer/forward.cu:325-373
Hard, non-differentiated decisions of the reference are reproduced as detached masks:
alpha<1/255, power>0, T'<1e-4 termination, min(.99, alpha) clamp passes gradient straight through.
Scenes used with this module must not trigger the +-1.3 tan(fov) clamp (backward.cu:175-176 is not
the autograd derivative there; that branch has its own known-answer test).
"""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_basis(deg, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    B = [torch.full_like(x, C0)]
    if deg > 0:
        B += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        B += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        B += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(B, 1)  # (P, n)


def preprocess(means3D, scales, rotations, opacities, shs, vm, pm, campos, W, H, tanfovx, tanfovy, deg,
               scale_modifier=1.0):
    """Returns pix (P,2) [pixel coords], ndc (P,2) leaf-like intermediate, conic (P,3), rgb (P,3), depth."""
    P = means3D.shape[0]
    one = torch.ones(P, 1, dtype=means3D.dtype)
    hom = torch.cat([means3D, one], 1)
    p_view = hom @ vm  # vm is W2C^T (row-vector convention), R/scene/cameras.py:60-71
    p_hom = hom @ pm
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    ndc.retain_grad()
    size = torch.tensor([W, H], dtype=means3D.dtype)
    pix = ((ndc + 1.0) * size - 1.0) * 0.5
    # cov3D (un-normalised quaternion)
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    Rq = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    S = torch.diag_embed(scales * scale_modifier)
    Sigma = Rq @ S @ S @ Rq.transpose(1, 2)
    # EWA projection
    t = p_view[:, :3]
    tz = t[:, 2]
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * t[:, 0] / (tz * tz), zero, fy / tz, -fy * t[:, 1] / (tz * tz)], 1).reshape(P, 2, 3)
    Rwc = vm[:3, :3].T  # rotation part of W2C
    A = J @ Rwc
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], 1)
    d = means3D - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    n = (deg + 1) ** 2
    B = sh_basis(deg, d)
    raw = (B[:, :, None] * shs[:, :n, :]).sum(1) + 0.5
    rgb = torch.clamp_min(raw, 0.0)
    return pix, ndc, conic, rgb, tz


def render(pix, conic, opac, rgb, bg, ranges, point_list, W, H):
    """Front-to-back alpha blending per pixel over the tile's sorted list; returns (3,H,W) and n_contrib."""
    out = torch.zeros(3, H, W, dtype=pix.dtype)
    ncon = torch.zeros(H, W, dtype=torch.int64)
    gx = (W + 15) // 16
    rows = []
    for tile in range(ranges.shape[0]):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        tx, ty = tile % gx, tile // gx
        xs = torch.arange(tx * 16, min(tx * 16 + 16, W))
        ys = torch.arange(ty * 16, min(ty * 16 + 16, H))
        if len(xs) == 0 or len(ys) == 0:
            continue
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        px = px.reshape(-1).to(pix.dtype)
        py = py.reshape(-1).to(pix.dtype)
        npx = px.shape[0]
        T = torch.ones(npx, dtype=pix.dtype)
        C = torch.zeros(npx, 3, dtype=pix.dtype)
        done = torch.zeros(npx, dtype=torch.bool)
        last = torch.zeros(npx, dtype=torch.int64)
        ids = torch.as_tensor(point_list[r0:r1].astype("int64"))
        for k, g in enumerate(ids.tolist()):
            dx = pix[g, 0] - px
            dy = pix[g, 1] - py
            power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
            a_raw = opac[g] * torch.exp(power)
            alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()  # straight-through clamp
            ok = (~done) & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
            test_T = T * (1 - alpha)
            term = ok & (test_T.detach() < 1e-4)
            done = done | term
            ok = ok & ~term
            w = torch.where(ok, alpha * T, torch.zeros_like(T))
            C = C + w[:, None] * rgb[g][None, :]
            T = torch.where(ok, test_T, T)
            last = torch.where(ok, torch.full_like(last, k + 1), last)
        col = C + T[:, None] * bg[None, :]
        yy = py.long()
        xx = px.long()
        rows.append((yy, xx, col, last))
    for yy, xx, col, last in rows:
        out[:, yy, xx] = col.T
        ncon[yy, xx] = last
    return out, ncon
