"""TEST-ONLY: a CPU `render()` built on the oracle, so that host logic which needs a differentiable rasterizer
(the trainer, data parallelism) can be exercised without a GPU.  Never imported by the product package."""
import math

import numpy as np
import torch

from oracle import oracle as orc


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, opacities, scales, rotations, shs, cam, bg, degree):
        a = [t.detach().numpy().astype(np.float32) for t in (means3D, opacities, scales, rotations, shs)]
        W, H = cam.image_width, cam.image_height
        tanx, tany = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
        vm, pm, cp = (cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy())
        f = orc.forward(bg.numpy(), a[0], None, a[1], a[2], a[3], 1.0, None, vm, pm, tanx, tany, H, W, a[4], degree, cp)
        ctx.pack = (f, a, (vm, pm, cp, tanx, tany), bg.numpy(), degree)
        return torch.tensor(f["color"]), torch.tensor(f["radii"])

    @staticmethod
    def backward(ctx, dL, _):
        f, a, (vm, pm, cp, tanx, tany), bg, degree = ctx.pack
        g = orc.backward(f, bg, a[0], None, a[2], a[3], 1.0, None, vm, pm, tanx, tany, dL.numpy(), a[4], degree, cp)
        t = torch.tensor
        return (t(g["dL_dmeans3D"]), t(g["dL_dopacity"]), t(g["dL_dscales"]), t(g["dL_drotations"]), t(g["dL_dsh"]),
                None, None, None)


def render(viewpoint_camera, pc, pipe, bg_color, d_xyz, d_rotation, d_scaling, is_6dof=False):
    means3D = pc.get_xyz + d_xyz
    img, radii = _OracleRasterize.apply(means3D, pc.get_opacity, pc.get_scaling + d_scaling,
                                        pc.get_rotation + d_rotation, pc.get_features, viewpoint_camera, bg_color,
                                        pc.active_sh_degree)
    return {"render": img, "viewspace_points": None, "visibility_filter": radii > 0, "radii": radii}
