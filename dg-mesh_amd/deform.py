"""Deformation / appearance MLPs with the reference's module API and state_dict layout.

Mirrors R/utils/time_utils.py (R/ = /root/reference/dgmesh/):
  get_embedder / Embedder         :7-55    positional encoding [v, sin(2^k v), cos(2^k v)]_{k<L}
  DeformNetwork                   :58-129  -> (d_xyz, rotation, scaling)
  DeformNetworkNormal             :132-204 -> (d_xyz, rotation, scaling, normal)
  DeformNetworkNormalSep          :207-266 -> normal          (head zero-initialised)
  AppearanceNetwork               :269-323 -> sigmoid(color)
and the thin optimiser wrappers R/scene/deform_model.py:8-138, R/scene/appearance_model.py:8-46.

Sub-modules are created in the same order and with the same names as the reference, so (a) a reference
checkpoint's state_dict loads unchanged (`timenet.{0,2}`, `linear.{0..7}`, `gaussian_warp|branch_w|branch_v`,
`gaussian_rotation`, `gaussian_scaling`, `gaussian_normal`, `color_warp.0`) and (b) default initialisation under
a given torch seed yields bit-identical weights (tests/golden pins this).

The trunk (PE(x) ++ t_emb -> 8 x 256 ReLU with the skip after layer 4 -> heads) is what costs FLOPs
(520 704 MAC per row); `trunk_impl` selects how it runs:
  "torch" : F.linear chain (rocBLAS on the GPU).  Used for parity against the reference module.
  "hip"   : fused fp32-MFMA kernels of libdgmesh_hip.so (dgm_mlp_*), same arithmetic contract.
The time branch (PE(t), `timenet`) is one row in training (t is identical for all rows, R/train.py:158) and
stays in torch either way.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def positional_encoding(v, n_freqs):
    """R/utils/time_utils.py:24-55: cat([v] + [sin(v * 2^k), cos(v * 2^k) for k < n_freqs], -1)."""
    out = [v]
    freqs = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    for f in freqs:
        out.append(torch.sin(v * f))
        out.append(torch.cos(v * f))
    return torch.cat(out, -1)


def skew(w):
    zeros = torch.zeros(w.shape[0], device=w.device)
    return torch.stack([zeros, -w[:, 2], w[:, 1], w[:, 2], zeros, -w[:, 0], -w[:, 1], w[:, 0], zeros], -1).reshape(-1, 3, 3)


def exp_se3(S, theta):
    """R/utils/rigid_utils.py:40-83 (Modern Robotics 3.51 / 3.88); only reached with is_6dof=True."""
    w, v = torch.split(S, 3, dim=-1)
    W = skew(w)
    eye = torch.eye(3, device=W.device).unsqueeze(0).repeat(W.shape[0], 1, 1)
    W_sqr = torch.bmm(W, W)
    R = eye + torch.sin(theta.unsqueeze(-1)) * W + (1.0 - torch.cos(theta.unsqueeze(-1))) * W_sqr
    th = theta.view(-1, 1, 1)
    p = torch.bmm(th * eye + (1.0 - torch.cos(th)) * W + (th - torch.sin(th)) * W_sqr, v.unsqueeze(-1))
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=R.device).repeat(R.shape[0], 1, 1)
    return torch.cat([torch.cat([R, p], dim=-1), bottom], dim=1)


class _TrunkNet(nn.Module):
    """Shared skeleton: time branch + 8-layer trunk with one skip; subclasses add heads."""

    def __init__(self, D=8, W=256, input_ch=3, output_ch=59, multires=10, is_blender=False, is_6dof=False,
                 trunk_impl=None):
        super().__init__()
        self.D, self.W = D, W
        self.output_ch = output_ch
        self.t_multires = 6 if is_blender else 10
        self.multires = multires
        self.skips = [D // 2]
        time_input_ch = 1 + 2 * self.t_multires
        xyz_input_ch = 3 + 3 * 2 * multires
        self.input_ch = xyz_input_ch + time_input_ch
        if is_blender:
            self.time_out = 30
            self.timenet = nn.Sequential(nn.Linear(time_input_ch, 256), nn.ReLU(inplace=True),
                                         nn.Linear(256, self.time_out))
            in0 = xyz_input_ch + self.time_out
        else:
            in0 = self.input_ch
        self.linear = nn.ModuleList([nn.Linear(in0, W)] + [
            nn.Linear(W, W) if i not in self.skips else nn.Linear(W + in0, W) for i in range(D - 1)])
        self.is_blender = is_blender
        self.is_6dof = is_6dof
        self.trunk_impl = trunk_impl or os.environ.get("DGM_MLP_IMPL", "torch")

    def time_embedding(self, t):
        t_emb = positional_encoding(t, self.t_multires)
        if self.is_blender:
            t_emb = self.timenet(t_emb)
        return t_emb

    def _time_rows(self, t):
        """(t_emb, broadcast): t identical on every row (expanded view, R/train.py:158) -> evaluate the time branch
        on ONE row and broadcast it."""
        if t.dim() == 2 and t.shape[0] > 1 and t.stride(0) == 0:
            if self.trunk_impl == "hip" and self.is_blender and t.is_cuda and t.shape[1] == 1:
                from . import mlp_hip
                return mlp_hip.time_row(self, t[:1]), True  # one small kernel instead of ~45 one-element ones
            return self.time_embedding(t[:1]), True
        return self.time_embedding(t), False

    def trunk(self, x, t):
        """PE(x) ++ t_emb -> 8 ReLU layers with the skip (time_utils.py:104-114), PyTorch ops."""
        t_emb, bcast = self._time_rows(t)
        if bcast:
            t_emb = t_emb.expand(x.shape[0], -1)
        x_emb = positional_encoding(x, self.multires)
        h = torch.cat([x_emb, t_emb], dim=-1)
        for i, _ in enumerate(self.linear):
            h = F.relu(self.linear[i](h))
            if i in self.skips:
                h = torch.cat([x_emb, t_emb, h], -1)
        return h

    def head_modules(self):
        """Linear heads applied to the trunk output, in output-column order."""
        raise NotImplementedError

    def heads_out(self, x, t):
        """Concatenated raw head outputs (N, sum of head widths)."""
        heads = self.head_modules()
        if self.trunk_impl == "hip":
            from . import mlp_hip
            t_emb, bcast = self._time_rows(t)
            return mlp_hip.network_forward(self, heads, x, t_emb, bcast)
        h = self.trunk(x, t)
        return torch.cat([m(h) for m in heads], dim=-1)


class DeformNetwork(_TrunkNet):
    def __init__(self, D=8, W=256, input_ch=3, output_ch=59, multires=10, is_blender=False, is_6dof=False,
                 trunk_impl=None):
        super().__init__(D, W, input_ch, output_ch, multires, is_blender, is_6dof, trunk_impl)
        if is_6dof:
            self.branch_w = nn.Linear(W, 3)
            self.branch_v = nn.Linear(W, 3)
        else:
            self.gaussian_warp = nn.Linear(W, 3)
        self.gaussian_rotation = nn.Linear(W, 4)
        self.gaussian_scaling = nn.Linear(W, 3)

    def head_modules(self):
        warp = [self.branch_w, self.branch_v] if self.is_6dof else [self.gaussian_warp]
        return warp + [self.gaussian_rotation, self.gaussian_scaling]

    def _split(self, o):
        if self.is_6dof and o.is_cuda and o.dtype == torch.float32 and self.trunk_impl == "hip":
            from .glue import se3_exp  # one kernel each way (dgm_se3_exp_*) instead of ~25 torch launches
            d_xyz, o = se3_exp(o[:, 0:6]), o[:, 6:]
        elif self.is_6dof:
            w, v, o = o[:, 0:3], o[:, 3:6], o[:, 6:]
            theta = torch.norm(w, dim=-1, keepdim=True)
            w = w / theta + 1e-5
            v = v / theta + 1e-5
            d_xyz = exp_se3(torch.cat([w, v], dim=-1), theta)
        else:
            d_xyz, o = o[:, 0:3], o[:, 3:]
        return d_xyz, o[:, 0:4], o[:, 4:7], o[:, 7:]

    def forward(self, x, t):
        d_xyz, rotation, scaling, _ = self._split(self.heads_out(x, t))
        return d_xyz, rotation, scaling


class DeformNetworkNormal(DeformNetwork):
    def __init__(self, D=8, W=256, input_ch=3, output_ch=59, multires=10, is_blender=False, is_6dof=False,
                 trunk_impl=None):
        super().__init__(D, W, input_ch, output_ch, multires, is_blender, is_6dof, trunk_impl)
        self.gaussian_normal = nn.Linear(W, 3)

    def head_modules(self):
        return super().head_modules() + [self.gaussian_normal]

    def forward(self, x, t):
        d_xyz, rotation, scaling, normal = self._split(self.heads_out(x, t))
        return d_xyz, rotation, scaling, normal


class DeformNetworkNormalSep(_TrunkNet):
    def __init__(self, D=8, W=256, input_ch=3, output_ch=59, multires=10, is_blender=False, is_6dof=False,
                 trunk_impl=None):
        super().__init__(D, W, input_ch, output_ch, multires, is_blender, is_6dof, trunk_impl)
        self.gaussian_normal = nn.Linear(W, 3)
        self.gaussian_normal.weight.data.zero_()
        self.gaussian_normal.bias.data.zero_()

    def head_modules(self):
        return [self.gaussian_normal]

    def forward(self, x, t):
        return self.heads_out(x, t)


class AppearanceNetwork(_TrunkNet):
    def __init__(self, D=8, W=256, input_ch=3, output_ch=59, multires=10, is_blender=False, trunk_impl=None):
        super().__init__(D, W, input_ch, output_ch, multires, is_blender, False, trunk_impl)
        self.color_warp = nn.Sequential(nn.Linear(W, 3), nn.Sigmoid())

    def head_modules(self):
        return [self.color_warp[0]]

    def forward(self, x, t):
        return self.color_warp[1](self.heads_out(x, t))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """R/utils/general_utils.py:42-75 (log-linear decay with optional delayed warm-up)."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return helper


def get_linear_noise_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """R/utils/general_utils.py:78-111: like get_expon_lr_func but LINEAR interpolation between lr_init and lr_final
    (the annealing factor of the time-input noise, R/train.py:119-121)."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * (lr_init * (1 - t) + lr_final * t)
    return helper


class _ModelWrapper:
    """DeformModel* / AppearanceModel: .step(), Adam(eps=1e-15) setup, lr schedule, save/load
    (R/scene/deform_model.py:8-138, R/scene/appearance_model.py:8-46)."""
    net_attr = "deform"

    def __init__(self, net, lr_scale, model_name, final_lr_scale=1.0):
        setattr(self, self.net_attr, net)
        if getattr(net, "trunk_impl", None) == "hip":  # head weights back to back: read in place by the fused kernels
            from . import mlp_hip
            mlp_hip.pack_heads(net.head_modules())
        self.optimizer = None
        self.spatial_lr_scale = lr_scale
        self.final_lr_scale = final_lr_scale
        self.model_name = model_name

    @property
    def net(self):
        return getattr(self, self.net_attr)

    def step(self, xyz, time_emb):
        return self.net(xyz, time_emb)

    def step_raw(self, xyz, time_emb):
        """The network's raw (N, n_out) head output [d_xyz | d_rotation | d_scaling | ...] without slicing it (the fused
        glue kernels of glue.py take it whole); None when the network has no such layout (is_6dof, other classes)."""
        net = self.net
        if isinstance(net, DeformNetwork) and not net.is_6dof:
            return net.heads_out(xyz, time_emb)
        return None

    def train_setting(self, training_args):
        lr0 = training_args.position_lr_init * self.spatial_lr_scale
        self.optimizer = torch.optim.Adam([{"params": list(self.net.parameters()), "lr": lr0, "name": self.model_name}],
                                          lr=0.0, eps=1e-15)
        self.scheduler = get_expon_lr_func(lr_init=lr0, lr_final=training_args.position_lr_final * self.final_lr_scale,
                                           lr_delay_mult=training_args.position_lr_delay_mult,
                                           max_steps=training_args.deform_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == self.model_name:
                lr = self.scheduler(iteration)
                group["lr"] = lr
                return lr

    def save_weights(self, model_path, iteration):
        out = os.path.join(model_path, f"{self.model_name}/iteration_{iteration}")
        os.makedirs(out, exist_ok=True)
        torch.save(self.net.state_dict(), os.path.join(out, f"{self.model_name}.pth"))

    def load_weights(self, model_path, iteration=-1):
        root = os.path.join(model_path, self.model_name)
        if iteration == -1:  # searchForMaxIteration, R/utils/system_utils.py:29-31
            iteration = max(int(f.split("_")[-1]) for f in os.listdir(root))
        self.net.load_state_dict(torch.load(os.path.join(root, f"iteration_{iteration}/{self.model_name}.pth")))


class DeformModel(_ModelWrapper):
    def __init__(self, is_blender=False, is_6dof=False, device="cuda", trunk_impl=None):
        super().__init__(DeformNetwork(is_blender=is_blender, is_6dof=is_6dof, trunk_impl=trunk_impl).to(device), 5, "deform")


class DeformModelNormal(_ModelWrapper):
    def __init__(self, is_blender=False, is_6dof=False, model_name="deform", device="cuda", trunk_impl=None):
        super().__init__(DeformNetworkNormal(is_blender=is_blender, is_6dof=is_6dof, trunk_impl=trunk_impl).to(device), 5,
                         model_name)


class DeformModelNormalSep(_ModelWrapper):
    def __init__(self, is_blender=False, is_6dof=False, model_name="deform_normal", device="cuda", trunk_impl=None):
        super().__init__(DeformNetworkNormalSep(is_blender=is_blender, is_6dof=is_6dof, trunk_impl=trunk_impl).to(device),
                         10.0, model_name, final_lr_scale=10.0)


class AppearanceModel(_ModelWrapper):
    net_attr = "appearance_net"

    def __init__(self, is_blender=False, is_6dof=False, device="cuda", trunk_impl=None):
        super().__init__(AppearanceNetwork(is_blender=is_blender, trunk_impl=trunk_impl).to(device), 1.0, "appearance")

    def train_setting(self, training_args):
        lr0 = training_args.apperance_lr_init
        self.optimizer = torch.optim.Adam([{"params": list(self.net.parameters()), "lr": lr0, "name": "appearance"}],
                                          lr=0.0, eps=1e-15)
        self.scheduler = get_expon_lr_func(lr_init=lr0, lr_final=training_args.apperance_lr_final,
                                           lr_delay_mult=training_args.apperance_lr_delay_mult,
                                           max_steps=training_args.apperance_lr_max_steps)
