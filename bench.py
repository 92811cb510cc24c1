"""bench.py -- DG-Mesh train-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic frame per rank: deformation MLP -> differentiable
Gaussian rasterizer (forward) -> deform_back MLP + cycle loss -> 0.8 L1 + 0.2 (1-SSIM) -> backward through the
rasterizer and both MLPs -> (N>1: one flat-bucket RCCL all-reduce) -> Adam updates.  Workload = BASELINE config
"D-NeRF jumpingjacks, 800x800, ~100k Gaussians, single MI355X, deformation MLP on" (cfg2), synthetic data.
Frames shard across ranks (weak scaling: every rank renders its own 800x800 frame each step).

Rank 0 prints ONE JSON line.  `value` = frames trained per second over all ranks, inputs resident in HBM.
`roofline` describes the DOMINANT hand-written kernel of the step -- chosen at run time as the instrumented kernel
with the largest (average launch time x launches per step); today the paired backward launch of a 256 -> 256 trunk layer
(mlp_bwd_pair_kernel: backward data + weight gradient on plane-format activations) -- timed with hipEvents recorded on
the launch stream inside the timed region (deferred read-out, no extra sync) and priced against BOTH roofs it can hit:
HBM (algorithmic bytes / 8 TB/s) and the f16 matrix pipe it issues on (3 MFMAs per fp32 product: 3 x flops /
2.5 PFLOP/s); `bound` is the nearer one.
`roofline_render_bwd` is the rasterizer backward, the kernel group BASELINE.json's north_star grades against HBM.
`kernels` lists the same two fractions for every instrumented kernel.  When K < 200 a second, 200-step steady-state
region is timed and reported as `steady_state` (SURVEY.md section 8d asks for >= 200 iterations).
`--gpus N` without a launcher environment (no WORLD_SIZE) re-executes this script under torch.distributed.run with N ranks on
127.0.0.1 (one per GPU, backend nccl = RCCL); it refuses to run when fewer than N devices are visible.  The JSON then also
carries the RCCL world size observed and the all-reduce of the step's gradient buckets timed on its own.
`mlp_f32_mode` = the same workload with the MLP GEMMs on the native fp32 MFMA instruction (dgm_mlp_set_gemm(1)), a short extra
region at N = 1.  `roofline_render_bwd_trained` = the rasterizer backward on SURVEY.md section 8(d)'s trained-like scene (the
distribution on which north_star's HBM target is approachable), measured here with the library's stage timers; `frac_valu` of
the blend kernels comes from the committed PMC pass (VALU lane operations / 78.6 T lane-op/s).
`cpu_baseline` = the same step on the host cores: the reference's OWN network and loss modules (utils/time_utils.py,
utils/loss_utils.py, byte-compiled by oracle/build_ref.sh; `kind_detail: "reference modules + oracle rasterizer"`) around the
oracle rasterizer (the reference has no CPU rasterizer, so `kind` is "port": the part that dominates the CPU step is the oracle's
restatement); this repo's torch restatement of the modules only when the byte-compiled files are absent or do not load.  Rank 0 at N=1 only, on a bounded sample; its scene is the
freshly initialised one (frame 0), the GPU headline's has taken ~35 Adam steps -- the two R values differ and `sample` says so.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_FP32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 cycles/SIMD)
MFMA_16BIT_PEAK_TF = 2500.0  # dense f16 / bf16 MFMA
STEADY_STEPS = int(os.environ.get("DGM_BENCH_STEADY_STEPS", "200"))
WORKLOAD = "cfg2"
TARGETS = "teacher"  # --targets: "teacher" (renders of a perturbed copy of the scene: stationary R) | "noise" (rounds 1-5)


def build_scene(dev, rank, world, mlp_impl, n_frames=200, n_gt=4, seed=0, phase="gs", dpsr_res=288, n_verts=60000, densify=False):
    syn = importlib.import_module("dg-mesh_amd.synthetic")
    S = importlib.import_module("dg-mesh_amd.scene")
    D = importlib.import_module("dg-mesh_amd.deform")
    T = importlib.import_module("dg-mesh_amd.trainer")
    c = syn.CONFIGS[WORKLOAD]
    P, W, H = c["P"], c["W"], c["H"]
    rng = np.random.RandomState(seed)
    xyz = ((rng.rand(P, 3) * 2 - 1) * 1.3).astype(np.float32)
    rgb = rng.rand(P, 3).astype(np.float32)
    g = S.GaussianModel(sh_degree=3, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    g.create_from_pcd(xyz, rgb, generator=gen)          # exercises simple-knn (distCUDA2)
    with torch.no_grad():
        g._features_rest.add_(0.05 * torch.randn(g._features_rest.shape, device=dev, generator=gen))
    g.active_sh_degree = 3
    cams = [S.TorchCamera(syn.config_camera(WORKLOAD, frame=f, n_frames=n_frames), dev) for f in range(n_frames)]
    bg = torch.tensor([1.0, 1.0, 1.0] if c["white_bg"] else [0.0, 0.0, 0.0], device=dev)
    if TARGETS == "teacher":
        # every frame's target = that camera's render of a perturbed copy of the scene (synthetic.TEACHER): the loss has a
        # reachable optimum with the student's own footprint statistics, so R is stationary over the run
        tgen = torch.Generator(device=dev)
        tgen.manual_seed(syn.TEACHER["seed"] + seed)
        teacher = syn.teacher_torch(S.GaussianModel, g, tgen)
        with torch.no_grad():
            for cam in cams:
                cam.original_image = S.render(cam, teacher, S.PipelineParams(), bg, 0.0, 0.0, 0.0)["render"].detach().clamp(0.0, 1.0).clone()
        del teacher
    else:  # "noise": rounds 1-5's smoothed-noise targets (the optimiser inflates the splats under them; kept for comparison)
        gts = [torch.tensor(syn.gt_image(W, H, seed=i), device=dev) for i in range(n_gt)]
        for f, cam in enumerate(cams):
            cam.original_image = gts[f % n_gt].clamp(0.0, 1.0)
    torch.manual_seed(seed)
    deform = D.DeformModelNormal(is_blender=c["is_blender"], model_name="deform", device=dev, trunk_impl=mlp_impl)
    deform_back = D.DeformModelNormal(is_blender=c["is_blender"], model_name="deform_back", device=dev, trunk_impl=mlp_impl)
    # A freshly initialised head emits O(0.1) deltas, which would triple every Gaussian's extent (R 3M -> 7M) and
    # measure a scene no trained model looks like.  Scale the output heads so the deformation field is small, as
    # after convergence; the trunk (where the FLOPs are) is untouched.
    with torch.no_grad():
        for m in (deform.net, deform_back.net):
            for head in (m.gaussian_warp, m.gaussian_rotation, m.gaussian_scaling, m.gaussian_normal):
                head.weight.mul_(0.01)
                head.bias.mul_(0.01)
    mesh = None
    if phase == "mesh":
        # mesh co-training phase (R/train.py:165-176, 243-285): the two normal networks on the P Gaussians, DPSR on the deformed
        # points (grid dpsr_res^3), deform_back + appearance on V vertices.  DiffMC / nvdiffrast are third-party and not
        # rebuilt: phi is probed at V fixed points that stand in for the mesh vertices (trainer.py).  The appearance network
        # differentiates w.r.t. its input (vertex positions moved by deform_back): dgm_mlp_backward_dx of the fused trunk.
        DP = importlib.import_module("dg-mesh_amd.dpsr")
        dn = D.DeformModelNormalSep(is_blender=c["is_blender"], model_name="deform_normal", device=dev, trunk_impl=mlp_impl)
        dbn = D.DeformModelNormalSep(is_blender=c["is_blender"], model_name="deform_back_normal", device=dev, trunk_impl=mlp_impl)
        app = D.AppearanceModel(is_blender=c["is_blender"], device=dev, trunk_impl=mlp_impl)
        with torch.no_grad():
            for m in (dn, dbn):  # zero-initialised head in the reference (time_utils.py:248-249): small but non-zero here
                torch.nn.init.normal_(m.net.gaussian_normal.weight, std=1e-3)
        extent = c.get("extent", 1.3)
        mesh = T.MeshPhase(dn, dbn, app, dpsr=DP.DPSR(res=(dpsr_res,) * 3, sig=2.0), n_verts=n_verts, scale=1.1 * extent, seed=seed,
                           device=dev)
    tr = T.Trainer(g, deform, deform_back, cams, background=bg, is_blender=c["is_blender"], rank=rank, world=world,
                   seed=seed, mesh=mesh, densify=densify, cameras_extent=c.get("extent", 1.3))
    return tr, (P, W, H)


def reference_host_modules():
    """The reference's own `utils/time_utils.py` (DeformNetworkNormal) and `utils/loss_utils.py` (l1_loss, ssim), byte-compiled
    by oracle/build_ref.sh where /root/reference exists (binaries only; they travel to the GPU box like the reference kernels).
    None when they were not built."""
    import importlib.machinery
    import importlib.util
    import types
    root = os.path.join(ROOT, "oracle", "_ref", "pyref")
    if not all(os.path.exists(os.path.join(root, f)) for f in ("time_utils.pyc", "loss_utils.pyc", "rigid_utils.pyc")):
        return None
    saved = {k: sys.modules.get(k) for k in ("utils", "utils.rigid_utils")}

    def load(name, fname):
        loader = importlib.machinery.SourcelessFileLoader(name, os.path.join(root, fname))
        mod = importlib.util.module_from_spec(importlib.util.spec_from_loader(name, loader))
        sys.modules[name] = mod
        loader.exec_module(mod)
        return mod
    try:
        sys.modules["utils"] = types.ModuleType("utils")
        load("utils.rigid_utils", "rigid_utils.pyc")
        return load("ref_time_utils", "time_utils.pyc"), load("ref_loss_utils", "loss_utils.pyc")
    except Exception:  # present but unloadable (another Python's magic number, a missing import): the port is timed instead
        return None
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def cpu_baseline(P, W, H, max_threads=32, headline_R=None, frame=0, scene=None):
    """The same train step on the host: the reference's own network and loss modules on PyTorch-CPU (this repo's torch
    restatement of them only if the byte-compiled modules are absent) around the oracle rasterizer (C, OpenMP; the reference
    has no CPU rasterizer).  A few steps of the full cfg2 workload are the bounded sample."""
    syn = importlib.import_module("dg-mesh_amd.synthetic")
    D = importlib.import_module("dg-mesh_amd.deform")
    S = importlib.import_module("dg-mesh_amd.scene")
    from oracle import oracle as orc

    cores = min(os.cpu_count() or 1, max_threads)
    orc.set_threads(cores)
    torch.set_num_threads(cores)
    if scene is not None:  # the GPU run's own Gaussians (as they stand after its timed region) and its target for that view
        g = scene["g"]
    else:
        xyz0 = ((np.random.RandomState(0).rand(P, 3) * 2 - 1) * 1.3).astype(np.float32)  # the points build_scene() draws
        g = syn.make_gaussians(P, seed=0, dist2=orc.knn(xyz0))  # same initialisation as create_from_pcd (simple-knn scales)
    cam = syn.config_camera(WORKLOAD, frame=frame)  # the view of the headline's last timed step (R differs by a few % between views)
    tanx, tany = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    white = bool(syn.CONFIGS[WORKLOAD]["white_bg"])
    bg = np.ones(3, np.float32) if white else np.zeros(3, np.float32)
    orc.lib()
    if scene is not None:
        gt = scene["gt"]
    elif TARGETS == "teacher":  # the same recipe as build_scene(): the view's target = the oracle's render of the perturbed copy
        at = syn.activate(syn.teacher_np(g))
        ft = orc.forward(bg, at["means3D"], None, at["opacities"], at["scales"], at["rotations"], 1.0, None, cam.world_view_transform,
                         cam.full_proj_transform, tanx, tany, H, W, at["shs"], 3, cam.camera_center)
        gt = torch.tensor(np.clip(ft["color"], 0.0, 1.0))
    else:
        gt = torch.tensor(syn.gt_image(W, H, 0))
    torch.manual_seed(0)
    ref = reference_host_modules()
    is_blender = bool(syn.CONFIGS[WORKLOAD]["is_blender"])
    if ref is not None:
        nets = [ref[0].DeformNetworkNormal(is_blender=is_blender) for _ in range(2)]
        l1_loss, ssim = ref[1].l1_loss, ref[1].ssim
    else:
        nets = [D.DeformNetworkNormal(is_blender=is_blender, trunk_impl="torch") for _ in range(2)]
        l1_loss, ssim = S.l1_loss, S.ssim
    with torch.no_grad():  # same small-deformation heads as the GPU workload (build_scene), so R is comparable
        for m in nets:
            for head in (m.gaussian_warp, m.gaussian_rotation, m.gaussian_scaling, m.gaussian_normal):
                head.weight.mul_(0.01)
                head.bias.mul_(0.01)
    params = [p for n in nets for p in n.parameters()]
    opt = torch.optim.Adam(params, lr=1e-4, eps=1e-15)
    xyz = torch.tensor(g["xyz"])
    t_in = torch.tensor([[0.3]]).expand(P, -1)

    def one_step():
        for p in params:
            p.grad = None
        d_xyz, d_rot, d_scale, _ = nets[0](xyz, t_in)
        a = syn.activate(g, d_xyz.detach().numpy(), d_rot.detach().numpy(), d_scale.detach().numpy())
        f = orc.forward(bg, a["means3D"], None, a["opacities"], a["scales"], a["rotations"], 1.0, None,
                        cam.world_view_transform, cam.full_proj_transform, tanx, tany, H, W, a["shs"], 3, cam.camera_center)
        img = torch.tensor(f["color"], requires_grad=True)
        loss_img = 0.8 * l1_loss(img, gt) + 0.2 * (1.0 - ssim(img, gt))
        loss_img.backward()
        gr = orc.backward(f, bg, a["means3D"], None, a["scales"], a["rotations"], 1.0, None, cam.world_view_transform,
                          cam.full_proj_transform, tanx, tany, img.grad.numpy(), a["shs"], 3, cam.camera_center)
        back = nets[1]((xyz + d_xyz).detach(), t_in)
        cyc = (l1_loss(-back[0], d_xyz) + l1_loss(-back[1], d_rot) + l1_loss(-back[2], d_scale)) / 3.0
        surrogate = (d_xyz * torch.tensor(gr["dL_dmeans3D"])).sum() + (d_rot * torch.tensor(gr["dL_drotations"])).sum() + \
            (d_scale * torch.tensor(gr["dL_dscales"])).sum()
        (cyc + surrogate).backward()
        opt.step()
        return f

    t0 = time.time()
    f = one_step()
    n_steps = 1
    while time.time() - t0 < 12.0 and n_steps < 6:  # bounded sample: about 10-20 s of CPU work
        f = one_step()
        n_steps += 1
    dt = (time.time() - t0) / n_steps
    detail = "reference modules + oracle rasterizer" if ref is not None else "torch restatement of the modules + oracle rasterizer"
    what = ("the reference's DeformNetworkNormal x 2 fwd+bwd and its l1_loss / ssim (utils/time_utils.py, utils/loss_utils.py, "
            "byte-compiled)" if ref is not None else "2 deformation MLPs fwd+bwd + L1/SSIM (this repo's torch restatement)")
    return {"value": 1.0 / dt, "unit": "it/s", "cores": cores, "kind": "port", "kind_detail": detail,
            "sample": f"{n_steps} full {WORKLOAD} train steps ({W}x{H}, P={P}, R={f['num_rendered']}): oracle rasterizer "
                      f"fwd+bwd (C/OpenMP; the reference has no CPU rasterizer) + {what} + torch.optim.Adam on PyTorch-CPU, "
                      f"{dt:.1f} s each.  " + ("The GPU headline's own Gaussians (copied to the host after its timed region) and its "
                                               f"target image for the view of its last timed step (frame {frame}); fresh networks; "
                                               if scene is not None else
                                               "Same workload recipe as the GPU headline (teacher-rendered targets, the view of its last "
                                               f"timed step -- frame {frame} --, fresh networks); ") +
                      f"R here {f['num_rendered']}" + (f", the headline's last frame R={headline_R} "
                                                     f"({100.0 * (f['num_rendered'] / headline_R - 1.0):+.1f} %)" if headline_R else ""),
            "num_rendered": int(f["num_rendered"])}


def trained_like_render_bwd(dev, iters=12):
    """Rasterizer forward + backward on SURVEY.md section 8(d)'s trained-like scene (shell of radius 0.8, sigma ~ 0.01,
    opacity U[0.5, 0.99]: early termination as in a converged scene) at the workload's resolution, stage timers of the library
    (hipEvents on the launch stream).  Returns the render-backward roofline block north_star asks for."""
    syn = importlib.import_module("dg-mesh_amd.synthetic")
    L = importlib.import_module("dg-mesh_amd._lib")
    RZ = importlib.import_module("dg-mesh_amd.rasterizer")
    c = syn.CONFIGS[WORKLOAD]
    P, W, H = c["P"], c["W"], c["H"]
    g = syn.make_gaussians(P, seed=0, kind="trained", dist2=np.full(P, 1e-4, np.float32))
    a = syn.activate(g)
    cam = syn.config_camera(WORKLOAD, frame=3)
    T = lambda x: torch.tensor(x, device=dev)
    bg = T(np.ones(3, np.float32))
    means3D, opac, scales, rots, sh = T(a["means3D"]), T(a["opacities"]), T(a["scales"]), T(a["rotations"]), T(a["shs"])
    vm, pm, campos = T(cam.world_view_transform), T(cam.full_proj_transform), T(cam.camera_center)
    tanx, tany = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    e = torch.empty(0, device=dev)
    dL = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    L.lib().dgm_set_profiling(1)
    acc = {}
    n = 0
    for it in range(iters + 3):
        n, color, radii, geom, binning, img = RZ._C.rasterize_gaussians(bg, means3D, e, opac, scales, rots, 1.0, e, vm, pm, tanx, tany,
                                                                       H, W, sh, 3, campos, False, False)
        RZ._C.rasterize_gaussians_backward(bg, means3D, radii, e, scales, rots, 1.0, e, vm, pm, tanx, tany, dL, sh, 3, campos,
                                           geom, n, binning, img, False)
        torch.cuda.synchronize()
        if it >= 3:
            for k, v in L.stage_ms().items():
                acc.setdefault(k, []).append(v)
    L.lib().dgm_set_profiling(0)
    med = {k: float(np.median(v)) for k, v in acc.items()}
    R = int(n)
    rb_bytes = 40.0 * R + 20.0 * W * H + 36.0 * P
    grp_bytes = 40.0 * R + 20.0 * W * H + 595.0 * P          # + preprocess backward: the quantity north_star grades
    rb, pb = med.get("render_bwd", 0.0), med.get("preprocess_bwd", 0.0)
    ach = rb_bytes / (rb * 1e-3) / 1e9 if rb > 0 else 0.0
    ach_grp = grp_bytes / ((rb + pb) * 1e-3) / 1e9 if rb + pb > 0 else 0.0
    return {"scene": "trained-like (SURVEY.md 8d): shell r=0.8, sigma~0.01, opacity U[0.5,0.99]", "num_rendered": R,
            "kernel": "render_bwd4_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes": rb_bytes, "avg_ms": rb,
            "group_with_preprocess_bwd": {"algorithmic_bytes": grp_bytes, "avg_ms": rb + pb, "achieved": ach_grp,
                                          "frac": ach_grp / HBM_PEAK_GBS},
            "render_fwd_ms": med.get("render_fwd", 0.0), "render_fwd_frac_hbm":
                ((40.0 * R + 20.0 * W * H) / (med["render_fwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if med.get("render_fwd") else None}


def frac_valu_from_profiles(scene):
    """VALU lane operations per launch of the blend kernels from the newest committed PMC pass OF THE NAMED SCENE
    (profiles/r0*_pmc_sq_<scene>*.json, scene = "bench" | "trained"; offline data, named in `source`) -- the caller divides by the
    kernel time it measured on the same scene.  `valu_busy_of_chip` is the pass's own busy fraction (VALU quad-cycles x 4 over wall
    cycles x 1024 SIMDs), the cross-check for the `frac_valu` derived here."""
    import glob
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r0*pmc_sq_{scene}*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            for short in ("render_bwd4_kernel", "render_fwd_kernel"):
                if short in k and short not in out and "SQ_ACTIVE_INST_VALU" in v and "wall_cycles" in v:
                    # SQ_ACTIVE_INST_VALU: quad-cycles of VALU execution summed over waves; x4 cycles x 32 lanes per cycle
                    lane_ops = 4.0 * v["SQ_ACTIVE_INST_VALU"] * 32.0
                    out[short] = {"valu_lane_ops_per_launch": lane_ops, "valu_busy_of_chip": v.get("valu_busy_of_chip"),
                                  "scene": scene, "source": "profiles/" + os.path.basename(f)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mlp", default=os.environ.get("DGM_MLP_IMPL", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default=os.environ.get("DGM_BENCH_WORKLOAD", "cfg2"),
                    help="BASELINE.json config the synthetic scene follows; the metric is quoted on cfg2 (default), the others are "
                         "informational")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-mode and trained-like extra regions")
    ap.add_argument("--targets", default=os.environ.get("DGM_BENCH_TARGETS", "teacher"), choices=["teacher", "noise"],
                    help="ground-truth frames: renders of a perturbed copy of the scene (default; R stays put) or rounds 1-5's smoothed noise")
    ap.add_argument("--phase", default="gs", choices=["gs", "mesh"],
                    help="gs: dynamic-Gaussian phase (deform + deform_back; the metric's configuration).  mesh: the mesh co-training "
                         "phase -- four networks on P, DPSR chain, deform_back + appearance on V (BASELINE config 5's 'DPSR mesh step')")
    ap.add_argument("--dpsr-res", type=int, default=288)
    ap.add_argument("--verts", type=int, default=60000)
    args = ap.parse_args()
    global WORKLOAD, TARGETS
    WORKLOAD = args.workload
    TARGETS = args.targets

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one.  N ranks on this node, one per GPU, rendezvous on 127.0.0.1.
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and os.environ.get("DGM_BENCH_SHARE_GPU") != "1":
            sys.exit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible -- refusing to time fewer ranks than asked for")
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and "WORLD_SIZE" in os.environ and args.gpus > 1:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    # test aid for one-GPU boxes: DGM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and exchanges over gloo, which
    # exercises the whole multi-rank code path (the measured configuration is one rank per GPU over RCCL)
    share_gpu = os.environ.get("DGM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    L = importlib.import_module("dg-mesh_amd._lib")
    mlp_impl = args.mlp
    if mlp_impl == "auto":
        mlp_impl = "hip" if hasattr(L.lib(), "dgm_mlp_forward") else "torch"

    tr, (P, W, H) = build_scene(dev, rank, world, mlp_impl, phase=args.phase, dpsr_res=args.dpsr_res, n_verts=args.verts)
    it0 = tr.opt.warm_up + 2000  # "deformation MLP on" phase (warm_up <= it < dpsr_iter)
    if args.phase == "mesh":     # every network on, positions unfrozen (it >= dpsr_iter + max(normal_warm_up, 2000))
        it0 = tr.opt.dpsr_iter + importlib.import_module("dg-mesh_amd.trainer").normal_deform_delay(tr.opt) + 1000

    # allocator / code-object / clock priming (untimed, before the W warm-up steps): three steps, then the set-up's 267 k objects are
    # frozen out of the cyclic collector (a full pass costs ~80 ms here) -- the collection that precedes the freeze returns blocks to
    # the caching allocator and the next five steps run 5-25 % slow while it re-forms its pools (tools/step_times.py), so the freeze
    # comes BEFORE the rest of the priming, not between priming and warm-up as in rounds 4-5 -- then 30 more steps (0.1 s)
    for i in range(3):
        tr.step(it0)
    tr.freeze_gc()
    for i in range(30):
        tr.step(it0)
    for i in range(args.warmup):
        tr.step(it0 + i)
    RZ = importlib.import_module("dg-mesh_amd.rasterizer")

    def timed(n_steps, first_it):
        """Barrier + synchronize on both sides, max over ranks; stage timers collected over exactly this region."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        L.lib().dgm_set_profiling_sampling(8)  # bracket every 8th launch of a stage: ~12 event pairs per step instead of ~90
        L.lib().dgm_set_profiling(2)
        torch.cuda.synchronize()
        RZ.FORWARD_CALL_SECONDS = 0.0
        RZ.SETTLE_WAIT_SECONDS = 0.0
        t0 = time.perf_counter()
        pk = None
        for i in range(n_steps):
            pk = None  # (let go of the previous step's render package BEFORE the next step allocates its own: holding it across
            #            the call kept one extra set of frame tensors alive and cost the region one device allocation)
            _, pk = tr.step(first_it + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = L.collect_stage_ms()
        L.lib().dgm_set_profiling(0)
        blocked = RZ.FORWARD_CALL_SECONDS
        timed.settle_wait = RZ.SETTLE_WAIT_SECONDS
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, st, blocked, pk

    def exchange_ms(events):
        """Mean device time of the gradient exchange section of a step (pack, all-reduce(s), wait) on the compute stream."""
        return sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)

    if world > 1:
        tr.exchange_events = []
    dev_allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    redos0 = RZ.OVERFLOW_REDOS + RZ.UNIT_REDOS
    elapsed, stages, blocked_s, pkg = timed(args.steps, it0 + args.warmup)
    settle_s, redos = timed.settle_wait, RZ.OVERFLOW_REDOS + RZ.UNIT_REDOS - redos0
    # hipMalloc calls of torch's caching allocator inside the timed region: 0 at steady state; a scene whose R keeps setting new
    # maxima (cfg4 under the random targets) pays one multi-GB allocation per new size of the binning buffer
    dev_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - dev_allocs0
    dp = None
    if world > 1:
        # Self-verification of the data-parallel run: every rank must hold bit-identical parameters and Adam moments after the
        # timed region (64-bit hash per tensor, all-gathered).  Then the same region with the Gaussian bucket's all-reduce
        # launched early from an autograd hook (Trainer(overlap=True): RCCL's queue runs beside the MLP backward passes).
        dp = {"overlap": False, "replicas_identical": bool(tr.replicas_identical()),
              "exchange_ms_per_step": round(exchange_ms(tr.exchange_events), 4)}
        tr.overlap = True
        tr._bind_parameters()
        for i in range(5):
            tr.step(it0 + i)
        tr.exchange_events = []
        n_ov = max(args.steps, 20)
        o_dt, _, _, _ = timed(n_ov, it0 + args.warmup)
        dp["overlap_on"] = {"value": n_ov * world / o_dt, "unit": "it/s", "ms_per_step": 1e3 * o_dt / n_ov, "steps": n_ov,
                            "replicas_identical": bool(tr.replicas_identical()),
                            "exchange_ms_per_step": round(exchange_ms(tr.exchange_events), 4),
                            "note": "Gaussian bucket all-reduced on RCCL's queue under the MLP backward passes; opt-in"}
        tr.exchange_events = None
        tr.overlap = False
        tr._bind_parameters()
    steady = None
    if args.steps < STEADY_STEPS:
        s_dt, _, _, _ = timed(STEADY_STEPS, it0 + args.warmup + args.steps)
        steady = {"steps": STEADY_STEPS, "value": STEADY_STEPS * world / s_dt, "ms_per_step": 1e3 * s_dt / STEADY_STEPS}

    n_inst_timed = int(RZ.LAST_NUM_RENDERED)  # tile instances R of the timed workload's last frame (the extras below render other scenes)
    last_frame = int(getattr(tr, "last_frame", 0))
    cpu_scene = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # the CPU leg times THIS scene: the Gaussians as the timed region left them
        host = lambda t: t.detach().cpu().numpy().copy()
        gm = tr.g
        cpu_scene = {"g": dict(xyz=host(gm._xyz), features_dc=host(gm._features_dc), features_rest=host(gm._features_rest),
                               scaling=host(gm._scaling), rotation=host(gm._rotation), opacity=host(gm._opacity)),
                     "gt": tr.cameras[last_frame].original_image.detach().cpu().clone()}
    # gradient rows a backward of this workload writes (what preprocess_bwd reads): counted on ONE extra untimed step, the only one
    # that runs with the counting switch on
    RZ.RECORD_LIVE_ROWS = True
    tr.step(it0 + args.warmup + args.steps)
    RZ.RECORD_LIVE_ROWS = False
    n_live_timed = RZ.live_rows() if rank == 0 else None
    # the gradient buckets' all-reduce on its own (what one step exchanges; in the step the larger bucket runs under the MLP
    # backward passes)
    allreduce = None
    if world > 1:
        bufs = [torch.zeros(n // 4, device=dev) for n in tr.bucket_bytes()]
        for _ in range(3):
            for b in bufs:
                dist.all_reduce(b)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            for b in bufs:
                dist.all_reduce(b)
        torch.cuda.synchronize()
        allreduce = {"ms_per_step_standalone": 1e3 * (time.perf_counter() - t0) / 20, "bucket_bytes": tr.bucket_bytes(),
                     "backend": dist.get_backend(), "world_size_observed": dist.get_world_size()}

    # strict-fp32 MLP arithmetic on the same workload (N = 1 only: a transparency line, not the headline)
    f32_mode = None
    if world == 1 and mlp_impl == "hip" and not args.no_extras and args.phase == "gs":
        prev = L.lib().dgm_mlp_set_gemm(1)
        try:
            for i in range(5):
                tr.step(it0 + i)
            n32 = 40
            f_dt, _, _, _ = timed(n32, it0 + 5)
            f32_mode = {"value": n32 / f_dt, "unit": "it/s", "ms_per_step": 1e3 * f_dt / n32, "steps": n32,
                        "arithmetic": "v_mfma_f32_32x32x2_f32 (native fp32 MFMA) in every MLP GEMM"}
        finally:
            L.lib().dgm_mlp_set_gemm(prev)

    # densification inside a timed region (R/train.py:489-515: every 100 iterations clone / split / prune + optimizer surgery;
    # the headline region times fixed-P steps only): a second trainer with densify=True, 200 steps that contain two such events
    with_densify = None
    if world == 1 and not args.no_extras and args.phase == "gs":
        try:
            tr2, _ = build_scene(dev, rank, world, mlp_impl, densify=True)
            itd = tr2.opt.warm_up + 2000 + 1          # 5001: the first step is not a densification step
            for i in range(8):
                tr2.step(itd + i)
            # the synthetic targets give smaller view-space gradients than a real scene: the reference's threshold (2e-4) would
            # select nothing, so it is set to the 95th percentile of the accumulated statistic -- ~5 % of the Gaussians are
            # cloned or split at each event, the order of magnitude of a real run's early densification steps
            with torch.no_grad():
                stat = (tr2.g.xyz_gradient_accum / tr2.g.denom.clamp_min(1.0)).reshape(-1)
                thr = float(torch.quantile(stat[stat > 0][:1_000_000], 0.95)) if bool((stat > 0).any()) else tr2.opt.densify_grad_threshold
            tr2.opt.densify_grad_threshold = thr
            tr2.freeze_gc()  # (the second scene's set-up objects out of the cyclic collector's way, like the headline's)
            P0 = int(tr2.g.get_xyz.shape[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_d, events, P_path = 200, 0, [P0]
            for i in range(n_d):
                it = itd + 8 + i
                tr2.step(it)
                if it > tr2.opt.densify_from_iter and it % tr2.opt.densification_interval == 0:
                    events += 1
                    P_path.append(int(tr2.g.get_xyz.shape[0]))
            torch.cuda.synchronize()
            d_dt = time.perf_counter() - t0
            with_densify = {"value": n_d / d_dt, "unit": "it/s", "ms_per_step": 1e3 * d_dt / n_d, "steps": n_d,
                            "densify_events": events, "P": P_path, "densify_grad_threshold": thr,
                            "note": "densify_and_prune every 100 iterations inside the timed region (amortised); threshold = 95th "
                                    "percentile of the accumulated view-space gradient statistic (the reference's 2e-4 selects nothing "
                                    "on synthetic targets); P changes at each event, so the steps after it are not the headline's workload"}
            del tr2
        except Exception as ex:  # an extra must never take the headline down
            with_densify = {"error": str(ex)}

    trained = None
    if rank == 0 and world == 1 and not args.no_extras and args.phase == "gs":
        try:
            trained = trained_like_render_bwd(dev)
        except Exception as ex:  # an extra must never take the headline down
            trained = {"error": str(ex)}

    if rank == 0:
        # R of the last frame (all frames of the synthetic orbit are statistically alike)
        R = int((pkg["radii"] > 0).sum().item())  # visible Gaussians (informational)
        n_inst = n_inst_timed  # tile instances R
        bwd_ms, bwd_n = stages.get("render_bwd", (0.0, 0))
        alg_bytes = 40.0 * n_inst + 20.0 * W * H + 36.0 * P  # render bwd per frame
        achieved = (alg_bytes / (bwd_ms * 1e-3) / 1e9) if bwd_ms > 0 else 0.0
        prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")

        def pmc_traffic(name, kernel=None, launches_per_step=None):
            """HBM bytes per launch from a committed rocprofv3 --pmc pass -- counters cannot be read without the profiler, so this
            is offline data; it is only printed when the file describes THIS run's launch: same workload, same kernel, same row
            count N = P and (where the file records it) the same launches per step.  Returns (bytes, file, commit) or Nones + why."""
            try:
                with open(os.path.join(prof_dir, name)) as fh:
                    pmc = json.load(fh)
            except (OSError, ValueError):
                return None, None, f"{name}: not readable"
            why = None
            if pmc.get("workload") != WORKLOAD:
                why = f"workload {pmc.get('workload')} != {WORKLOAD}"
            elif kernel is not None and pmc.get("kernel", "").split("<")[0] not in kernel:
                why = f"kernel {pmc.get('kernel')} is not this run's {kernel.split(' ')[0]}"
            elif pmc.get("N") is not None and int(pmc["N"]) != P:
                why = f"N {pmc['N']} != {P}"
            elif pmc.get("launches_per_step") is not None and launches_per_step is not None and \
                    abs(float(pmc["launches_per_step"]) - launches_per_step) > 0.02 * launches_per_step:
                why = f"launches per step {pmc['launches_per_step']} != {launches_per_step}"
            elif pmc.get("N") is None or pmc.get("commit") is None:
                why = "the file records neither N nor the commit it was taken at (a pass of an earlier round)"
            if why is not None:
                return None, None, f"profiles/{name} refused: {why}"
            return float(pmc["fetch_bytes"]) + float(pmc["write_bytes"]), name, pmc["commit"]

        gemm_mode = L.lib().dgm_mlp_set_gemm(-1) if mlp_impl == "hip" else -1  # (-1: query, mode unchanged)
        mfma_per_product = 3.0  # f16x3 / f16x3p: 3 MFMAs per fp32 product on the f16 pipe
        layer_flops = 2.0 * P * 256 * 256                       # one 256 -> 256 layer over N = P rows
        layer_bytes = 2.0 * P * 256 * 4 + P * 32 + 256 * 256 * 4  # A in + C out (fp32) + ReLU mask bits + the weights once
        dw_bytes = 2.0 * P * 256 * 4 + 256 * 256 * 4  # X in + G in + the gradient once (the per-CU partial tiles are overhead)
        planes = gemm_mode >= 3
        kn = (("mlp_gemm4_kernel<16,1024,512,0> (256->256 layer forward on planes, N rows)",
               "mlp_gemm4_kernel<16,1024,512,1> (256->256 layer backward-data on planes, N rows)",
               "mlp_dw4_kernel<8,8> (256x256 weight gradient over N rows, planes)") if planes else
              ("mlp_gemm_kernel<0> (fp32 MFMA)", "mlp_gemm_kernel<1> (fp32 MFMA)", "mlp_dw_kernel (fp32 MFMA)"))
        pm = ("r06_pmc_gemm4_fwd.json", None, None) if planes else (None, None, None)
        kern = {  # stage -> (kernel name, algorithmic flops, algorithmic bytes, committed PMC file)
            "mlp_layer_fwd": (kn[0], layer_flops, layer_bytes, pm[0]),
            "mlp_layer_bwd": (kn[1], layer_flops, layer_bytes, pm[1]),
            "mlp_layer_dw": (kn[2], layer_flops, dw_bytes, pm[2]),
            # backward data + weight gradient of one layer in one launch (plane arithmetic): G_l read once algorithmically
            "mlp_bwd_pair": ("mlp_bwd_pair_kernel (256->256 backward-data on part of the CUs + 256x256 weight gradient on the rest)",
                             2.0 * layer_flops, 3.0 * P * 256 * 4 + P * 32 + 2 * 256 * 256 * 4, "r06_pmc_bwd_pair.json"),
            "render_bwd": ("render_bwd4_kernel", 0.0, alg_bytes, "r06_pmc_render_bwd4.json"),
            "render_fwd": ("render_fwd_kernel", 0.0, 40.0 * n_inst + 20.0 * W * H, "r06_pmc_render_fwd.json"),
            "tile_sort": ("tile_sort_radix_kernel (+ the mid / big worklists' launch)", 0.0, 12.0 * n_inst, "r06_pmc_tile_sort_radix.json"),  # 8-byte records in, point_list out
            # 559 B per Gaussian + the 36-byte rows some pixel blended (live rows; the others are neither written nor read)
            "preprocess_bwd": ("preprocess_bwd_kernel", 0.0, 559.0 * P + 36.0 * (n_live_timed if n_live_timed is not None else n_inst),
                               "r06_pmc_preprocess_bwd.json"),
            "preprocess_fwd": ("preprocess_fwd_kernel", 0.0, 311.0 * P, None),
        }
        kernels, best, best_ms = {}, None, -1.0
        for st_name, (kname, fl, by, pmc_file) in kern.items():
            ms, n = stages.get(st_name, (0.0, 0))
            if ms <= 0 or n == 0:
                continue
            per_step = ms * n / args.steps
            hbm_gbs = by / (ms * 1e-3) / 1e9
            rec = {"kernel": kname, "avg_ms": round(ms, 5), "launches_per_step": round(n / args.steps, 2), "ms_per_step": round(per_step, 4),
                   "hbm_GBps": round(hbm_gbs, 1), "frac_hbm": round(hbm_gbs / HBM_PEAK_GBS, 4)}
            if fl > 0:
                tf = fl / (ms * 1e-3) / 1e12
                rec.update({"fp32_equiv_TFLOPs": round(tf, 1), "mfma_issued_TFLOPs": round(mfma_per_product * tf, 1),
                            "frac_mfma_pipe": round(mfma_per_product * tf / MFMA_16BIT_PEAK_TF, 4)})
            kernels[st_name] = rec
            if per_step > best_ms:
                best, best_ms = st_name, per_step
        roof = {"kernel": None, "bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}
        if best is not None:
            r = kernels[best]
            kname, fl, by, pmc_file = kern[best]
            traffic, src, commit = pmc_traffic(pmc_file, kname, r["launches_per_step"]) if pmc_file else (None, None, "no PMC pass committed for this kernel")
            if r.get("frac_mfma_pipe", 0.0) > r["frac_hbm"]:
                roof = {"kernel": kname, "bound": "mfma", "achieved": r["mfma_issued_TFLOPs"], "peak": MFMA_16BIT_PEAK_TF,
                        "unit": "TFLOP/s", "frac": r["frac_mfma_pipe"]}
            else:
                roof = {"kernel": kname, "bound": "hbm", "achieved": r["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac_hbm"]}
            roof.update({"traffic": traffic, "traffic_source": (f"profiles/{src} taken at commit {commit}: rocprofv3 --pmc FETCH_SIZE / "
                                                                 "WRITE_SIZE passes of this workload, kernel, N and launch count, "
                                                                 "committed (counters cannot be read without the profiler: not "
                                                                 "measured in this run)") if src else commit,
                         "algorithmic_bytes": by, "algorithmic_flops": fl, "avg_ms": r["avg_ms"], "launches_per_step": r["launches_per_step"],
                         "ms_per_step": r["ms_per_step"], "frac_hbm": r["frac_hbm"], "frac_mfma_pipe": r.get("frac_mfma_pipe"),
                         "arithmetic": ("f16x3p: activations / gradients stored as 2 binary16 planes with one exponent per 32-row tile, "
                                        "split once by the producer, 3 MFMAs per product" if planes else "native fp32 MFMA")})
        if best is not None and roof.get("bound") == "hbm" and not args.no_extras:
            # context for `frac`: what a plain device-to-device copy moving the same number of bytes (half read, half written) reaches on
            # THIS device in THIS run -- the practical rate of a read + write stream (torch's copy kernel, HIP events on its stream)
            half = int(roof["algorithmic_bytes"] // 8) * 4
            src_t, dst_t = torch.empty(half // 4, device=dev), torch.empty(half // 4, device=dev)
            src_t.normal_()
            for _ in range(3):
                dst_t.copy_(src_t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dst_t.copy_(src_t)
            e1.record()
            torch.cuda.synchronize()
            copy_ms = e0.elapsed_time(e1) / 20
            copy_gbs = 2.0 * half / (copy_ms * 1e-3) / 1e9
            roof["device_copy_of_the_same_bytes"] = {"avg_ms": round(copy_ms, 5), "GBps": round(copy_gbs, 1),
                                                     "kernel_rate_over_copy_rate": round(roof["achieved"] / copy_gbs, 4),
                                                     "note": "informational: torch's device-to-device copy, half of the kernel's "
                                                             "algorithmic bytes read and half written, same run; `frac` stays "
                                                             "against the 8 TB/s of the data sheet"}
            del src_t, dst_t
        rb_traffic, rb_src, _ = pmc_traffic("r06_pmc_render_bwd4.json", "render_bwd4_kernel")
        out = {
            "metric": ("train-step iters/sec (800x800, ~100k Gaussians)" if WORKLOAD == "cfg2"
                       else f"train-step iters/sec ({WORKLOAD}: {W}x{H}, P={P}; informational, the metric is quoted on cfg2)"),
            "value": args.steps * world / elapsed,
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("f32 (f16x3p split products, fp32 accumulate)" if mlp_impl == "hip" and gemm_mode >= 3 else
                                          "f32 (native fp32 MFMA)" if gemm_mode == 1 else "f32"), "data": "synthetic",
            "phase": ("dynamic Gaussian splatting (deform + deform_back)" if args.phase == "gs" else
                      f"mesh co-training: deform, deform_normal, deform_back, deform_back_normal on P + DPSR {args.dpsr_res}^3 (splat, "
                      f"spectral solve, read-back, adjoints) + deform_back and appearance on V={args.verts} vertices; DPSR step included, "
                      "DiffMC / nvdiffrast replaced by a probe of phi (third-party, out of scope)"),
            "config": {"workload": ("D-NeRF jumpingjacks-like cfg2: 800x800, P=100000 Gaussians, deformation MLP on "
                                    "(deform + deform_back, is_blender), 1 frame per rank per step, fixed P (no densification "
                                    "inside the timed region); output heads of both networks scaled x0.01 so that the deformation "
                                    "field is small as after convergence (default init triples every splat's extent)") if WORKLOAD == "cfg2" else
                                   f"{WORKLOAD} of BASELINE.json: {W}x{H}, P={P} Gaussians, deformation MLP on, 1 frame per rank per "
                                   "step, fixed P",
                       "P": P, "W": W, "H": H, "num_rendered": n_inst, "live_rows": n_live_timed, "visible": R, "mlp_impl": mlp_impl,
                       "parallelism": f"dp{world} (frame-parallel, flat-bucket all-reduce {tr.grad_bytes() / 1e6:.1f} MB)"},
            "roofline": roof,
            # the rasterizer backward, graded against HBM by BASELINE.json's north_star (VALU-bound in practice:
            # profiles/*pmc_sq*.json)
            "roofline_render_bwd": {"kernel": "render_bwd4_kernel", "bound": "hbm", "achieved": achieved,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                    "traffic": rb_traffic, "algorithmic_bytes": alg_bytes, "avg_ms": bwd_ms,
                                    "launches": bwd_n},
            "kernels": kernels,
            "stages_ms": {k: round(v[0], 4) for k, v in stages.items()},
            # host side: time inside the rasterizer forward call (synchronous protocol: its R read-back is the step's only sync;
            # DGM_SYNC_FREE=1: the call only enqueues and {R, flags} are looked at after the backward is enqueued -- `settle_wait`)
            "host_ms_per_step": {"blocked_on_gpu": round(1e3 * blocked_s / args.steps, 3),
                                 "busy": round(1e3 * (elapsed - blocked_s) / args.steps, 3),
                                 "device_allocations_in_timed_region": int(dev_allocs),
                                 "sync_free_forward": bool(RZ.SYNC_FREE), "settle_wait": round(1e3 * settle_s / args.steps, 3),
                                 "frames_redone_for_capacity": int(redos)},
        }
        out["rccl_world_size"] = dist.get_world_size() if world > 1 else 1
        if dp is not None:
            out["replicas_identical"] = dp["replicas_identical"]
            out["data_parallel"] = dp
        if allreduce is not None:
            out["allreduce"] = allreduce
        if f32_mode is not None:
            out["mlp_f32_mode"] = f32_mode
        if with_densify is not None:
            out["with_densify"] = with_densify
        if trained is not None:
            out["roofline_render_bwd_trained"] = trained
        # VALU fraction of the blend kernels: lane operations counted by the committed PMC pass OF THE SAME SCENE over the kernel time
        # measured here on that scene, against the 2.4 GHz peak (256 CU x 4 SIMD x 32 lanes = 78.6 T lane-op/s)
        fv = frac_valu_from_profiles("bench") if WORKLOAD == "cfg2" else {}  # (the committed passes are cfg2's)
        for short, st_name in (("render_bwd4_kernel", "render_bwd"), ("render_fwd_kernel", "render_fwd")):
            if short in fv and st_name in out["kernels"]:
                fv[short]["frac_valu"] = fv[short]["valu_lane_ops_per_launch"] / (out["kernels"][st_name]["avg_ms"] * 1e-3 * 78.6e12)
        if fv:
            out["frac_valu"] = fv
        if trained is not None and "avg_ms" in trained:
            fvt = frac_valu_from_profiles("trained")
            for short, key in (("render_bwd4_kernel", "avg_ms"), ("render_fwd_kernel", "render_fwd_ms")):
                if short in fvt and trained.get(key):
                    fvt[short]["frac_valu"] = fvt[short]["valu_lane_ops_per_launch"] / (trained[key] * 1e-3 * 78.6e12)
            if fvt:
                trained["frac_valu"] = fvt
        if steady is not None:
            out["steady_state"] = steady
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(P, W, H, headline_R=n_inst, frame=last_frame, scene=cpu_scene)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        if WORKLOAD == "cfg2":  # informational, offline: the reference's own step on this GPU model (a -m gpu test measures it)
            try:
                import glob
                src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_ref_vs_ours_step.json")))[-1]
                r = json.load(open(src))
                out["reference_same_gpu"] = {"value": r["reference_shaped_it_s"], "unit": "it/s",
                                             "source": f"profiles/{os.path.basename(src)} (committed measurement of "
                                                       "tests/test_gpu_vs_reference.py, not taken in this run)"}
            except Exception:
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
