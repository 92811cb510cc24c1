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
with the largest (average launch time x launches per step); today one 256 -> 256 trunk-layer GEMM of the deformation
MLP (mlp_gemm3p_kernel) -- timed with hipEvents recorded on the launch stream inside the timed region (deferred
read-out, no extra sync) and priced against BOTH roofs it can hit: HBM (algorithmic bytes / 8 TB/s) and the f16
matrix pipe it issues on (3 MFMAs per fp32 product: 3 x flops / 2.5 PFLOP/s); `bound` is the nearer one.
`roofline_render_bwd` is the rasterizer backward, the kernel group BASELINE.json's north_star grades against HBM.
`kernels` lists the same two fractions for every instrumented kernel.  When K < 200 a second, 200-step steady-state
region is timed and reported as `steady_state` (SURVEY.md section 8d asks for >= 200 iterations).
`cpu_baseline` = the same step on the host cores (oracle rasterizer + PyTorch-CPU MLPs; a port of the reference's CPU
path, not the reference itself, which is not on the GPU box), rank 0 at N=1 only, on a bounded sample.
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


def build_scene(dev, rank, world, mlp_impl, n_frames=200, n_gt=4, seed=0):
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
    gts = [torch.tensor(syn.gt_image(W, H, seed=i), device=dev) for i in range(n_gt)]
    cams = [S.TorchCamera(syn.config_camera(WORKLOAD, frame=f, n_frames=n_frames), dev, gts[f % n_gt])
            for f in range(n_frames)]
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
    bg = torch.tensor([1.0, 1.0, 1.0] if c["white_bg"] else [0.0, 0.0, 0.0], device=dev)
    tr = T.Trainer(g, deform, deform_back, cams, background=bg, is_blender=c["is_blender"], rank=rank, world=world,
                   seed=seed)
    return tr, (P, W, H)


def cpu_baseline(P, W, H, max_threads=32):
    """The same train step on the host: oracle rasterizer (C, OpenMP) + the MLPs on PyTorch-CPU.  One step of the
    full cfg2 workload is the bounded sample."""
    syn = importlib.import_module("dg-mesh_amd.synthetic")
    D = importlib.import_module("dg-mesh_amd.deform")
    S = importlib.import_module("dg-mesh_amd.scene")
    from oracle import oracle as orc

    cores = min(os.cpu_count() or 1, max_threads)
    orc.set_threads(cores)
    torch.set_num_threads(cores)
    xyz0 = ((np.random.RandomState(0).rand(P, 3) * 2 - 1) * 1.3).astype(np.float32)  # the points build_scene() draws
    g = syn.make_gaussians(P, seed=0, dist2=orc.knn(xyz0))  # same initialisation as create_from_pcd (simple-knn scales)
    cam = syn.config_camera(WORKLOAD, frame=0)
    gt = torch.tensor(syn.gt_image(W, H, 0))
    torch.manual_seed(0)
    nets = [D.DeformNetworkNormal(is_blender=True, trunk_impl="torch") for _ in range(2)]
    with torch.no_grad():  # same small-deformation heads as the GPU workload (build_scene), so R is comparable
        for m in nets:
            for head in (m.gaussian_warp, m.gaussian_rotation, m.gaussian_scaling, m.gaussian_normal):
                head.weight.mul_(0.01)
                head.bias.mul_(0.01)
    params = [p for n in nets for p in n.parameters()]
    opt = torch.optim.Adam(params, lr=1e-4, eps=1e-15)
    xyz = torch.tensor(g["xyz"])
    t_in = torch.tensor([[0.3]]).expand(P, -1)
    tanx, tany = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    bg = np.ones(3, np.float32)
    orc.lib()

    def one_step():
        for p in params:
            p.grad = None
        d_xyz, d_rot, d_scale, _ = nets[0](xyz, t_in)
        a = syn.activate(g, d_xyz.detach().numpy(), d_rot.detach().numpy(), d_scale.detach().numpy())
        f = orc.forward(bg, a["means3D"], None, a["opacities"], a["scales"], a["rotations"], 1.0, None,
                        cam.world_view_transform, cam.full_proj_transform, tanx, tany, H, W, a["shs"], 3, cam.camera_center)
        img = torch.tensor(f["color"], requires_grad=True)
        loss_img = 0.8 * S.l1_loss(img, gt) + 0.2 * (1.0 - S.ssim(img, gt))
        loss_img.backward()
        gr = orc.backward(f, bg, a["means3D"], None, a["scales"], a["rotations"], 1.0, None, cam.world_view_transform,
                          cam.full_proj_transform, tanx, tany, img.grad.numpy(), a["shs"], 3, cam.camera_center)
        back = nets[1]((xyz + d_xyz).detach(), t_in)
        cyc = (S.l1_loss(-back[0], d_xyz) + S.l1_loss(-back[1], d_rot) + S.l1_loss(-back[2], d_scale)) / 3.0
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
    return {"value": 1.0 / dt, "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"{n_steps} full {WORKLOAD} train steps ({W}x{H}, P={P}, R={f['num_rendered']}): oracle rasterizer "
                      f"fwd+bwd (C/OpenMP) + 2 deformation MLPs fwd+bwd + L1/SSIM + Adam on PyTorch-CPU, {dt:.1f} s each"}


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
    args = ap.parse_args()
    global WORKLOAD
    WORKLOAD = args.workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    tr, (P, W, H) = build_scene(dev, rank, world, mlp_impl)
    it0 = tr.opt.warm_up + 2000  # "deformation MLP on" phase (warm_up <= it < dpsr_iter)

    for i in range(10):  # allocator / code-object / clock priming (untimed, not part of the W warm-up steps requested below)
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
        t0 = time.perf_counter()
        pk = None
        for i in range(n_steps):
            _, pk = tr.step(first_it + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = L.collect_stage_ms()
        L.lib().dgm_set_profiling(0)
        blocked = RZ.FORWARD_CALL_SECONDS
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, st, blocked, pk

    elapsed, stages, blocked_s, pkg = timed(args.steps, it0 + args.warmup)
    steady = None
    if args.steps < STEADY_STEPS:
        s_dt, _, _, _ = timed(STEADY_STEPS, it0 + args.warmup + args.steps)
        steady = {"steps": STEADY_STEPS, "value": STEADY_STEPS * world / s_dt, "ms_per_step": 1e3 * s_dt / STEADY_STEPS}

    if rank == 0:
        # R of the last frame (all frames of the synthetic orbit are statistically alike)
        R = int((pkg["radii"] > 0).sum().item())  # visible Gaussians (informational)
        n_inst = int(importlib.import_module("dg-mesh_amd.rasterizer").LAST_NUM_RENDERED)  # tile instances R
        bwd_ms, bwd_n = stages.get("render_bwd", (0.0, 0))
        alg_bytes = 40.0 * n_inst + 20.0 * W * H + 36.0 * P  # render bwd per frame
        achieved = (alg_bytes / (bwd_ms * 1e-3) / 1e9) if bwd_ms > 0 else 0.0
        prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")

        def pmc_traffic(name):  # HBM bytes per launch from the committed rocprofv3 --pmc passes (same workload); offline data
            try:
                with open(os.path.join(prof_dir, name)) as fh:
                    pmc = json.load(fh)
                if pmc.get("workload") == WORKLOAD:
                    return float(pmc["fetch_bytes"]) + float(pmc["write_bytes"]), name
            except (OSError, ValueError, KeyError):
                pass
            return None, None

        gemm_mode = L.lib().dgm_mlp_set_gemm(-1) if mlp_impl == "hip" else -1  # (-1: query, mode unchanged)
        f16x3 = gemm_mode in (2, 3)
        mfma_per_product = 3.0 if f16x3 else 6.0  # f16x3: 3 MFMAs per fp32 product on the f16 pipe; bf16x6: 6 on the bf16 pipe
        layer_flops = 2.0 * P * 256 * 256                       # one 256 -> 256 layer over N = P rows
        layer_bytes = 2.0 * P * 256 * 4 + P * 32 + 256 * 256 * 4  # A in + C out (fp32) + ReLU mask bits + the weights once
        dw_bytes = 2.0 * P * 256 * 4 + 256 * 256 * 4  # X in + G in + the gradient once (the per-CU partial tiles are overhead)
        kern = {  # stage -> (kernel name, algorithmic flops, algorithmic bytes, committed PMC file)
            "mlp_layer_fwd": ("mlp_gemm3p_kernel<0> (256->256 layer forward, N rows)" if f16x3 else "mlp_gemm6r_kernel<0,16,1,8>", layer_flops, layer_bytes, "pmc_gemm3r_fwd.json"),
            "mlp_layer_bwd": ("mlp_gemm3p_kernel<1> (256->256 layer backward-data, N rows)" if f16x3 else "mlp_gemm6r_kernel<1,16,1,8>", layer_flops, layer_bytes, "pmc_gemm3r_bwd.json"),
            "mlp_layer_dw": ("mlp_dw3b_kernel (256x256 weight gradient over N rows)" if f16x3 else "mlp_dw6b_kernel", layer_flops, dw_bytes, "pmc_dw3b.json"),
            "render_bwd": ("render_bwd3_kernel", 0.0, alg_bytes, "pmc_render_bwd3.json"),
            "render_fwd": ("render_fwd_kernel", 0.0, 40.0 * n_inst + 20.0 * W * H, "pmc_render_fwd.json"),
            "tile_sort": ("tile_sort_radix_kernel (+ mid / big worklists)", 0.0, 24.0 * n_inst, "pmc_tile_sort_radix.json"),  # 16-byte records in, point_list + upos out
            "preprocess_bwd": ("preprocess_bwd_kernel", 0.0, 559.0 * P + 48.0 * n_inst, None),
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
            traffic, src = pmc_traffic(pmc_file) if pmc_file else (None, None)
            if r.get("frac_mfma_pipe", 0.0) > r["frac_hbm"]:
                roof = {"kernel": kname, "bound": "mfma", "achieved": r["mfma_issued_TFLOPs"], "peak": MFMA_16BIT_PEAK_TF,
                        "unit": "TFLOP/s", "frac": r["frac_mfma_pipe"]}
            else:
                roof = {"kernel": kname, "bound": "hbm", "achieved": r["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac_hbm"]}
            roof.update({"traffic": traffic, "traffic_source": (f"profiles/{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                                                 "workload, committed (not measured in this run)") if src else None,
                         "algorithmic_bytes": by, "algorithmic_flops": fl, "avg_ms": r["avg_ms"], "launches_per_step": r["launches_per_step"],
                         "ms_per_step": r["ms_per_step"], "frac_hbm": r["frac_hbm"], "frac_mfma_pipe": r.get("frac_mfma_pipe"),
                         "arithmetic": ("f16x3: fp32 operands as 2 power-of-two-scaled binary16 planes, 3 MFMAs per product"
                                        if f16x3 else "bf16x6: 3 bf16 planes, 6 MFMAs per product")})
        rb_traffic, rb_src = pmc_traffic("pmc_render_bwd3.json")
        out = {
            "metric": ("train-step iters/sec (800x800, ~100k Gaussians)" if WORKLOAD == "cfg2"
                       else f"train-step iters/sec ({WORKLOAD}: {W}x{H}, P={P}; informational, the metric is quoted on cfg2)"),
            "value": args.steps * world / elapsed,
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("D-NeRF jumpingjacks-like cfg2: 800x800, P=100000 Gaussians, deformation MLP on "
                                    "(deform + deform_back, is_blender), 1 frame per rank per step, fixed P (no densification "
                                    "inside the timed region)") if WORKLOAD == "cfg2" else
                                   f"{WORKLOAD} of BASELINE.json: {W}x{H}, P={P} Gaussians, deformation MLP on, 1 frame per rank per "
                                   "step, fixed P",
                       "P": P, "W": W, "H": H, "num_rendered": n_inst, "visible": R, "mlp_impl": mlp_impl,
                       "parallelism": f"dp{world} (frame-parallel, flat-bucket all-reduce {tr.grad_bytes() / 1e6:.1f} MB)"},
            "roofline": roof,
            # the rasterizer backward, graded against HBM by BASELINE.json's north_star (VALU-bound in practice:
            # profiles/*pmc_sq*.json)
            "roofline_render_bwd": {"kernel": "render_bwd3_kernel", "bound": "hbm", "achieved": achieved,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                    "traffic": rb_traffic, "algorithmic_bytes": alg_bytes, "avg_ms": bwd_ms,
                                    "launches": bwd_n},
            "kernels": kernels,
            "stages_ms": {k: round(v[0], 4) for k, v in stages.items()},
            # host side: time blocked in the rasterizer forward (its R read-back is the step's only sync) vs busy
            "host_ms_per_step": {"blocked_on_gpu": round(1e3 * blocked_s / args.steps, 3),
                                 "busy": round(1e3 * (elapsed - blocked_s) / args.steps, 3)},
        }
        if steady is not None:
            out["steady_state"] = steady
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(P, W, H)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        if WORKLOAD == "cfg2":  # informational, offline: the reference's own step on this GPU model (a -m gpu test measures it)
            try:
                r = json.load(open(os.path.join(ROOT, "profiles", "r02_ref_vs_ours_step.json")))
                out["reference_same_gpu"] = {"value": r["reference_shaped_it_s"], "unit": "it/s",
                                             "source": "profiles/r02_ref_vs_ours_step.json (committed measurement of "
                                                       "tests/test_gpu_vs_reference.py, not taken in this run)"}
            except Exception:
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
