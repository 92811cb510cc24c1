"""MLP-only micro benchmark (one DeformNetworkNormal, fwd+bwd, N rows): python tools/mlp_bench.py [N] [iters] [impl]"""
import sys, os, importlib, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D = importlib.import_module("dg-mesh_amd.deform")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
impl = sys.argv[3] if len(sys.argv) > 3 else "hip"
dev = "cuda"
torch.manual_seed(0)
net = D.DeformNetworkNormal(is_blender=True, trunk_impl=impl).to(dev)
if os.environ.get("DGM_BENCH_ZERO") == "1":  # power experiment: all-zero operands (same instruction stream, no toggling)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.zero_()
x = (torch.rand(N, 3, device=dev) * 2 - 1) * 1.3
t = torch.tensor([[0.3]], device=dev).expand(N, -1)
w = torch.randn(N, 13, device=dev)
def step():
    for p in net.parameters():
        p.grad = None
    o = torch.cat(net(x, t), -1)
    (o * w).sum().backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
flops = 3 * 2 * 520704 * N
print(f"impl={impl} N={N}: {dt*1e3:.3f} ms per fwd+bwd, {flops/dt/1e12:.1f} TFLOP/s (of 157.3 fp32 MFMA peak: {flops/dt/157.3e12:.1%})")

# phase timers of a -DP4_TIMING variant build (tools/build_variant.sh): cycles per wave and phase of the last layer-GEMM launch
import ctypes
L = importlib.import_module("dg-mesh_amd._lib").lib()
if hasattr(L, "dgm_p4_timing"):
    buf = (ctypes.c_ulonglong * (256 * 8 * 16))()
    L.dgm_p4_timing(buf)
    full = np.array(buf, dtype=np.float64).reshape(256, 8, 16)
    a = full[..., :8]
    t_in, t_out, nt = full[..., 10], full[..., 11], full[..., 12]
    print(f"prologue parts: copies+exps+bias issue {full[..., 13].mean():.0f}, weight loads issue {full[..., 14].mean():.0f}, wait {full[..., 15].mean():.0f}")
    print(f"prologue (entry -> loop) {full[..., 8].mean():.0f} cycles; wave lifetime {full[..., 9].mean():.0f} (min {full[..., 9].min():.0f} max {full[..., 9].max():.0f}); "
          f"kernel span first entry -> last exit {t_out.max() - t_in.min():.0f}; entry spread {t_in.max() - t_in.min():.0f}; "
          f"lifetime of 13-tile WGs {full[..., 9][nt == nt.max()].mean():.0f} vs others {full[..., 9][nt < nt.max()].mean() if (nt < nt.max()).any() else 0:.0f}")
    names = ["k-loop 1st half (+E1)", "pre-barrier (max, S reads)", "vmcnt(0) wait", "mid barrier", "burst issue + exponent",
             "k-loop 2nd half (+E2) + sums", "end barrier", "step head"]
    tot = a.sum(-1).mean()
    print("phase cycles per wave (mean over 256 WG x 8 waves; min / max of wave means):")
    for k, n in enumerate(names):
        print(f"   {n:32s} {a[..., k].mean():9.0f}  ({100 * a[..., k].mean() / tot:4.1f}%)   wave w min/max {a[..., k].mean(0).min():.0f} / {a[..., k].mean(0).max():.0f}")
    print(f"   total {tot:.0f} cycles per wave")
