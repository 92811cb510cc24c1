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
