import sys, os, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D = importlib.import_module("dg-mesh_amd.deform")
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.RandomState(1)
x = torch.tensor(((rng.rand(N, 3) * 2 - 1) * 1.3).astype(np.float32), device=dev)
t = torch.tensor(rng.rand(N, 1).astype(np.float32), device=dev)
nets = {}
for impl in ("torch", "hip"):
    torch.manual_seed(0)
    nets[impl] = D.DeformNetworkNormal(is_blender=True, trunk_impl=impl).to(dev)
w = [torch.randn(N, k, device=dev) for k in (3, 4, 3, 3)]
res = {}
for impl, net in nets.items():
    temb = net.time_embedding(t).detach().requires_grad_(True)
    heads = net.head_modules()
    if impl == "hip":
        mh = importlib.import_module("dg-mesh_amd.mlp_hip")
        o = mh.network_forward(net, heads, x, temb, False)
    else:
        xe = D.positional_encoding(x, 10)
        h = torch.cat([xe, temb], -1)
        for i in range(8):
            h = torch.relu(net.linear[i](h))
            if i == 4:
                h = torch.cat([xe, temb, h], -1)
        o = torch.cat([m(h) for m in heads], -1)
    (o * torch.cat(w, -1)).sum().backward()
    res[impl] = (o.detach(), temb.grad.clone())
a, b = res["hip"], res["torch"]
print("out err", ((a[0] - b[0]).abs().max() / b[0].abs().max()).item())
g1, g2 = a[1], b[1]
print("dtemb max", g2.abs().max().item(), "err", ((g1 - g2).abs().max() / g2.abs().max()).item())
err = (g1 - g2).abs().max(0).values / g2.abs().max()
print("per-column err", [round(v, 4) for v in err.tolist()])
print("row0 hip", g1[0, :6].tolist(), "\nrow0 ref", g2[0, :6].tolist())

rowerr = (g1 - g2).abs().max(1).values / g2.abs().max()
bad = (rowerr > 1e-4).nonzero().flatten()
print("rows with err>1e-4:", bad.numel(), "of", N, "first", bad[:10].tolist(), "last", bad[-5:].tolist())
for (n, p), (_, q) in zip(nets["hip"].named_parameters(), nets["torch"].named_parameters()):
    if p.grad is not None and q.grad is not None:
        e = ((p.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-30)).item()
        if e > 1e-4: print("  param", n, "err", e)
