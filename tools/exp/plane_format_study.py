"""CPU study for the next structural step of the MLP: what does ONE power-of-two scale per
TENSOR (so that activations / gradients could live in HBM as the two binary16 planes, split once by their producer) cost in
accuracy against today's exact per-row / per-column scales?   python tools/exp/plane_format_study.py [N]

The trunk of DeformNetworkNormal (reference shapes, seed 0) runs forward + backward in fp64 (the truth) on N seeded rows; the
three GEMM kinds of one layer are then re-evaluated with the f16x3 arithmetic emulated in numpy (h = rne16(s x), l = rne16(s x -
h); ah*bh + ah*bl + al*bh accumulated in fp32) under both scale choices.  The gradient entering the heads has rows spread over
ten decades plus exact zero rows, like dL/d(delta) of barely visible / invisible Gaussians.  Metric: max |err| / max |truth|
per tensor (what the parity tests bound by 1e-4), and the same per ROW for the row-wise outputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib

D = importlib.import_module("dg-mesh_amd.deform")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000


def pow2_scale(mx):
    """2^(14 - floor(log2 max)) as mlp_f16x3.hpp: scale_from_max_bits (max -> [2^14, 2^15))."""
    mx = np.maximum(mx, 1e-30)
    return np.exp2(14 - np.floor(np.log2(mx))).astype(np.float32)


def split(x, s):
    xs = (x.astype(np.float32) * s).astype(np.float32)
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)


def gemm3(a, sa, b, sb):
    """(a * sa) @ (b * sb) in the three-product form, unscaled again.  sa / sb broadcast against a / b."""
    ah, al = split(a, sa)
    bh, bl = split(b, sb)
    acc = ah @ bh + ah @ bl + al @ bh  # fp32
    return acc


def rel(err, ref):
    return float(np.abs(err).max() / max(np.abs(ref).max(), 1e-300))


torch.manual_seed(0)
net = D.DeformNetworkNormal(is_blender=True, trunk_impl="torch").double()
rng = np.random.RandomState(0)
x = torch.tensor((rng.rand(N, 3) * 2 - 1) * 1.3, dtype=torch.float64)
t = torch.full((N, 1), 0.3, dtype=torch.float64)
# layer activations in fp64
t_emb, _ = net._time_rows(t)
emb = torch.cat([D.positional_encoding(x, net.multires), t_emb.expand(N, -1) if t_emb.shape[0] == 1 else t_emb], -1)
hs, h = [], emb
for i, lin in enumerate(net.linear):
    h = torch.relu(lin(h))
    hs.append(h)
    if i in net.skips:
        h = torch.cat([emb, h], -1)
Y6, Y7 = hs[6].detach().numpy(), hs[7].detach().numpy()
W7 = net.linear[7].weight.detach().numpy()  # (256 out, 256 in)
# gradient entering layer 7's output: rows over ten decades, 30 % exact zeros
mag = 10.0 ** rng.uniform(-12, -2, size=(N, 1))
mag[rng.rand(N, 1) < 0.3] = 0.0
G7 = (rng.randn(N, 256) * mag) * (Y7 > 0)

truth = dict(fwd=np.maximum(Y6 @ W7.T, 0), bwd=G7 @ W7, dw=G7.T @ Y6)

res = {}
for name in ("exact per-row / per-column scales (today)", "one scale per tensor"):
    per_tensor = name.startswith("one")
    sY = pow2_scale(np.abs(Y6).max()) if per_tensor else pow2_scale(np.abs(Y6).max(1, keepdims=True))
    sG = pow2_scale(np.abs(G7).max()) if per_tensor else pow2_scale(np.abs(G7).max(1, keepdims=True))
    sWc = pow2_scale(np.abs(W7).max(1, keepdims=True))  # per output column of W^T  (weights: prepared once, as today)
    sWk = pow2_scale(np.abs(W7).max(0, keepdims=True))  # per column of W           (backward data)
    # forward: Y7 = relu(Y6 W7^T)
    f = gemm3(Y6, sY, W7.T, sWc.T) / (np.broadcast_to(sY, (N, 1)) * sWc.T)
    f = np.maximum(f, 0)
    # backward data: dY6 = G7 W7
    b = gemm3(G7, sG, W7, sWk) / (np.broadcast_to(sG, (N, 1)) * sWk)
    # weight gradient: dW7 = G7^T Y6   (contraction over rows: scales must be per column of G7 and of Y6)
    if per_tensor:
        cG, cY = sG, sY
    else:
        cG, cY = pow2_scale(np.abs(G7).max(0, keepdims=True)), pow2_scale(np.abs(Y6).max(0, keepdims=True))
    gh, gl = split(G7, cG)
    yh, yl = split(Y6, cY)
    dw = (gh.T @ yh + gh.T @ yl + gl.T @ yh) / (np.broadcast_to(cG, (1, 256)).T * np.broadcast_to(cY, (1, 256)))
    rowrel = np.abs(b - truth["bwd"]).max(1) / np.maximum(np.abs(truth["bwd"]).max(1), 1e-300)
    live = np.abs(truth["bwd"]).max(1) > 0
    res[name] = dict(fwd=rel(f - truth["fwd"], truth["fwd"]), bwd=rel(b - truth["bwd"], truth["bwd"]),
                     dw=rel(dw - truth["dw"], truth["dw"]), bwd_row_median=float(np.median(rowrel[live])),
                     bwd_row_p99=float(np.percentile(rowrel[live], 99)), bwd_row_max=float(rowrel[live].max()))
f32 = dict(fwd=rel(np.maximum(Y6.astype(np.float32) @ W7.T.astype(np.float32), 0) - truth["fwd"], truth["fwd"]),
           bwd=rel(G7.astype(np.float32) @ W7.astype(np.float32) - truth["bwd"], truth["bwd"]),
           dw=rel(G7.T.astype(np.float32) @ Y6.astype(np.float32) - truth["dw"], truth["dw"]))
print(f"N = {N}; errors as max|err| / max|truth| per tensor (parity bound 1e-4); plain fp32 GEMM for scale:")
print(f"  fp32 GEMM                                  fwd {f32['fwd']:.2e}  bwd-data {f32['bwd']:.2e}  dW {f32['dw']:.2e}")
for k, v in res.items():
    print(f"  {k:42s} fwd {v['fwd']:.2e}  bwd-data {v['bwd']:.2e}  dW {v['dw']:.2e}   "
          f"bwd-data per row: median {v['bwd_row_median']:.1e}  p99 {v['bwd_row_p99']:.1e}  max {v['bwd_row_max']:.1e}")
