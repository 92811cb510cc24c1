"""autograd.Function over the fused MLP entry points of libdgmesh_hip.so (dgm_mlp_forward / dgm_mlp_backward)."""
import ctypes

import torch

from . import _lib


def _params(net, W, b, Wh, bh, t_dim):
    p = _lib.MlpParams()
    p.n_layers, p.width, p.emb_dim, p.t_dim, p.skip_layer, p.n_out = 8, 256, 63 + t_dim, t_dim, 5, Wh.shape[0]
    for l in range(8):
        p.W[l] = W[l].data_ptr()
        p.b[l] = b[l].data_ptr()
    p.Wh, p.bh = Wh.data_ptr(), bh.data_ptr()
    return p


class _MLPFunction(torch.autograd.Function):
    """inputs: x (N,3) [no grad], t_emb ((1,T) when broadcast else (N,T)), Wh, bh, W0..W7, b0..b7."""

    @staticmethod
    def forward(ctx, x, t_emb, bcast, Wh, bh, *wb):
        L = _lib.lib()
        W, b = [w.contiguous() for w in wb[:8]], [v.contiguous() for v in wb[8:]]
        x, t_emb, Wh, bh = x.contiguous(), t_emb.contiguous(), Wh.contiguous(), bh.contiguous()
        N, T = x.shape[0], t_emb.shape[1]
        ws = torch.empty(L.dgm_mlp_workspace_bytes(N), dtype=torch.uint8, device=x.device)
        out = torch.empty((N, Wh.shape[0]), dtype=torch.float32, device=x.device)
        p = _params(None, W, b, Wh, bh, T)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        with torch.cuda.device(x.device):
            _lib.check(L.dgm_mlp_forward(ctypes.byref(p), N, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(t_emb.data_ptr()),
                                         0 if bcast else T, ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()), st))
        ctx.save_for_backward(ws, Wh, bh, *W, *b)
        ctx.meta = (N, T, bool(bcast), t_emb.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dOut):
        L = _lib.lib()
        ws, Wh, bh, *wb = ctx.saved_tensors
        W, b = wb[:8], wb[8:]
        N, T, bcast, need_t = ctx.meta
        dOut = dOut.contiguous()
        dev = dOut.device
        dW = [torch.empty_like(w) for w in W]
        db = [torch.empty_like(v) for v in b]
        dWh, dbh = torch.empty_like(Wh), torch.empty_like(bh)
        dtemb = torch.empty((1, T) if bcast else (N, T), dtype=torch.float32, device=dev)
        p = _params(None, W, b, Wh, bh, T)
        arr = ctypes.c_void_p * 8
        dWp = arr(*[t.data_ptr() for t in dW])
        dbp = arr(*[t.data_ptr() for t in db])
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(L.dgm_mlp_backward(ctypes.byref(p), N, ctypes.c_void_p(dOut.data_ptr()), 0 if bcast else T,
                                          ctypes.c_void_p(ws.data_ptr()), dWp, dbp, ctypes.c_void_p(dWh.data_ptr()),
                                          ctypes.c_void_p(dbh.data_ptr()), ctypes.c_void_p(dtemb.data_ptr()), st))
        return (None, dtemb if need_t else None, None, dWh, dbh, *dW, *db)


def network_forward(net, heads, x, t_emb, bcast):
    if not x.is_cuda:
        raise RuntimeError("trunk_impl='hip' needs CUDA/HIP tensors (dg-mesh_amd has no CPU path for its kernels)")
    Wh = torch.cat([m.weight for m in heads], 0)
    bh = torch.cat([m.bias for m in heads], 0)
    W = [l.weight for l in net.linear]
    b = [l.bias for l in net.linear]
    return _MLPFunction.apply(x.detach() if not x.requires_grad else x, t_emb, bcast, Wh, bh, *W, *b)
