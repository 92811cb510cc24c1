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


def _adjacent(ts):
    """True when the tensors lie back to back in memory, in order (fp32, contiguous)."""
    at = ts[0].data_ptr()
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != at:
            return False
        at += 4 * t.numel()
    return True


def _one_storage(ts):
    """All tensors live in ONE storage that reaches to the end of the last one (separately allocated tensors can happen to sit
    back to back in the allocator's address space: as_strided over the first one's storage would then overrun it)."""
    st = ts[0].untyped_storage()
    if any(t.untyped_storage().data_ptr() != st.data_ptr() for t in ts):
        return False
    need = (ts[0].storage_offset() + sum(t.numel() for t in ts)) * ts[0].element_size()
    return st.nbytes() >= need


def _stacked(ts, shape):
    """The head tensors as ONE (n_out, ...) tensor: a view of their common buffer when pack_heads() laid them out back to
    back (no launch), else a concatenation."""
    if _adjacent(ts) and _one_storage(ts):
        stride = (shape[1], 1) if len(shape) == 2 else (1,)
        return torch.as_strided(ts[0].detach(), shape, stride, ts[0].storage_offset())
    return torch.cat([t.detach() for t in ts], 0).contiguous()


def pack_heads(heads):
    """Re-home the weights (and the biases) of the head modules in one buffer each, in output order.  The Parameters keep
    their identity, names and shapes (state_dict / optimizers are unaffected); the fused kernels then read [Wh | bh] in
    place instead of through two torch.cat launches per forward.  Moving the module to another device afterwards undoes
    the layout (harmless: _stacked() falls back to concatenating)."""
    with torch.no_grad():
        for ts in ([m.weight for m in heads], [m.bias for m in heads]):
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            at = 0
            for t in ts:
                t.data = flat[at:at + t.numel()].view(t.shape)
                at += t.numel()


class _MLPFunction(torch.autograd.Function):
    """inputs: x (N,3), t_emb ((1,T) when broadcast else (N,T)), the number of heads n, then the n head weights, the n head
    biases, W0..W7, b0..b7.  x gets a gradient when it asks for one (dgm_mlp_backward_dx: the appearance network on mesh
    vertices moved by deform_back, R/utils/renderer.py:179-181); the training loop's networks detach it (R/train.py:156)."""

    @staticmethod
    def forward(ctx, x, t_emb, bcast, n_heads, *tensors):
        L = _lib.lib()
        hw, hb, wb = tensors[:n_heads], tensors[n_heads:2 * n_heads], tensors[2 * n_heads:]
        n_out = sum(w.shape[0] for w in hw)
        Wh, bh = _stacked(hw, (n_out, hw[0].shape[1])), _stacked(hb, (n_out,))
        W, b = [w.contiguous() for w in wb[:8]], [v.contiguous() for v in wb[8:]]
        x, t_emb = x.contiguous(), t_emb.contiguous()
        N, T = x.shape[0], t_emb.shape[1]
        ws = torch.empty(L.dgm_mlp_workspace_bytes(N), dtype=torch.uint8, device=x.device)
        out = torch.empty((N, Wh.shape[0]), dtype=torch.float32, device=x.device)
        p = _params(None, W, b, Wh, bh, T)
        st = _lib.stream_ptr()
        with _lib.device_guard(x.device):
            _lib.check(L.dgm_mlp_forward(ctypes.byref(p), N, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(t_emb.data_ptr()),
                                         0 if bcast else T, ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()), st))
        ctx.save_for_backward(ws, Wh, bh, x, *W, *b)
        ctx.meta = (N, T, bool(bcast), t_emb.requires_grad, [w.shape[0] for w in hw])
        return out

    @staticmethod
    def backward(ctx, dOut):
        L = _lib.lib()
        ws, Wh, bh, x, *wb = ctx.saved_tensors
        W, b = wb[:8], wb[8:]
        N, T, bcast, need_t, head_rows = ctx.meta
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
        st = _lib.stream_ptr()
        dX = None
        with _lib.device_guard(dev):
            if ctx.needs_input_grad[0]:
                dX = torch.empty((N, 3), dtype=torch.float32, device=dev)
                _lib.check(L.dgm_mlp_backward_dx(ctypes.byref(p), N, ctypes.c_void_p(dOut.data_ptr()), 0 if bcast else T,
                                                 ctypes.c_void_p(ws.data_ptr()), dWp, dbp, ctypes.c_void_p(dWh.data_ptr()),
                                                 ctypes.c_void_p(dbh.data_ptr()), ctypes.c_void_p(dtemb.data_ptr()),
                                                 ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dX.data_ptr()), st))
            else:
                _lib.check(L.dgm_mlp_backward(ctypes.byref(p), N, ctypes.c_void_p(dOut.data_ptr()), 0 if bcast else T,
                                              ctypes.c_void_p(ws.data_ptr()), dWp, dbp, ctypes.c_void_p(dWh.data_ptr()),
                                              ctypes.c_void_p(dbh.data_ptr()), ctypes.c_void_p(dtemb.data_ptr()), st))
        # the heads' gradients: row blocks of the two stacked buffers (contiguous views, no copies)
        dWs, dbs = torch.split(dWh, head_rows, 0), torch.split(dbh, head_rows, 0)
        return (dX, dtemb if need_t else None, None, None, *dWs, *dbs, *dW, *db)


class _TimeNetFunction(torch.autograd.Function):
    """t (one device float) -> timenet(PE(t)) as a (1, n_out) row; gradients for the four timenet tensors only."""

    @staticmethod
    def forward(ctx, t, n_freq, W1, b1, W2, b2):
        L = _lib.lib()
        W1, b1, W2, b2 = W1.contiguous(), b1.contiguous(), W2.contiguous(), b2.contiguous()
        t = t.reshape(-1)[:1].contiguous().float()
        hidden, n_out = W1.shape[0], W2.shape[0]
        if W1.shape[1] != 2 * n_freq + 1 or W2.shape[1] != hidden:
            raise RuntimeError("timenet: weight shapes do not match PE(t) / hidden width")
        save = torch.empty(2 * n_freq + 1 + hidden, dtype=torch.float32, device=t.device)
        out = torch.empty((1, n_out), dtype=torch.float32, device=t.device)
        st = _lib.stream_ptr()
        vp = lambda x: ctypes.c_void_p(x.data_ptr())
        with _lib.device_guard(t.device):
            _lib.check(L.dgm_timenet_forward(vp(t), n_freq, vp(W1), vp(b1), hidden, vp(W2), vp(b2), n_out, vp(save), vp(out), st))
        ctx.save_for_backward(save, W1, W2)
        ctx.n_freq = n_freq
        return out

    @staticmethod
    def backward(ctx, d_out):
        L = _lib.lib()
        save, W1, W2 = ctx.saved_tensors
        d_out = d_out.contiguous()
        hidden, n_out = W1.shape[0], W2.shape[0]
        dW1, db1 = torch.empty_like(W1), torch.empty(hidden, dtype=torch.float32, device=W1.device)
        dW2, db2 = torch.empty_like(W2), torch.empty(n_out, dtype=torch.float32, device=W1.device)
        st = _lib.stream_ptr()
        vp = lambda x: ctypes.c_void_p(x.data_ptr())
        with _lib.device_guard(W1.device):
            _lib.check(L.dgm_timenet_backward(vp(d_out), ctx.n_freq, vp(W2), hidden, n_out, vp(save), vp(dW1), vp(db1), vp(dW2),
                                              vp(db2), st))
        return None, None, dW1, db1, dW2, db2


def time_row(net, t_row):
    """timenet(PE(t)) of the is_blender networks for the single time value in t_row ((1, 1) device tensor)."""
    lin1, lin2 = net.timenet[0], net.timenet[2]
    return _TimeNetFunction.apply(t_row, net.t_multires, lin1.weight, lin1.bias, lin2.weight, lin2.bias)


def check_supported(net, heads, t_emb):
    """The fused kernels are specialised to the reference's only trunk shape (D=8, W=256, PE(x) with 10 frequencies, skip
    after layer 4: R/utils/time_utils.py:60-103).  Anything else would make them read the weight tensors with wrong
    strides, so refuse it here -- the C side cannot see the module."""
    T = t_emb.shape[1]
    key = (T, tuple(m.weight.shape[0] for m in heads))
    if net.__dict__.get("_dgm_supported") == key:  # (checked once per network and head set: 30 us of attribute look-ups per call)
        return
    in0 = 63 + T
    ok = (getattr(net, "D", None) == 8 and getattr(net, "W", None) == 256 and getattr(net, "multires", None) == 10
          and list(getattr(net, "skips", [])) == [4] and len(net.linear) == 8 and in0 <= 96)
    if ok:
        for l, lin in enumerate(net.linear):
            want = (256, in0) if l == 0 else ((256, 256 + in0) if l == 5 else (256, 256))
            ok = ok and tuple(lin.weight.shape) == want and lin.bias is not None and tuple(lin.bias.shape) == (256,)
        for m in heads:
            ok = ok and m.weight.shape[1] == 256 and m.bias is not None
        ok = ok and 1 <= sum(m.weight.shape[0] for m in heads) <= 16
    if not ok:
        raise RuntimeError("trunk_impl='hip' supports the reference trunk only (D=8, W=256, multires=10, skips=[4], at most "
                           "16 head outputs); build the network with trunk_impl='torch' for other shapes")
    net.__dict__["_dgm_supported"] = key


def network_forward(net, heads, x, t_emb, bcast):
    if not x.is_cuda:
        raise RuntimeError("trunk_impl='hip' needs CUDA/HIP tensors (dg-mesh_amd has no CPU path for its kernels)")
    if x.requires_grad and not bcast:
        raise RuntimeError("trunk_impl='hip' differentiates w.r.t. the positions only with a broadcast time row (the plane "
                           "arithmetic); use trunk_impl='torch' for per-row time inputs")
    if x.requires_grad and _lib.lib().dgm_mlp_set_gemm(-1) not in (3, 4, 5):  # (-1: query) dgm_mlp_backward_dx exists in the plane arithmetic only
        raise RuntimeError("trunk_impl='hip' differentiates w.r.t. the positions only in the default f16x3p arithmetic "
                           "(DGM_MLP_GEMM / dgm_mlp_set_gemm select another one): use trunk_impl='torch' for this network")
    check_supported(net, heads, t_emb)
    W = [l.weight for l in net.linear]
    b = [l.bias for l in net.linear]
    return _MLPFunction.apply(x, t_emb, bcast, len(heads), *[m.weight for m in heads], *[m.bias for m in heads], *W, *b)
