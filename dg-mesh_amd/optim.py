"""torch.optim.Adam semantics over many optimizers / parameter groups in ONE kernel launch (dgm_adam_step).

The reference ends every iteration with `gaussians.optimizer.step(); deform.optimizer.step(); ...`
(R/train.py:518-524), each a torch.optim.Adam(lr=0.0, eps=1e-15) whose group learning rates are rewritten every
iteration by `update_learning_rate`.  MultiAdam keeps those optimizer objects as the source of truth for hyper
parameters (it reads `param_groups[*]["lr"]`, betas, eps at every step, so the reference's lr schedulers keep working)
but owns the moments itself and applies the update to every tensor of every group with a single HIP kernel.
No CPU / PyTorch fallback: it needs the HIP library.
"""
import ctypes

import torch

from . import _lib


class MultiAdam:
    def __init__(self, optimizers):
        self.optimizers = list(optimizers)
        self.state = {}  # id(param) -> [exp_avg, exp_avg_sq, step]
        for o in self.optimizers:
            for g in o.param_groups:
                if g.get("amsgrad") or g.get("weight_decay", 0) != 0 or g.get("maximize"):
                    raise ValueError("MultiAdam implements plain Adam only (amsgrad / weight_decay / maximize unsupported)")

    def _slot(self, p):
        s = self.state.get(id(p))
        if s is None or s[0].shape != p.shape:
            s = [torch.zeros_like(p, memory_format=torch.contiguous_format),
                 torch.zeros_like(p, memory_format=torch.contiguous_format), 0]
            self.state[id(p)] = s
        return s

    @torch.no_grad()
    def step(self, grads=None):
        """grads: optional dict id(param) -> gradient tensor overriding param.grad (e.g. views of a reduced bucket)."""
        L = _lib.lib()
        by_hyper = {}
        for o in self.optimizers:
            for g in o.param_groups:
                key = (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
                for p in g["params"]:
                    gr = grads.get(id(p)) if grads is not None else p.grad
                    if gr is None:
                        continue
                    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                        raise ValueError("MultiAdam: parameters must be contiguous fp32 device tensors")
                    if gr.dtype != torch.float32 or gr.shape != p.shape:
                        raise ValueError("MultiAdam: gradient / parameter mismatch")
                    gr = gr if gr.is_contiguous() else gr.contiguous()
                    s = self._slot(p)
                    s[2] += 1
                    by_hyper.setdefault(key, []).append((p, gr, s, float(g["lr"])))
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for (b1, b2, eps), items in by_hyper.items():
            n = len(items)
            VP = ctypes.c_void_p * n
            rc = L.dgm_adam_step(
                n, VP(*[it[0].data_ptr() for it in items]), VP(*[it[1].data_ptr() for it in items]),
                VP(*[it[2][0].data_ptr() for it in items]), VP(*[it[2][1].data_ptr() for it in items]),
                (ctypes.c_longlong * n)(*[it[0].numel() for it in items]), (ctypes.c_float * n)(*[it[3] for it in items]),
                (ctypes.c_int * n)(*[it[2][2] for it in items]), b1, b2, eps, stream)
            if rc != 0:
                raise RuntimeError(L.dgm_last_error().decode())
