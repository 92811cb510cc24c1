"""torch.optim.Adam semantics over many optimizers / parameter groups in ONE kernel launch (dgm_adam_step).

The reference ends every iteration with `gaussians.optimizer.step(); deform.optimizer.step(); ...`
(R/train.py:518-524), each a torch.optim.Adam(lr=0.0, eps=1e-15) whose group learning rates are rewritten every
iteration by `update_learning_rate`.  MultiAdam keeps those optimizer objects as the source of truth for hyper
parameters (it reads `param_groups[*]["lr"]`, betas, eps at every step, so the reference's lr schedulers keep working),
keeps the moments in those optimizers' own `state` (torch layout) and applies the update to every tensor of every group with a single HIP kernel.
No CPU / PyTorch fallback: it needs the HIP library.
"""
import ctypes

import torch

from . import _lib


class MultiAdam:
    def __init__(self, optimizers):
        self.optimizers = list(optimizers)
        for o in self.optimizers:
            for g in o.param_groups:
                if g.get("amsgrad") or g.get("weight_decay", 0) != 0 or g.get("maximize"):
                    raise ValueError("MultiAdam implements plain Adam only (amsgrad / weight_decay / maximize unsupported)")
            if hasattr(o, "register_load_state_dict_post_hook"):  # a loaded state replaces the moment tensors: plan again
                o.register_load_state_dict_post_hook(lambda *_a, **_k: setattr(self, "_sig", None))

    @staticmethod
    def _slot(opt, p):
        """Moments live in the OWNING optimizer's state[p] with torch.optim.Adam's own keys ('step', 'exp_avg',
        'exp_avg_sq'), so the reference's optimizer surgery (_prune_optimizer / cat_tensors_to_optimizer /
        replace_tensor_to_optimizer, R/scene/gaussian_model_dpsr_dynamic_anchor.py:364-440) and state_dict() see and
        edit the real state; entries die with their parameter (no id() keyed side table)."""
        s = opt.state[p]  # defaultdict: creates {} for a new / replaced Parameter
        if "exp_avg" not in s:
            s["step"] = torch.tensor(0.0)
            s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        elif s["exp_avg"].shape != p.shape or s["exp_avg_sq"].shape != p.shape:
            raise ValueError("MultiAdam: optimizer state does not match its parameter's shape (surgery left it stale)")
        return s

    def _plan(self):
        """Everything that only changes when the parameter set does is gathered once per parameter set and reused: the grouping
        by hyper-parameters, the state slots, the ctypes argument arrays with the pointers of parameters and moments and the
        sizes already filled in.  Per call only gradients, learning rates and step counts are written."""
        sig = tuple((id(p), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
                    for o in self.optimizers for g in o.param_groups for p in g["params"])  # (hyper-parameters too: a changed beta / eps regroups)
        if getattr(self, "_sig", None) == sig:
            return self._cached
        by_hyper = {}
        for o in self.optimizers:
            for g in o.param_groups:
                key = (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
                for p in g["params"]:
                    by_hyper.setdefault(key, []).append((o, g, p))
        plans = []
        for key, entries in by_hyper.items():
            n = len(entries)
            VP = ctypes.c_void_p * n
            plans.append(dict(key=key, entries=entries, n=n, slots=[None] * n, mom=[None] * n, P=VP(), G=VP(), M=VP(), V=VP(),
                              N=(ctypes.c_longlong * n)(), LR=(ctypes.c_float * n)(), ST=(ctypes.c_int * n)(), pos=[-1] * n,
                              steps=torch.zeros(n, dtype=torch.float32), cnt=[0] * n, fast_left=0,
                              params=[e[2] for e in entries], shapes=[e[2].shape for e in entries], groups=self._group_runs(entries)))
        self._sig, self._cached = sig, plans
        return plans

    @staticmethod
    def _group_runs(entries):
        """[(first, last + 1, param_group)]: runs of consecutive entries that share a param_group (one learning rate)."""
        runs, lo = [], 0
        for i in range(1, len(entries) + 1):
            if i == len(entries) or entries[i][1] is not entries[lo][1]:
                runs.append((lo, i, entries[lo][1]))
                lo = i
        return runs

    def _fast_step(self, pl, grads):
        """The call that repeats the previous one -- the same entries have a gradient (and the others still have none), each where
        it sat -- only rewrites the gradient pointers, one learning rate per run of a param_group and the common step count:
        ~0.6 us per tensor instead of ~5.  Anything unusual returns False and the general path below does the work (and re-validates
        the bound state every 64 calls).  Known limit: an IN-PLACE edit of a step tensor (`state[p]["step"].fill_(k)`) keeps the
        tensor's identity, so this path goes on counting from its own `cnt` and the edit takes effect at the next re-validation --
        within 64 calls; replacing the tensor (what checkpoint loading and the densification surgery do) is seen at once."""
        f32, ptrs = torch.float32, []
        params, shapes = pl["params"], pl["shapes"]
        # the bound state must still be the live one: in-place surgery on an unchanged Parameter (state[p] replaced, a moment or
        # the step tensor swapped, p.data re-pointed) sends the call down the general path, which binds again
        entries, slots, mom, Pp = pl["entries"], pl["slots"], pl["mom"], pl["P"]
        for k_, i in enumerate(pl["act"]):
            sl, m = slots[i], mom[i]
            if entries[i][0].state.get(params[i]) is not sl or sl.get("exp_avg") is not m[0] or sl.get("exp_avg_sq") is not m[1] \
                    or sl.get("step") is not m[2] or params[i].data_ptr() != Pp[k_]:
                return False
        try:
            if grads is None:
                for i in pl["act"]:
                    gr = params[i].grad
                    if gr.dtype is not f32 or gr.shape != shapes[i] or not gr.is_contiguous():
                        return False
                    ptrs.append(gr.data_ptr())
                for i in pl["inact"]:
                    if params[i].grad is not None:
                        return False
            else:
                for i in pl["act"]:
                    gr = grads[id(params[i])]
                    if gr.dtype is not f32 or gr.shape != shapes[i] or not gr.is_contiguous():
                        return False
                    ptrs.append(gr.data_ptr())
                for i in pl["inact"]:
                    if id(params[i]) in grads:
                        return False
        except (AttributeError, KeyError):  # an entry of the active set without a gradient this step
            return False
        k, cnt = len(ptrs), pl["cnt"]
        c = cnt[pl["act"][0]] + 1
        pl["G"][0:k] = ptrs
        LR = pl["LR"]
        for lo, hi, g in pl["act_runs"]:
            LR[lo:hi] = [g["lr"]] * (hi - lo)
        pl["ST"][0:k] = [c] * k
        for i in pl["act"]:
            cnt[i] = c
        if k == pl["n"]:
            pl["steps"].add_(1.0)
        else:
            pl["steps"][pl["act"]] += 1.0
        pl["fast_left"] -= 1
        pl["k"] = k
        return True

    def _bind(self, pl, i):
        """(Re)binds entry i of a plan to its parameter's current state tensors; returns the state slot.  The step counter of the
        entry becomes a 0-dim VIEW into the plan's flat host tensor `steps` (torch.optim.Adam's convention -- a host scalar tensor per
        parameter, visible to state_dict() and optimizer surgery -- with ONE add per call instead of a foreach over 45 tensors),
        and is mirrored as a Python int in `cnt` (the kernel's argument: no tensor -> int conversion per entry and call)."""
        o, g, p = pl["entries"][i]
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
            raise ValueError("MultiAdam: parameters must be contiguous fp32 device tensors")
        s = self._slot(o, p)
        if not (s["exp_avg"].is_contiguous() and s["exp_avg_sq"].is_contiguous()):
            s["exp_avg"], s["exp_avg_sq"] = s["exp_avg"].contiguous(), s["exp_avg_sq"].contiguous()
        count = int(float(s["step"]))
        view = pl["steps"][i]
        view.fill_(float(count))
        s["step"] = view
        pl["cnt"][i] = count
        pl["slots"][i], pl["mom"][i] = s, (s["exp_avg"], s["exp_avg_sq"], view)
        return s

    @torch.no_grad()
    def step(self, grads=None):
        """grads: optional dict id(param) -> gradient tensor overriding param.grad (e.g. views of a reduced bucket)."""
        L = _lib.lib()
        stream = None
        for pl in self._plan():
            if pl["fast_left"] > 0 and self._fast_step(pl, grads):
                if stream is None:
                    stream = _lib.stream_ptr()
                b1, b2, eps = pl["key"]
                rc = L.dgm_adam_step(pl["k"], pl["P"], pl["G"], pl["M"], pl["V"], pl["N"], pl["LR"], pl["ST"], b1, b2, eps, stream)
                if rc != 0:
                    raise RuntimeError(L.dgm_last_error().decode())
                continue
            entries, slots, mom, cnt = pl["entries"], pl["slots"], pl["mom"], pl["cnt"]
            P, G, M, V, N, LR, ST = pl["P"], pl["G"], pl["M"], pl["V"], pl["N"], pl["LR"], pl["ST"]
            k = 0
            keep = []    # gradient tensors made contiguous for this call
            active = []  # entries that take part in this call
            pos = pl["pos"]  # array position each entry had in the previous call (-1: took no part): the cached pointers at a
            host_steps = pl["steps"].tolist()  # (the step tensors are views into this host tensor: an in-place edit of one shows here)
            for i, (o, g, p) in enumerate(entries):  # position are valid as long as the same entry lands there again
                gr = grads.get(id(p)) if grads is not None else p.grad
                if gr is None:
                    pos[i] = -1
                    continue
                if gr.dtype != torch.float32 or gr.shape != p.shape:
                    raise ValueError("MultiAdam: gradient / parameter mismatch")
                if not gr.is_contiguous():
                    gr = gr.contiguous()
                    keep.append(gr)
                s = slots[i]
                # the state dict entry, its moment tensors and its step tensor must still be the ones bound last time
                # (optimizer surgery on an unchanged Parameter, load_state_dict ...): otherwise bind again
                if s is None or pos[i] != k or o.state.get(p) is not s or s.get("exp_avg") is not mom[i][0] \
                        or s.get("exp_avg_sq") is not mom[i][1] or s.get("step") is not mom[i][2] or p.data_ptr() != P[k]:
                    s = self._bind(pl, i)
                    P[k], M[k], V[k], N[k] = p.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), p.numel()
                    pos[i] = k
                else:
                    cnt[i] = int(host_steps[i])
                G[k], LR[k] = gr.data_ptr(), g["lr"]
                cnt[i] += 1
                ST[k] = cnt[i]
                active.append(i)
                k += 1
            if k == 0:
                continue
            if k == len(entries):
                pl["steps"].add_(1.0)
            else:
                pl["steps"][active] += 1.0
            # the next calls may take the fast path while the same entries take part and their counts are equal
            aset = set(active)
            pl["act"], pl["inact"] = active, [i for i in range(len(entries)) if i not in aset]
            pl["act_runs"] = self._group_runs([entries[i] for i in active])
            pl["fast_left"] = 64 if len({cnt[i] for i in active}) == 1 else 0
            if stream is None:
                stream = _lib.stream_ptr()
            b1, b2, eps = pl["key"]
            rc = L.dgm_adam_step(k, P, G, M, V, N, LR, ST, b1, b2, eps, stream)
            if rc != 0:
                raise RuntimeError(L.dgm_last_error().decode())
