"""AdaBelief with decoupled weight decay and RAdam rectification — the optimiser of every NFI run
(torch-points3d/conf/training/nfi/minkowski.yaml:17-20), same update rule and state layout
(``step``, ``exp_avg``, ``exp_avg_var``) as torch_points3d/core/optimizer/adabelief.py:89-201, including
its in-place ``exp_avg_var.add_(eps)`` (the eps accumulates in the second-moment state every step).

Implemented with multi-tensor (``torch._foreach``) updates: a handful of launches per step instead of
~12 per parameter.
"""
import math

import torch
from torch.optim.optimizer import Optimizer


class AdaBelief(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-16, weight_decay=0, amsgrad=False,
                 decoupled_decay=True, fixed_decay=False, rectify=True, degenerated_to_sgd=True, fused=False,
                 clip_value=None):
        """fused=True: one HIP launch per parameter group (csrc/optim.hip) instead of ~12 multi-tensor launches;
        clip_value: clip_grad_value_ folded into that launch (the gradients themselves are left untouched)."""
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid betas: {betas}")
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the NFI recipes and is not implemented")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                        degenerated_to_sgd=degenerated_to_sgd, decoupled_decay=decoupled_decay, rectify=rectify,
                        fixed_decay=fixed_decay)
        super().__init__(params, defaults)
        self.fused, self.clip_value = bool(fused), clip_value
        self._fused_cache = {}

    @staticmethod
    def _rectified_step(step, beta1, beta2, degenerated_to_sgd):
        beta2_t = beta2 ** step
        num_sma_max = 2 / (1 - beta2) - 1
        num_sma = num_sma_max - 2 * step * beta2_t / (1 - beta2_t)
        if num_sma >= 5:
            step_size = math.sqrt((1 - beta2_t) * (num_sma - 4) / (num_sma_max - 4) * (num_sma - 2) / num_sma *
                                  num_sma_max / (num_sma_max - 2)) / (1 - beta1 ** step)
        elif degenerated_to_sgd:
            step_size = 1.0 / (1 - beta1 ** step)
        else:
            step_size = -1
        return num_sma, step_size

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params, grads, m, v = [], [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("AdaBelief does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_var"] = torch.zeros_like(p)
                st["step"] += 1
                params.append(p)
                grads.append(p.grad)
                m.append(st["exp_avg"])
                v.append(st["exp_avg_var"])
            if not params:
                continue
            steps = {self.state[p]["step"] for p in params}
            if len(steps) != 1:
                raise RuntimeError("parameters of one group must share the step count")
            step = steps.pop()
            beta1, beta2 = group["betas"]
            lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
            if self.fused and params[0].is_cuda and group["decoupled_decay"]:
                self._fused_step(group, params, grads, m, v, step)
                continue
            if self.fused and self.clip_value:
                # a group that cannot take the fused launch is clipped here (the caller skips clip_grad_value_ for
                # fused optimisers); out of place, like the fused kernel leaves the gradients untouched
                grads = [g.clamp(-float(self.clip_value), float(self.clip_value)) for g in grads]

            if group["decoupled_decay"]:
                torch._foreach_mul_(params, 1.0 - (wd if group["fixed_decay"] else lr * wd))
            elif wd != 0:
                grads = torch._foreach_add(grads, params, alpha=wd)

            # m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) (g-m)^2 ; v += eps (in place, as the reference does)
            torch._foreach_mul_(m, beta1)
            torch._foreach_add_(m, grads, alpha=1 - beta1)
            resid = torch._foreach_sub(grads, m)
            torch._foreach_mul_(v, beta2)
            torch._foreach_addcmul_(v, resid, resid, value=1 - beta2)
            torch._foreach_add_(v, eps)

            if not group["rectify"]:
                bc1 = 1 - beta1 ** step
                bc2 = 1 - beta2 ** step
                denom = torch._foreach_sqrt(v)
                torch._foreach_div_(denom, math.sqrt(bc2))
                torch._foreach_add_(denom, eps)
                torch._foreach_addcdiv_(params, m, denom, value=-lr / bc1)
            else:
                num_sma, step_size = self._rectified_step(step, beta1, beta2, group["degenerated_to_sgd"])
                if num_sma >= 5:
                    denom = torch._foreach_sqrt(v)
                    torch._foreach_add_(denom, eps)
                    torch._foreach_addcdiv_(params, m, denom, value=-step_size * lr)
                elif step_size > 0:
                    torch._foreach_add_(params, m, alpha=-step_size * lr)
        return loss

    # ------------------------------------------------------------------ fused HIP path
    def _fused_step(self, group, params, grads, m, v, step):
        from . import _lib
        _V, _I, _F = _lib.c_void_p, _lib.c_int, _lib.c_float
        if "agb_adabelief_step" not in _lib._SIGNATURES:
            _lib.declare("agb_adabelief_chunk", [])
            _lib.declare("agb_adabelief_step", [_V, _V, _V, _I, _F, _F, _F, _F, _F, _F, _F, _F, _I, _F, _V])
        dev = params[0].device
        key = (id(group), tuple(p.numel() for p in params))
        cache = self._fused_cache.get(key)
        if cache is None:
            chunk = _lib.load().agb_adabelief_chunk()
            ct, ci = [], []
            for t, p in enumerate(params):
                nck = (p.numel() + chunk - 1) // chunk
                ct += [t] * nck
                ci += list(range(nck))
            cache = dict(ct=torch.tensor(ct, dtype=torch.int32, device=dev),
                         ci=torch.tensor(ci, dtype=torch.int32, device=dev), n=len(ct),
                         # ring of pinned staging tables: the host may run a step ahead of the device
                         # ring of pinned staging tables: the host may run ahead of the device; a slot is rewritten
                         # only after the device has executed the copy that read it (event per slot)
                         host=[torch.empty(len(params), 5, dtype=torch.int64).pin_memory() for _ in range(4)],
                         dev=[torch.empty(len(params), 5, dtype=torch.int64, device=dev) for _ in range(4)],
                         done=[None] * 4, it=0)
            self._fused_cache[key] = cache
        slot = cache["it"] % 4
        cache["it"] += 1
        table = []
        for p, g, mm, vv in zip(params, grads, m, v):
            if not (p.is_contiguous() and g.is_contiguous() and mm.is_contiguous() and vv.is_contiguous()):
                raise RuntimeError("fused AdaBelief needs contiguous parameters, gradients and state")
            if not (p.dtype == g.dtype == mm.dtype == vv.dtype == torch.float32):
                raise RuntimeError("fused AdaBelief needs fp32 parameters, gradients and state")
            table.append((p.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), p.numel()))
        if cache["done"][slot] is not None:
            cache["done"][slot].synchronize()   # the copy that last read this pinned slot has run
        cache["host"][slot].numpy()[:] = table
        descs = cache["dev"][slot]
        descs.copy_(cache["host"][slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        cache["done"][slot] = ev
        beta1, beta2 = group["betas"]
        lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
        decay = 1.0 - (wd if group["fixed_decay"] else lr * wd)
        inv_sqrt_bc2 = 1.0
        if not group["rectify"]:
            mode, stepv = 3, lr / (1 - beta1 ** step)
            inv_sqrt_bc2 = 1.0 / math.sqrt(1 - beta2 ** step)
        else:
            num_sma, step_size = self._rectified_step(step, beta1, beta2, group["degenerated_to_sgd"])
            if num_sma >= 5:
                mode, stepv = 0, step_size * lr
            elif step_size > 0:
                mode, stepv = 1, step_size * lr
            else:
                mode, stepv = 2, 0.0
        clip = float(self.clip_value) if self.clip_value else 0.0
        _lib.call("agb_adabelief_step", _lib.ptr(descs), _lib.ptr(cache["ct"]), _lib.ptr(cache["ci"]),
                  cache["n"], decay, beta1, beta2, 1 - beta1, 1 - beta2, eps, stepv, inv_sqrt_bc2, mode, clip,
                  _lib.stream())
        # (the parameters were rewritten through raw pointers: torch's version counters did not move)
        from .sparse_ops import bump_weight_epoch
        bump_weight_epoch()
