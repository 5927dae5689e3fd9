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
                 decoupled_decay=True, fixed_decay=False, rectify=True, degenerated_to_sgd=True):
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
