"""Generates tests/golden/adabelief_golden.npz by importing the REFERENCE's optimiser
(/root/reference/torch-points3d/torch_points3d/core/optimizer/adabelief.py) in this container: a 12-step trajectory of
three parameter tensors under seeded gradients with the NFI recipe (lr 0.005, wd 1e-2, eps 1e-16, rectify) and
clip_grad_value_(100) before every step (models/base_model.py:241-245).  Data only."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location(
    "ref_adabelief", "/root/reference/torch-points3d/torch_points3d/core/optimizer/adabelief.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)


def main():
    torch.manual_seed(0)
    shapes = [(27, 8, 16), (16,), (5, 3)]
    params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    opt = mod.AdaBelief(params, lr=0.005, weight_decay=1e-2)
    out = {f"p0_{i}": p.detach().numpy().copy() for i, p in enumerate(params)}
    g = torch.Generator().manual_seed(1)
    for step in range(12):
        for i, p in enumerate(params):
            scale = 300.0 if step == 3 else 1.0      # one step exceeds the clip value
            p.grad = torch.randn(*shapes[i], generator=g) * scale
            out[f"g{step}_{i}"] = p.grad.numpy().copy()
        torch.nn.utils.clip_grad_value_(params, 100)
        opt.step()
        for i, p in enumerate(params):
            out[f"p{step + 1}_{i}"] = p.detach().numpy().copy()
    for i, p in enumerate(params):
        out[f"m_{i}"] = opt.state[p]["exp_avg"].numpy().copy()
        out[f"v_{i}"] = opt.state[p]["exp_avg_var"].numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", "adabelief_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
