"""Generates tests/golden/kpconv_layer_golden.npz by IMPORTING the reference's Python (torch_points3d.modules.KPConv.
{blocks,architectures,kernel_points}) from /root/reference in this container, and its C++ index code through
oracle/_ref.  The .npz holds inputs, parameters and the reference's outputs/gradients only.
Run:  python tests/golden/make_kpconv_layer_golden.py   (takes ~1 min: the reference optimises kernel points)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/torch-points3d")
os.chdir("/tmp")  # the reference creates ./kernels/dispositions on import-time use

from torch_points3d.modules.KPConv import architectures, blocks, kernel_points  # noqa: E402
from oracle import kpconv_index as K  # noqa: E402
from dpcr_agb_amd import synthetic  # noqa: E402


def t2n(t):
    return t.detach().cpu().numpy()


def main():
    out = {}
    torch.manual_seed(0)
    np.random.seed(0)
    # 1. one kernel-point disposition exactly as the reference draws it (kernel_points.py:338-413)
    np.random.seed(0)
    out["kp_seed0_radius1"] = kernel_points.load_kernels(1.0, 15, dimension=3, fixed="center")

    # 2. geometry: two ragged clouds, self neighbours + one pooled level
    sizes = [260, 140]
    pts = np.concatenate([synthetic.make_plot(300 + i, n_points=n)[0] for i, n in enumerate(sizes)])
    lens = np.array(sizes, dtype=np.int32)
    r0, dl = 0.08, 0.064
    nbr0 = K.ref_batch_neighbors(pts, pts, lens, lens, r0)
    pool_p, pool_b = K.batch_grid_subsampling(pts, lens, sampleDl=dl, order="canonical")
    pool_i = K.batch_neighbors(pool_p, pts, pool_b, lens, r0)
    nbr1 = K.batch_neighbors(pool_p, pool_p, pool_b, pool_b, 2 * r0)
    out.update(points0=pts, lens0=lens, neighbors0=nbr0, points1=pool_p, lens1=pool_b, pools0=pool_i, neighbors1=nbr1)

    # 3. a single KPConv layer, forward + gradients (blocks.py:264-400)
    cin, cout, ext = 8, 12, 0.04
    conv = blocks.KPConv(15, 3, cin, cout, ext, r0)
    x = torch.randn(len(pts), cin, requires_grad=True)
    q, s, idx = torch.from_numpy(pts), torch.from_numpy(pts), torch.from_numpy(nbr0.astype(np.int64))
    y = conv(q, s, idx, x)
    g = torch.randn_like(y)
    y.backward(g)
    out.update(L_kp=t2n(conv.kernel_points), L_w=t2n(conv.weights), L_x=t2n(x), L_y=t2n(y), L_g=t2n(g),
               L_dx=t2n(x.grad), L_dw=t2n(conv.weights.grad), L_ext=np.float32(ext))
    # strided use: queries = pooled points
    conv.zero_grad()
    x2 = torch.randn(len(pts), cin, requires_grad=True)
    y2 = conv(torch.from_numpy(pool_p), s, torch.from_numpy(pool_i.astype(np.int64)), x2)
    g2 = torch.randn_like(y2)
    y2.backward(g2)
    out.update(S_x=t2n(x2), S_y=t2n(y2), S_g=t2n(g2), S_dx=t2n(x2.grad), S_dw=t2n(conv.weights.grad))

    # 4. max_pool / global_sum helpers (blocks.py:98-114,141-160)
    xm = torch.randn(len(pts), 6) - 0.5
    out["P_x"] = t2n(xm)
    out["P_maxpool"] = t2n(blocks.max_pool(xm, torch.from_numpy(pool_i.astype(np.int64))))
    out["P_globalsum"] = t2n(blocks.global_sum(xm, torch.from_numpy(lens.astype(np.int64))))

    # 5. a small KPCNN end to end (architectures.py:67-151) with the reference's own initialisation
    cfg = types.SimpleNamespace(
        first_subsampling_dl=0.032, conv_radius=2.5, in_features_dim=3, first_features_dim=16, activation="relu",
        num_kernel_points=15, architecture=["simple", "resnetb", "resnetb_strided", "resnetb", "global_sum"],
        use_batch_norm=True, batch_norm_momentum=0.02, KP_extent=1.0, KP_influence="linear", aggregation_mode="sum",
        fixed_kernel_points="center", modulated=False, in_points_dim=3, deform_fitting_mode="point2point",
        deform_fitting_power=1.0, deform_lr_factor=0.1, repulse_extent=1.2)
    torch.manual_seed(1)
    np.random.seed(1)
    net = architectures.KPCNN(cfg)
    net.train()
    feats = torch.from_numpy(np.concatenate([synthetic.make_plot(300 + i, n_points=n)[1]
                                             for i, n in enumerate(sizes)]))
    batch = types.SimpleNamespace(
        features=feats,
        points=[torch.from_numpy(pts), torch.from_numpy(pool_p)],
        neighbors=[torch.from_numpy(nbr0.astype(np.int64)), torch.from_numpy(nbr1.astype(np.int64))],
        pools=[torch.from_numpy(pool_i.astype(np.int64)), torch.zeros(0, 1, dtype=torch.int64)],
        lengths=[torch.from_numpy(lens.astype(np.int64)), torch.from_numpy(pool_b.astype(np.int64))])
    sd0 = {k: t2n(v).copy() for k, v in net.state_dict().items()}
    yn = net(batch)
    gn = torch.randn_like(yn)
    yn.backward(gn)
    out["N_feats"], out["N_y"], out["N_g"] = t2n(feats), t2n(yn), t2n(gn)
    for k, v in sd0.items():
        out["N_sd/" + k] = v
    for k, p in net.named_parameters():
        if p.grad is not None:
            out["N_grad/" + k] = t2n(p.grad)
    for k, v in net.state_dict().items():
        if "running" in k:
            out["N_after/" + k] = t2n(v)
    path = os.path.join(ROOT, "tests", "golden", "kpconv_layer_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
    print("neighbors0", nbr0.shape, "pools0", pool_i.shape, "neighbors1", nbr1.shape, "KPCNN out", tuple(yn.shape))


if __name__ == "__main__":
    main()
