"""How much do the neighbour lists of CONSECUTIVE query rows overlap, per level of the KPConv input pyramid?
(The backward gather scatters one 64-byte atomic per (row, neighbour, 16 channels); rows that share support points could
be combined in LDS first.)  Prints, per level and group size R: valid entries / distinct support rows per group.
With --sort the level-0 points are first ordered by grid cell (what a spatial pre-sort of the raw plot would give).
Usage (GPU box): python tools/kp_overlap.py [--points 16000] [--sort]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def group_stats(nb, ns, R):
    n, h = nb.shape
    g = n // R
    if g == 0:
        return None
    ids = nb[:g * R].view(g, R * h).to(torch.int64)
    ids, _ = ids.sort(dim=1)
    valid = ids < ns
    new = torch.ones_like(valid)
    new[:, 1:] = ids[:, 1:] != ids[:, :-1]
    tot = valid.sum().item()
    uniq = (valid & new).sum().item()
    return tot / g, uniq / g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--sort", action="store_true")
    a = ap.parse_args()
    import dpcr_agb_amd
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import KPConvModel
    dpcr_agb_amd.limit_host_threads()
    dev = torch.device("cuda:0")
    np.random.seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_032))
    model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds)
    b = synthetic.make_point_batch(list(range(a.batch)), n_points=a.points)
    pos, x = b.pos.view(-1, 3), b.x.view(-1, b.x.shape[-1])
    lens = (b.ptr[1:] - b.ptr[:-1]).numpy().astype(np.int64)
    if a.sort:
        cfg = model.config
        cell = cfg.first_subsampling_dl * cfg.conv_radius
        bid = torch.repeat_interleave(torch.arange(len(lens)), torch.from_numpy(lens))
        c = torch.floor((pos - pos.min(0).values) / cell).to(torch.int64)
        key = ((bid * 4096 + c[:, 0]) * 4096 + c[:, 1]) * 4096 + c[:, 2]
        order = torch.argsort(key)
        pos, x = pos[order], x[order]
    pyr = model.prepare_inputs(pos, x, lens, dev)
    for lvl, (pts, nb, pl) in enumerate(zip(pyr["points"], pyr["neighbors"], pyr["pools"])):
        for name, m, ns in (("conv", nb, len(pts)), ("pool", pl, len(pts))):
            m = m.padded() if hasattr(m, "padded") else m
            if m is None or m.numel() == 0 or m.shape[0] == 0:
                continue
            line = f"level {lvl} {name}: rows {m.shape[0]:7d} width {m.shape[1]:4d}"
            for R in (4, 16, 32, 64):
                st = group_stats(m, ns, R)
                if st:
                    line += f" | R={R}: {st[0]:7.1f} -> {st[1]:6.1f} ({st[0] / max(st[1], 1):.2f}x)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
