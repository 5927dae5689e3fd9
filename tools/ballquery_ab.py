"""Timings of the radius search (csrc/kpindex.hip) on the point sets of the KPConv input pyramid of B synthetic plots: count
pass and fill pass of the self-search of a level and of the pooling search into the next level.

    python tools/ballquery_ab.py --plots 32 --points 16000 --reps 20
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plots", type=int, default=32)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import dpcr_agb_amd  # noqa: F401
    from dpcr_agb_amd import kp_index, synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import KPConvModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016))
    opt = Opt(MODEL_OPTIONS["KPConv"])
    model = KPConvModel(opt, "kpconv", ds).to(dev)
    b = synthetic.make_point_batch(list(range(args.plots)), n_points=args.points)
    lens = np.bincount(b.batch.numpy()).astype(np.int64)
    inp = model.prepare_inputs(b.pos, b.x, lens, dev)
    cfg = opt.config
    cases = []
    for lvl in range(2):
        r = cfg.first_subsampling_dl * cfg.conv_radius * 2 ** lvl
        pts, ln = inp["points"][lvl], inp["lengths"][lvl].numpy()
        cases.append((f"level {lvl} self-search", pts, pts, ln, ln, r))
        nxt, ln2 = inp["points"][lvl + 1], inp["lengths"][lvl + 1].numpy()
        cases.append((f"level {lvl} -> {lvl + 1} pooling search", nxt, pts, ln2, ln, r))
    for name, q, s, ql, sl, r in cases:
        bounds = kp_index.support_bounds(s, sl)
        tb, tf = [], []
        for rep in range(args.reps + 3):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            job = kp_index.neighbors_begin(q, s, ql, sl, r, bounds)
            e[1].record()
            width, total = kp_index.read_back(job.max_count, job.row_ptr[-1:])
            e1b = torch.cuda.Event(enable_timing=True)
            e1b.record()
            nb = kp_index.neighbors_finish_csr(job, width[0], total[0])
            e[2].record()
            torch.cuda.synchronize()
            if rep >= 3:
                tb.append(e[0].elapsed_time(e[1]))
                tf.append(e1b.elapsed_time(e[2]))
        print(f"[ballquery_ab] {name}: {q.shape[0]} queries, {s.shape[0]} supports, {int(total[0])} neighbours "
              f"({int(total[0]) / q.shape[0]:.1f} per query, max {int(width[0])}): grid + count pass {np.median(tb) * 1e3:.0f} us, "
              f"offsets + fill pass {np.median(tf) * 1e3:.0f} us", flush=True)


if __name__ == "__main__":
    main()
