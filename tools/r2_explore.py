"""Exploration for the R2 acceptance set (GPU only, seconds per variant): which synthetic label / schedule gives a
POSITIVE, PLATEAUED validation R2 with a small trial-to-trial spread, so that the reference's "median of 5 trials"
protocol (README.md:24-56) can be read at +-0.005.  Prints one JSON line per (variant, precision).
Usage: python tools/r2_explore.py [variant ...]"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dpcr_agb_amd  # noqa: E402

dpcr_agb_amd.limit_host_threads()


def relabel(batch, mode):
    """Deterministic labels computed from the voxelised plot itself (what the encoder sees)."""
    if mode == "trees":
        return batch
    h = batch.pos[:, 2].double() * 40.0        # metres
    B = len(batch)
    idx = batch.batch
    if mode == "voxsum":
        y0 = torch.zeros(B, dtype=torch.float64).index_add_(0, idx, torch.where(h > 1.0, h, 0 * h) ** 1.5) * 0.05
        y1 = torch.zeros(B, dtype=torch.float64).index_add_(0, idx, h ** 1.2) * 0.2
    elif mode == "count":
        cnt = torch.bincount(idx, minlength=B).double()
        y0 = 0.5 * cnt
        y1 = torch.zeros(B, dtype=torch.float64).index_add_(0, idx, h) * 0.1
    elif mode == "voxmean":
        cnt = torch.bincount(idx, minlength=B).double()
        can = (h > 2.0).double()
        ncan = torch.zeros(B, dtype=torch.float64).index_add_(0, idx, can)
        mh = torch.zeros(B, dtype=torch.float64).index_add_(0, idx, h * can) / ncan.clamp(min=1)
        y0 = 3.0 * mh ** 1.6 * (ncan / cnt)
        y1 = 1.87 * y0 * (1 + 0.02 * (mh - 15) / 15)
    else:
        raise ValueError(mode)
    batch.y_reg = torch.stack([y0, y1], 1).float()
    return batch


def run_variant(v, dev, precision="fp32"):
    from dpcr_agb_amd import sparse_ops, synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    from dpcr_agb_amd.metrics import RegressionMeter
    bs = v["batch"]

    def one_batch(seeds):
        if v.get("density") is None:
            return synthetic.make_sparse_batch(seeds, n_points=v["points"])
        # returns per plot follow the stand: n_points = base + per_tree * n_trees (the generator's first draw)
        base, per_tree = v["density"]
        parts = []
        for sd in seeds:
            n_trees = int(np.random.default_rng(sd).integers(15, 61))
            parts.append(synthetic.make_sparse_batch([sd], n_points=base + per_tree * n_trees))
        cat = lambda name: torch.cat([getattr(p, name) for p in parts])  # noqa: E731
        batch = torch.cat([torch.full((len(p.batch),), i, dtype=torch.int64) for i, p in enumerate(parts)])
        return synthetic.PlotBatch(batch, cat("coords"), cat("x"), cat("pos"), cat("y_reg"), cat("y_reg_mask"), len(parts))

    mk = lambda lo, n: [relabel(one_batch(list(range(lo + i, lo + i + bs))), v["label"]) for i in range(0, n, bs)]  # noqa: E731
    t0 = time.time()
    train_h, val_h = mk(0, v["train"]), mk(500_000, v["val"])
    ys = torch.cat([b.y_reg for b in train_h]).double()
    train, val = [b.to(dev) for b in train_h], [b.to(dev) for b in val_h]
    val_mean = torch.cat([b.y_reg for b in val_h]).double().mean(0)
    t_data = time.time() - t0
    old = sparse_ops.set_conv_precision(precision)
    trials = []
    try:
        for t in range(v["trials"]):
            ds = synthetic.SyntheticDataset(stat_seeds=range(0, 8))
            ds._stats = {"mean": ys.mean(0).numpy(), "std": ys.std(0).numpy(), "min": ys.min(0).values.numpy(),
                         "max": ys.max(0).values.numpy()}
            torch.manual_seed(t)
            model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds).to(dev)
            model.init_train_objects(TRAINING_NFI)
            random.seed(1234 + t)
            nb = len(train)
            path = []
            for epoch in range(v["epochs"]):
                model.train()
                for i in np.random.default_rng(1000 * t + epoch).permutation(nb):
                    model.set_input(train[i], dev)
                    model.optimize_parameters(epoch, bs, nb)
                if (epoch + 1) % max(1, v["epochs"] // 6) == 0 or epoch + 1 == v["epochs"]:
                    m = model.evaluate(val, dev, val_mean)
                    path.append([epoch + 1] + [round(x, 4) for x in m["r2"]])
            model.calibrate_bn(train, dev, epochs=v.get("calibrate", 2))
            m = model.evaluate(val, dev, val_mean)
            trials.append(dict(r2=[round(x, 5) for x in m["r2"]], rmse=[round(x, 2) for x in m["rmse"]],
                               loss=round(float(model.loss.detach()), 5), path=path))
    finally:
        sparse_ops.set_conv_precision(old)
    r2 = np.array([t["r2"] for t in trials])
    return dict(variant=v, precision=precision, median=np.median(r2, 0).round(5).tolist(),
                spread=(r2.max(0) - r2.min(0)).round(5).tolist(), std=r2.std(0).round(5).tolist(), trials=trials,
                target_std=ys.std(0).numpy().round(2).tolist(), data_s=round(t_data, 1), total_s=round(time.time() - t0, 1))


VARIANTS = {
    "trees_1k_b16_e70": dict(label="trees", points=1000, train=256, val=128, batch=16, epochs=70, trials=5),
    "trees_dens_b16_e70": dict(label="trees", points=1000, density=(400, 20), train=256, val=128, batch=16, epochs=70, trials=5),
    "trees_dens_b16_e30": dict(label="trees", points=1000, density=(400, 20), train=256, val=128, batch=16, epochs=30, trials=5),
    "count_dens_b16_e70": dict(label="count", points=1000, density=(400, 20), train=256, val=128, batch=16, epochs=70, trials=5),
    "count_dens_b16_e30": dict(label="count", points=1000, density=(400, 20), train=256, val=128, batch=16, epochs=30, trials=5),
    "trees_dens_b8_e70": dict(label="trees", points=1000, density=(400, 20), train=256, val=128, batch=8, epochs=70, trials=5),
    "trees_dens_b16_e150": dict(label="trees", points=1000, density=(400, 20), train=256, val=128, batch=16, epochs=150, trials=5),
    "trees_dens2k_b16_e70": dict(label="trees", points=2000, density=(800, 40), train=256, val=128, batch=16, epochs=70, trials=5),
    "trees_2k_e10": dict(label="trees", points=2000, train=512, val=128, batch=32, epochs=10, trials=3),
    "trees_2k_e30": dict(label="trees", points=2000, train=512, val=128, batch=32, epochs=30, trials=3),
    "voxsum_1k_e10": dict(label="voxsum", points=1000, train=512, val=128, batch=32, epochs=10, trials=3),
    "voxsum_1k_e30": dict(label="voxsum", points=1000, train=512, val=128, batch=32, epochs=30, trials=3),
    "voxsum_2k_e30": dict(label="voxsum", points=2000, train=512, val=128, batch=32, epochs=30, trials=3),
    "voxmean_1k_e30": dict(label="voxmean", points=1000, train=512, val=128, batch=32, epochs=30, trials=3),
    "voxsum_1k_e70": dict(label="voxsum", points=1000, train=512, val=128, batch=32, epochs=70, trials=3),
    "voxsum_1k_b16_e30": dict(label="voxsum", points=1000, train=512, val=128, batch=16, epochs=30, trials=3),
}


def main():
    dev = torch.device("cuda:0")
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(VARIANTS)
    for name in names:
        for prec in (("fp32", "bf16") if "--bf16" in sys.argv else ("fp32",)):
            out = run_variant(dict(VARIANTS[name], name=name), dev, prec)
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
