"""Trains MSENet14 (reference recipe: AdaBelief lr 0.005 / wd 1e-2, cosine warm restarts per batch, clip 100, smooth-L1
on standardised targets) on synthetic labelled plots on the GPU and reports val RMSE / R2 with the reference's metric
definitions — the "val RMSE" half of BASELINE.json's metric (the NFI data is not available offline).
Usage: python tools/train_eval.py [--train 1024] [--val 256] [--epochs 6] [--points 16000] [--batch 32]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=1024)
    ap.add_argument("--val", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="SENet14")
    a = ap.parse_args()
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    from dpcr_agb_amd.metrics import RegressionMeter
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    t0 = time.time()
    mk = lambda lo, n: [synthetic.make_sparse_batch(list(range(lo + i, lo + i + a.batch)), n_points=a.points).to(dev)  # noqa: E731
                        for i in range(0, n, a.batch)]
    train, val = mk(0, a.train), mk(500_000, a.val)
    ys = torch.cat([b.y_reg for b in train]).cpu().double()
    ds = synthetic.SyntheticDataset(stat_seeds=range(0, a.train))
    ds._stats = {"mean": ys.mean(0).numpy(), "std": ys.std(0).numpy(), "min": ys.min(0).values.numpy(),
                 "max": ys.max(0).values.numpy()}
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS[a.model]), "minkowski", ds).to(dev)
    model.init_train_objects(TRAINING_NFI)
    print(f"data ready in {time.time() - t0:.1f}s: {len(train)} train / {len(val)} val batches", flush=True)
    nb = len(train)
    hist = []
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()      # manual GC (once per epoch below): a gen-2 sweep inside a step stalls the device queue for ~80 ms
    for epoch in range(a.epochs):
        model.train()
        t1 = time.time()
        order = np.random.default_rng(epoch).permutation(nb)
        for i in order:
            model.set_input(train[i], dev)
            model.optimize_parameters(epoch, a.batch, nb)
        torch.cuda.synchronize()
        dt = time.time() - t1
        gc.collect()
        model.eval()
        meter = RegressionMeter(torch.cat([b.y_reg for b in val]).cpu().double().mean(0))
        with torch.no_grad():
            for b in val:
                model.set_input(b, dev)
                model.forward()
                meter.add(model.get_reg_output(), model.get_reg_input())
        m = meter.value()
        hist.append(dict(epoch=epoch, train_plots_per_s=round(a.train / dt, 1), val_rmse=[round(v, 3) for v in m["rmse"]],
                         val_r2=[round(v, 4) for v in m["r2"]], train_loss=round(float(model.loss.detach()), 5)))
        print(json.dumps(hist[-1]), flush=True)
    print(json.dumps(dict(model=a.model, final=hist[-1], target_std=[round(float(v), 2) for v in ys.std(0)])))


if __name__ == "__main__":
    main()
