"""CPU leg of the R2 acceptance check (BASELINE.json north_star: "test-set R2 within +-0.005"; SURVEY.md §8c last row:
the HIP path against this repo's CPU restatement on one synthetic labelled set, the NFI data being unavailable offline).

Trains MSENet14 with the reference recipe (AdaBelief lr 0.005 / wd 1e-2, clip 100, cosine warm restarts stepped per
batch, smooth-L1 on standardised targets, drop-path 0.01) on ``oracle/sparse_ref.py`` in fp32 on the CPU and writes the
validation metrics (RMSE / MAE / R2 as metrics/instance_tracker.py:85-87 and meters/r2meter.py:15-26 define them) and
the final predictions to tests/golden/r2_cpu_leg.json.  tests/test_r2_acceptance.py runs the identical schedule (same
initial weights, batch order, drop-path draws) on the HIP path and compares.

    python tests/golden/make_r2_cpu_leg.py            # ~10 minutes on 8 cores
"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

# epochs = 10 = one full cosine cycle (T_0 = 10): the learning rate ends at ~0.  calibrate_passes: forward-only passes
# over the training batches in train mode (the reference's calibrate_bn.py flow, trainer.py:230-283, README.md:156-171)
# before the final evaluation — BatchNorm running statistics lag the weights badly after so few steps otherwise.
CFG = dict(model="SENet14", train=256, val=64, points=4000, batch=32, epochs=10, train_seed0=0, val_seed0=500_000,
           init_seed=0, drop_seed=1234, calibrate_passes=4)


def batches(seed0, n, cfg):
    from dpcr_agb_amd import synthetic
    return [synthetic.make_sparse_batch(list(range(seed0 + i, seed0 + i + cfg["batch"])), n_points=cfg["points"])
            for i in range(0, n, cfg["batch"])]


def build_model(cfg, train):
    """The product's model object on the CPU (for its initial weights and target statistics only)."""
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    ys = torch.cat([b.y_reg for b in train]).double()
    ds = synthetic.SyntheticDataset(stat_seeds=range(0, 8))
    ds._stats = {"mean": ys.mean(0).numpy(), "std": ys.std(0).numpy(), "min": ys.min(0).values.numpy(),
                 "max": ys.max(0).values.numpy()}
    torch.manual_seed(cfg["init_seed"])
    return MinkowskiBaselineModel(Opt(MODEL_OPTIONS[cfg["model"]]), "minkowski", ds)


def main():
    if len(sys.argv) > 1:      # overrides "key=value ..." (all integers), e.g. after a sweep on the GPU
        CFG.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[1:]})
    from oracle import sparse_ref as R
    from dpcr_agb_amd.metrics import RegressionMeter
    from dpcr_agb_amd.optim import AdaBelief
    cfg = CFG
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    train, val = batches(cfg["train_seed0"], cfg["train"], cfg), batches(cfg["val_seed0"], cfg["val"], cfg)
    model = build_model(cfg, train)
    center, scale, w = model.reg_center_targets, model.reg_scale_targets, model.reg_weights
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)
          for k, v in model.model.state_dict().items()}
    head = [v for k, v in sd.items() if v.requires_grad and "final.linears" in k]
    backbone = [v for k, v in sd.items() if v.requires_grad and "final.linears" not in k]
    opt = AdaBelief([{"params": head}, {"params": backbone}], lr=0.005, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2)
    nb = len(train)
    val_mean = torch.cat([b.y_reg for b in val]).double().mean(0)
    random.seed(cfg["drop_seed"])
    hist, seen = [], 0
    t0 = time.time()
    for epoch in range(cfg["epochs"]):
        for i in np.random.default_rng(epoch).permutation(nb):
            b = train[i]
            coords = torch.cat([b.batch[:, None], b.coords.long()], 1).numpy()
            upd = {}
            out = R.resnet_forward(sd, coords, b.x, (1, 1, 1, 1), batch_size=len(b), drop_path_prob=0.01, update=upd)
            loss = R.reg_loss(out, b.y_reg, center, scale, w)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_value_(head + backbone, 100)
            opt.step()
            seen += 1
            sched.step(seen / nb)
            for k, v in upd.items():
                sd[k] = v
            for k in sd:
                if k.endswith("num_batches_tracked"):
                    sd[k] = sd[k] + 1
        if epoch + 1 == cfg["epochs"]:
            with torch.no_grad():
                for _ in range(cfg["calibrate_passes"]):
                    for b in train:
                        coords = torch.cat([b.batch[:, None], b.coords.long()], 1).numpy()
                        upd = {}
                        R.resnet_forward(sd, coords, b.x, (1, 1, 1, 1), batch_size=len(b), drop_path_prob=0.01,
                                         update=upd)
                        for k, v in upd.items():
                            sd[k] = v
        rec = dict(epoch=epoch, train_loss=float(loss.detach()))
        # "bs": BatchNorm on the statistics of the evaluated batch (train-mode forward, no gradients, drop-path off);
        # "rs": running statistics (the reference's eval.py)
        for tag in ("bs", "rs"):
            meter, preds = RegressionMeter(val_mean), []
            with torch.no_grad():
                for b in val:
                    coords = torch.cat([b.batch[:, None], b.coords.long()], 1).numpy()
                    out = R.resnet_forward(sd, coords, b.x, (1, 1, 1, 1), batch_size=len(b), training=tag == "bs")
                    pred = out * scale + center
                    meter.add(pred, b.y_reg)
                    preds.append(pred)
            rec.update({f"{k}_{tag}": v for k, v in meter.value().items()})
            if tag == "bs":
                preds_bs = preds
        hist.append(rec)
        print(json.dumps(hist[-1]), f"[{time.time() - t0:.0f}s]", flush=True)
    preds = preds_bs
    out = dict(config=cfg, history=hist, final=hist[-1], val_predictions=torch.cat(preds).tolist(),
               threads=torch.get_num_threads(), seconds=round(time.time() - t0, 1))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r2_cpu_leg.json"), "w") as f:
        json.dump(out, f)
    print("written r2_cpu_leg.json")


if __name__ == "__main__":
    main()
