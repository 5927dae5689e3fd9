"""Trains MSENet14 (reference recipe: AdaBelief lr 0.005 / wd 1e-2, cosine warm restarts per batch, clip 100, smooth-L1
on standardised targets) on synthetic labelled plots on the GPU and reports val RMSE / R2 with the reference's metric
definitions — the "val RMSE" half of BASELINE.json's metric (the NFI data is not available offline).
Usage: python tools/train_eval.py [--train 1024] [--val 256] [--epochs 6] [--points 16000] [--batch 32]
       python tools/train_eval.py --acceptance     # R2 acceptance: the HIP leg of tests/golden/make_r2_cpu_leg.py's
                                                   # schedule, compared with the committed CPU leg (|dR2| <= 0.005)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dpcr_agb_amd  # noqa: E402

dpcr_agb_amd.limit_host_threads()


def acceptance_gpu_leg(cfg, dev, log=None, keep=None):
    """The schedule of tests/golden/make_r2_cpu_leg.py (same initial weights, batch order, drop-path draws, recipe) on the
    HIP path.  Returns dict(history, final, val_predictions) with the reference's metric definitions."""
    import importlib.util
    import random
    spec = importlib.util.spec_from_file_location("make_r2_cpu_leg",
                                                  os.path.join(ROOT, "tests", "golden", "make_r2_cpu_leg.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)       # (imports the oracle only inside its own main())
    from dpcr_agb_amd.config import TRAINING_NFI
    from dpcr_agb_amd.metrics import RegressionMeter
    train, val = gen.batches(cfg["train_seed0"], cfg["train"], cfg), gen.batches(cfg["val_seed0"], cfg["val"], cfg)
    model = gen.build_model(cfg, train).to(dev)
    model.init_train_objects(TRAINING_NFI)
    train, val = [b.to(dev) for b in train], [b.to(dev) for b in val]
    val_mean = torch.cat([b.y_reg for b in val]).cpu().double().mean(0)
    nb = len(train)
    random.seed(cfg["drop_seed"])
    hist = []
    for epoch in range(cfg["epochs"]):
        model.train()
        for i in np.random.default_rng(epoch).permutation(nb):
            model.set_input(train[i], dev)
            model.optimize_parameters(epoch, cfg["batch"], nb)
        loss = float(model.loss.detach())
        if epoch + 1 == cfg["epochs"] and cfg.get("calibrate_passes", 0):
            model.calibrate_bn(train, dev, epochs=cfg["calibrate_passes"])
        rec = dict(epoch=epoch, train_loss=loss)
        # two evaluation protocols: "bs" = BatchNorm on the statistics of the evaluated batch (the calibrate_bn forward:
        # train mode, no gradients, drop-path off) and "rs" = running statistics (eval mode, the reference's eval.py)
        for tag in ("bs", "rs"):
            model.train(tag == "bs")
            for m in model.modules():
                if m.__class__.__name__ == "MinkowskiDropPath":
                    m.eval()
            saved = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
            meter, preds = RegressionMeter(val_mean), []
            with torch.no_grad():
                for b in val:
                    model.set_input(b, dev)
                    model.forward()
                    meter.add(model.get_reg_output(), model.get_reg_input())
                    preds.append(model.get_reg_output().detach().cpu())
            model.load_state_dict(saved, strict=False)     # the "bs" pass must not move the running statistics
            rec.update({f"{k}_{tag}": v for k, v in meter.value().items()})
            if tag == "bs":
                preds_bs = preds
        hist.append(rec)
        if log:
            log(json.dumps(hist[-1]))
    preds = preds_bs
    if keep is not None:     # hand the trained model and the validation batches to the caller (same-weights check)
        keep.update(model=model, val=val, val_mean=val_mean)
    return dict(history=hist, final=hist[-1], val_predictions=torch.cat(preds).tolist())


def acceptance(dev):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "r2_cpu_leg.json")))
    got = acceptance_gpu_leg(ref["config"], dev, log=lambda m: print("hip ", m, flush=True))
    for h in ref["history"]:
        print("cpu ", json.dumps(h))
    d_r2 = [g - c for g, c in zip(got["final"]["r2_bs"], ref["final"]["r2_bs"])]
    d_rmse = [g - c for g, c in zip(got["final"]["rmse_bs"], ref["final"]["rmse_bs"])]
    pg, pc = torch.tensor(got["val_predictions"]), torch.tensor(ref["val_predictions"])
    print(json.dumps(dict(check="same schedule on both legs (HIP path vs oracle/sparse_ref.py fp32 CPU): REPORTED; the "
                                "asserted +-0.005 check is the same-weights one in tests/test_r2_acceptance.py",
                          config=ref["config"], r2_hip=got["final"]["r2_bs"], r2_cpu=ref["final"]["r2_bs"], d_r2=d_r2,
                          rmse_hip=got["final"]["rmse_bs"], rmse_cpu=ref["final"]["rmse_bs"], d_rmse=d_rmse,
                          running_stats_protocol=dict(r2_hip=got["final"]["r2_rs"], r2_cpu=ref["final"]["r2_rs"]),
                          max_abs_prediction_diff=float((pg - pc).abs().max()),
                          passed=bool(all(abs(d) <= 0.005 for d in d_r2)))))
    return all(abs(d) <= 0.005 for d in d_r2)


def acceptance_sweep(dev):
    """Which schedule gives a well-conditioned R2 (seconds per variant on the GPU; the CPU leg takes ~25 minutes)."""
    base = dict(model="SENet14", train=256, val=64, points=4000, batch=32, epochs=10, train_seed0=0, val_seed0=500_000,
                init_seed=0, drop_seed=1234, calibrate_passes=4)
    from dpcr_agb_amd import sparse_ops
    for over in ({}, dict(precision="bf16x3"), dict(drop_seed=99), dict(drop_seed=99, precision="bf16x3"),
                 dict(train=512, points=2000), dict(train=512, points=2000, precision="bf16x3"), dict(precision="bf16")):
        cfg = dict(base, **over)
        old = sparse_ops.set_conv_precision(cfg.pop("precision", "fp32"))   # a 1e-5-level perturbation of every conv
        try:
            got = acceptance_gpu_leg(cfg, dev)
        finally:
            sparse_ops.set_conv_precision(old)
        print(json.dumps(dict(over=over, r2_bs_path=[[round(v, 4) for v in h["r2_bs"]] for h in got["history"]],
                              final=got["final"])), flush=True)


def main():
    if "--acceptance-sweep" in sys.argv:
        return acceptance_sweep(torch.device("cuda:0"))
    if "--acceptance" in sys.argv:
        sys.exit(0 if acceptance(torch.device("cuda:0")) else 1)
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
