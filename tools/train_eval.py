"""Trains MSENet14 (reference recipe: AdaBelief lr 0.005 / wd 1e-2, cosine warm restarts per batch, clip 100, smooth-L1
on standardised targets) on synthetic labelled plots on the GPU and reports val RMSE / R2 with the reference's metric
definitions — the "val RMSE" half of BASELINE.json's metric (the NFI data is not available offline).
Usage: python tools/train_eval.py [--train 1024] [--val 256] [--epochs 6] [--points 16000] [--batch 32]
       python tools/train_eval.py --acceptance     # R2 acceptance: the five trials of tests/golden/make_r2_cpu_leg.py's
                                                   # schedule on the HIP path (fp32 and bf16), medians against the
                                                   # committed CPU trials (|d median R2| <= 0.005)"""
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


def _load_cpu_leg_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_r2_cpu_leg",
                                                  os.path.join(ROOT, "tests", "golden", "make_r2_cpu_leg.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)       # (imports the oracle only inside its own run_trial())
    return gen


def acceptance_data(cfg, dev):
    """The acceptance set of tests/golden/make_r2_cpu_leg.py on the device: (train, val, host train batches, val mean)."""
    gen = _load_cpu_leg_module()
    train_h, val_h = gen.batches(cfg["train_seed0"], cfg["train"], cfg), gen.batches(cfg["val_seed0"], cfg["val"], cfg)
    val_mean = torch.cat([b.y_reg for b in val_h]).double().mean(0)
    return [b.to(dev) for b in train_h], [b.to(dev) for b in val_h], train_h, val_mean


def acceptance_gpu_trial(cfg, trial, dev, precision="fp32", data=None, log=None, keep=None, deterministic=True,
                         **kernel_options):
    """Trial `trial` of the schedule of tests/golden/make_r2_cpu_leg.py (same initial weights, batch order, drop-path draws,
    recipe, calibrate_bn passes, running-statistics evaluation) on the HIP path, in the given operand precision.
    Returns dict(history, final) with the reference's metric definitions."""
    import random
    gen = _load_cpu_leg_module()
    from dpcr_agb_amd.config import TRAINING_NFI
    train, val, train_h, val_mean = data if data is not None else acceptance_data(cfg, dev)
    model = gen.build_model(cfg, train_h, trial).to(dev)
    # fixed-order weight-gradient sums (every operand precision since round 4): a trial is bitwise reproducible from run to
    # run and from box to box — the test's outcome is not a draw.  deterministic=False: the default atomic kernels.
    # ("bf16rows": bf16 operands AND bf16 row storage, KernelOptions.bf16_activations — BASELINE config 5's fastest mode)
    rows16 = precision == "bf16rows"
    model.set_kernel_options(precision="bf16" if rows16 else precision, bf16_activations=rows16,
                             deterministic_wgrad=bool(deterministic), **kernel_options)
    model.init_train_objects(TRAINING_NFI)
    nb = len(train)
    random.seed(gen.trial_seeds(trial)["drop_seed"])
    hist = []
    for epoch in range(cfg["epochs"]):
        model.train()
        for i in gen.shuffle_rng(trial, epoch).permutation(nb):
            model.set_input(train[i], dev)
            model.optimize_parameters(epoch, cfg["batch"], nb)
        last = epoch + 1 == cfg["epochs"]
        if last:
            loss = float(model.loss.detach())
            model.calibrate_bn(train, dev, epochs=cfg["calibrate_passes"])
        if last or (epoch + 1) % cfg["eval_every"] == 0:
            rec = dict(epoch=epoch + 1, train_loss=loss if last else float(model.loss.detach()))
            rec.update({f"{k}_rs": v for k, v in model.evaluate(val, dev, val_mean).items()})
            hist.append(rec)
            if log:
                log(json.dumps(rec))
    if keep is not None:     # hand the trained model and the validation batches to the caller (same-weights check)
        keep.update(model=model, val=val, val_mean=val_mean)
    return dict(trial=trial, precision=precision, history=hist, final=hist[-1])


def acceptance(dev, precisions=("fp32", "bf16"), trials=None):
    """R2 acceptance, the reference's protocol (median of 5 trials): HIP legs against the committed CPU trials.  Prints the
    spread table; returns True when every |median R2_HIP - median R2_CPU| <= 0.005."""
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "r2_cpu_leg.json")))
    cfg = ref["config"]
    n = trials or cfg["trials"]
    cpu = np.array(ref["r2_rs"])
    data = acceptance_data(cfg, dev)
    print(json.dumps(dict(leg="cpu (oracle/sparse_ref.py fp32)", r2=cpu.round(4).tolist(), median=np.median(cpu, 0).tolist(),
                          std=cpu.std(0, ddof=1).tolist())), flush=True)
    ok = True
    for prec in precisions:
        t0 = time.time()
        r2 = np.array([acceptance_gpu_trial(cfg, t, dev, prec, data)["final"]["r2_rs"] for t in range(n)])
        gap = np.median(r2, 0) - np.median(cpu, 0)
        ok = ok and bool((np.abs(gap) <= 0.005).all())
        print(json.dumps(dict(leg=f"hip {prec}", r2=r2.round(4).tolist(), median=np.median(r2, 0).tolist(),
                              std=r2.std(0, ddof=1).tolist(), median_gap_to_cpu=gap.tolist(),
                              within_0p005=bool((np.abs(gap) <= 0.005).all()), seconds=round(time.time() - t0, 1))),
              flush=True)
    return ok


def main():
    if "--acceptance-hip-only" in sys.argv:     # the HIP trials alone (before the CPU fixture exists; run-to-run spread)
        cfg = _load_cpu_leg_module().CFG
        dev = torch.device("cuda:0")
        data = acceptance_data(cfg, dev)
        for prec in ("fp32", "fp32", "bf16"):
            t0 = time.time()
            r2 = np.array([acceptance_gpu_trial(cfg, t, dev, prec, data)["final"]["r2_rs"] for t in range(cfg["trials"])])
            print(json.dumps(dict(leg=f"hip {prec}", r2=r2.round(5).tolist(), median=np.median(r2, 0).tolist(),
                                  std=r2.std(0, ddof=1).tolist(), seconds=round(time.time() - t0, 1))), flush=True)
        return
    if "--acceptance" in sys.argv:
        sys.exit(0 if acceptance(torch.device("cuda:0")) else 1)
    if "--acceptance-trial" in sys.argv:        # ONE trial of the acceptance schedule (bench.py's val_rmse / val_r2), ~11 s
        trial = int(sys.argv[sys.argv.index("--acceptance-trial") + 1])
        cfg = _load_cpu_leg_module().CFG
        dev = torch.device("cuda:0")
        t0 = time.time()
        res = acceptance_gpu_trial(cfg, trial, dev, "fp32", acceptance_data(cfg, dev))
        f = res["final"]
        print(json.dumps(dict(model=cfg["model"], trial=trial, final=dict(epoch=f["epoch"] - 1, val_rmse=f["rmse_rs"], val_r2=f["r2_rs"],
                                                                          train_loss=f["train_loss"]),
                              config=cfg, seconds=round(time.time() - t0, 1))), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=1024)
    ap.add_argument("--val", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="SENet14")
    ap.add_argument("--deterministic", action="store_true", help="fixed-order weight-gradient sums and seeded drop-path draws: "
                    "the run is bitwise repeatable (what bench.py reports as val_rmse / val_r2)")
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
    if a.deterministic:
        import random
        random.seed(0)
        model.set_kernel_options(deterministic_wgrad=True)
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
