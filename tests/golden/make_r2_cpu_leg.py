"""CPU leg of the R2 acceptance check (BASELINE.json north_star: "test-set R2 within +-0.005"; SURVEY.md §8c last row:
the HIP path against this repo's CPU restatement on one synthetic labelled set, the NFI data being unavailable offline).

Protocol = the reference's own (README.md:24-56: every published R2 is the MEDIAN OF 5 TRIALS): five trainings of MSENet14
that differ only in their seeds (initial weights, batch order, drop-path draws), the reference recipe (AdaBelief lr 0.005 /
wd 1e-2, clip 100, cosine warm restarts T_0 = 10, T_mult = 2 stepped per batch, smooth-L1 on standardised targets, drop-path
0.01), run on ``oracle/sparse_ref.py`` in fp32 on the CPU; validation metrics as metrics/instance_tracker.py:85-87 and
meters/r2meter.py:15-26 define them.  tests/test_zz_r2_acceptance.py runs the same five seeds on the HIP path and compares
medians.

The set (found with tools/r2_explore.py on the GPU, profiles/r03_r2_explore*.log): 256 training / 128 validation plots whose
point count follows the stand (400 + 20 returns per tree: 700-1600 points), batch 16, 150 epochs = the end of the fourth
cosine cycle (10 + 20 + 40 + 80), where the learning rate is ~0 and the validation R2 of BOTH targets has plateaued at
~0.77 with a trial-to-trial standard deviation of 0.010 (30 epochs: 0.52 +- 0.06; 70 epochs: 0.72 +- 0.02).

    python tests/golden/make_r2_cpu_leg.py trial=0        # one trial -> r2_cpu_trials/trial_0.json (~40 min on 8 cores)
    python tests/golden/make_r2_cpu_leg.py merge          # trials -> r2_cpu_leg.json (the committed fixture)
Trials are independent processes (run two or three at a time with threads=3 on an 8-core container).
"""
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

CFG = dict(model="SENet14", train=256, val=128, batch=16, density=[400, 20], epochs=150, train_seed0=0,
           val_seed0=500_000, calibrate_passes=2, trials=5, eval_every=25)
TRIAL_DIR = os.path.join(HERE, "r2_cpu_trials")


def trial_seeds(t):
    """What distinguishes trial t: initial weights, drop-path draws, batch order (shuffle_rng(t, epoch))."""
    return dict(init_seed=int(t), drop_seed=1234 + int(t))


def shuffle_rng(t, epoch):
    return np.random.default_rng(1000 * int(t) + int(epoch))


def batches(seed0, n, cfg):
    from dpcr_agb_amd import synthetic
    return [synthetic.make_sparse_batch(list(range(seed0 + i, seed0 + i + cfg["batch"])), density=tuple(cfg["density"]))
            for i in range(0, n, cfg["batch"])]


def build_model(cfg, train, trial):
    """The product's model object on the CPU (for its initial weights and target statistics only)."""
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    ys = torch.cat([b.y_reg for b in train]).double()
    ds = synthetic.SyntheticDataset(stat_seeds=range(0, 8))
    ds._stats = {"mean": ys.mean(0).numpy(), "std": ys.std(0).numpy(), "min": ys.min(0).values.numpy(),
                 "max": ys.max(0).values.numpy()}
    torch.manual_seed(trial_seeds(trial)["init_seed"])
    return MinkowskiBaselineModel(Opt(MODEL_OPTIONS[cfg["model"]]), "minkowski", ds)


def run_trial(cfg, trial, threads):
    from oracle import sparse_ref as R
    from dpcr_agb_amd.metrics import RegressionMeter
    from dpcr_agb_amd.optim import AdaBelief
    torch.set_num_threads(threads)
    train, val = batches(cfg["train_seed0"], cfg["train"], cfg), batches(cfg["val_seed0"], cfg["val"], cfg)
    model = build_model(cfg, train, trial)
    center, scale, w = model.reg_center_targets, model.reg_scale_targets, model.reg_weights
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)
          for k, v in model.model.state_dict().items()}
    head = [v for k, v in sd.items() if v.requires_grad and "final.linears" in k]
    backbone = [v for k, v in sd.items() if v.requires_grad and "final.linears" not in k]
    opt = AdaBelief([{"params": head}, {"params": backbone}], lr=0.005, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2)
    nb = len(train)
    val_mean = torch.cat([b.y_reg for b in val]).double().mean(0)

    def coords_of(b):
        return torch.cat([b.batch[:, None], b.coords.long()], 1).numpy()

    # every batch is revisited each epoch: its coordinate levels / kernel maps / pair lists are built once
    cms = {}

    def cm_of(tag, i, b):
        if (tag, i) not in cms:
            cm = R.Coords(coords_of(b), len(b))
            cm.cache_pairs = True
            cms[(tag, i)] = cm
        return cms[(tag, i)]

    def evaluate(training):
        meter, preds = RegressionMeter(val_mean), []
        with torch.no_grad():
            for i, b in enumerate(val):
                out = R.resnet_forward(sd, None, b.x, (1, 1, 1, 1), batch_size=len(b), training=training,
                                       cm=cm_of("val", i, b))
                pred = out * scale + center
                meter.add(pred, b.y_reg)
                preds.append(pred)
        return meter.value(), torch.cat(preds)

    random.seed(trial_seeds(trial)["drop_seed"])
    hist, seen, t0 = [], 0, time.time()
    for epoch in range(cfg["epochs"]):
        for i in shuffle_rng(trial, epoch).permutation(nb):
            b = train[i]
            upd = {}
            out = R.resnet_forward(sd, None, b.x, (1, 1, 1, 1), batch_size=len(b), drop_path_prob=0.01, update=upd,
                                   cm=cm_of("train", int(i), b))
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
        last = epoch + 1 == cfg["epochs"]
        if last:
            # calibrate_bn flow (trainer.py:230-283, README.md:156-171): forward-only passes in train mode
            with torch.no_grad():
                for _ in range(cfg["calibrate_passes"]):
                    for i, b in enumerate(train):
                        upd = {}
                        R.resnet_forward(sd, None, b.x, (1, 1, 1, 1), batch_size=len(b), drop_path_prob=0.01, update=upd,
                                         cm=cm_of("train", i, b))
                        for k, v in upd.items():
                            sd[k] = v
        if last or (epoch + 1) % cfg["eval_every"] == 0:
            rec = dict(epoch=epoch + 1, train_loss=float(loss.detach()))
            m, preds = evaluate(training=False)          # running statistics: the reference's eval.py protocol
            rec.update({f"{k}_rs": v for k, v in m.items()})
            if last:
                mb, _ = evaluate(training=True)          # statistics of the evaluated batch (calibrate_bn's forward)
                rec.update({f"{k}_bs": v for k, v in mb.items()})
            hist.append(rec)
            print(f"trial {trial}", json.dumps(rec), f"[{time.time() - t0:.0f}s]", flush=True)
    return dict(trial=trial, seeds=trial_seeds(trial), config=cfg, history=hist, final=hist[-1],
                val_predictions=preds.tolist(), threads=threads, seconds=round(time.time() - t0, 1))


def merge(cfg):
    trials = []
    for t in range(cfg["trials"]):
        with open(os.path.join(TRIAL_DIR, f"trial_{t}.json")) as f:
            trials.append(json.load(f))
        assert trials[-1]["config"] == cfg, f"trial {t} was run with another configuration"
    r2 = np.array([tr["final"]["r2_rs"] for tr in trials])
    out = dict(config=cfg, protocol="median of 5 trials (reference README.md:24-56); R2 per meters/r2meter.py:15-26",
               trials=[{k: tr[k] for k in ("trial", "seeds", "history", "final", "seconds", "threads")} for tr in trials],
               r2_rs=r2.tolist(), median_r2_rs=np.median(r2, 0).tolist(), std_r2_rs=r2.std(0, ddof=1).tolist(),
               spread_r2_rs=(r2.max(0) - r2.min(0)).tolist(),
               val_predictions_trial0=trials[0]["val_predictions"])
    with open(os.path.join(HERE, "r2_cpu_leg.json"), "w") as f:
        json.dump(out, f)
    print("written r2_cpu_leg.json: median R2", out["median_r2_rs"], "std", out["std_r2_rs"])


def main():
    args = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
    cfg = dict(CFG)
    for k in ("epochs", "train", "val", "batch", "calibrate_passes", "eval_every"):
        if k in args:
            cfg[k] = int(args[k])
    if "merge" in sys.argv[1:]:
        return merge(cfg)
    trial = int(args.get("trial", 0))
    threads = int(args.get("threads", max(1, min(8, os.cpu_count() or 1))))
    res = run_trial(cfg, trial, threads)
    os.makedirs(TRIAL_DIR, exist_ok=True)
    with open(os.path.join(TRIAL_DIR, f"trial_{trial}.json"), "w") as f:
        json.dump(res, f)
    print(f"written r2_cpu_trials/trial_{trial}.json")


if __name__ == "__main__":
    main()
