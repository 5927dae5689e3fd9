"""R2 of the three operand modes at the WORKLOAD's plot size.  The acceptance table (tests/golden/r2_hip_expected.json, DESIGN.md
section 6) lives on 700-1600-point plots because its CPU leg (the oracle) needs minutes per 16 k-point plot; this run repeats the
HIP side of that protocol — same recipe, same trial seeds (initial weights, batch order, drop-path draws), calibrate_bn,
running-statistics evaluation — on plots of 10-21 k points (base 6000 + 250 returns per tree) and reports, per trial, the R2 of
fp32, bf16 and bf16 rows and the paired differences to fp32: what the reduced-precision modes cost where the benchmark runs.

    python tools/r2_workload_size.py [--trials 5] [--train 128] [--val 64] [--epochs 60]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=5)
    ap.add_argument("--train", type=int, default=128)
    ap.add_argument("--val", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--density", type=str, default="6000,250")
    args = ap.parse_args()
    import train_eval as TE
    gen = TE._load_cpu_leg_module()
    cfg = dict(gen.CFG, train=args.train, val=args.val, epochs=args.epochs, eval_every=args.epochs,
               density=[int(v) for v in args.density.split(",")])
    dev = torch.device("cuda:0")
    data = TE.acceptance_data(cfg, dev)
    pts = [int(b.coords.shape[0]) // len(b) for b in data[0]]
    print(f"[r2_workload_size] {args.train} / {args.val} plots, {int(np.mean(pts))} voxels per plot on average "
          f"({min(pts)}-{max(pts)} per batch mean), batch {cfg['batch']}, {args.epochs} epochs", flush=True)
    modes = ("fp32", "bf16", "bf16rows")
    table = {m: [] for m in modes}
    for t in range(args.trials):
        for m in modes:
            res = TE.acceptance_gpu_trial(cfg, t, dev, precision=m, data=data)
            table[m].append(res["final"])
        row = {m: [round(float(v), 4) for v in table[m][-1]["r2_rs"]] for m in modes}
        print(f"[r2_workload_size] trial {t}: R2 (biomass, volume) " + json.dumps(row), flush=True)
    r2 = {m: np.array([f["r2_rs"] for f in table[m]], dtype=np.float64) for m in modes}
    out = dict(config=cfg, voxels_per_plot=int(np.mean(pts)), trials=args.trials,
               r2={m: r2[m].round(5).tolist() for m in modes},
               median={m: np.median(r2[m], 0).round(4).tolist() for m in modes},
               paired_diff_to_fp32={m: dict(mean=(r2[m] - r2["fp32"]).mean(0).round(4).tolist(),
                                            sem=((r2[m] - r2["fp32"]).std(0, ddof=1) / np.sqrt(args.trials)).round(4).tolist())
                                    for m in modes[1:]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
