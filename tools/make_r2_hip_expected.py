"""Writes the per-trial R2 of every HIP leg of the R2 acceptance (tests/test_zz_r2_acceptance.py) as a JSON fixture.

Every leg trains with KernelOptions.deterministic_wgrad (fixed-order weight-gradient sums in fp32, bf16 and bf16 on bf16
rows): a trial is a pure function of (tree, seed) — the same numbers on every run and every MI355X box.  The fixture is
therefore a sharp regression guard: the GPU suite must REPRODUCE it (1e-9), and the statistical statements about the
HIP-vs-CPU gap are made on these known numbers, not on a fresh draw.

Run on the GPU box after ANY change that alters a summation order of the sparse path:
    python tools/make_r2_hip_expected.py [--trials 13] [--out gpurun_out/r2_hip_expected.json] [--check-repeat 2]
then copy the output to tests/golden/r2_hip_expected.json.  (The script is data generation for a fixture: it runs the
product path only; the CPU trials it is later compared with are tests/golden/r2_cpu_trials/*.json.)"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

LEGS = ("fp32", "bf16", "bf16rows")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=13)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r2_hip_expected.json"))
    ap.add_argument("--check-repeat", type=int, default=2, help="re-run the first N trials of every leg and require identity")
    ap.add_argument("--bias-grad-leg", action="store_true", help="also run the fp32 leg with KernelOptions."
                    "closed_form_bias_grad off (column sums of the incoming gradient, as the reference computes the bias "
                    "gradient of a convolution in front of a BatchNorm): key 'fp32_colsum_bias' of the output, not a leg of the test")
    ap.add_argument("--tree", default=os.environ.get("AGB_TREE"), help="commit hash of the tree the table is generated on (the "
                    "GPU box holds a snapshot without .git: pass `git rev-parse HEAD` from the build container)")
    a = ap.parse_args()
    import torch
    from train_eval import acceptance_data, acceptance_gpu_trial
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "r2_cpu_leg.json")))
    cfg = ref["config"]
    dev = torch.device("cuda:0")
    data = acceptance_data(cfg, dev)
    out = dict(config=cfg, trials=a.trials, legs={}, rmse={}, tree=a.tree or subprocess.run(
        ["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "snapshot",
        device=torch.cuda.get_device_name(0))
    for leg in LEGS:
        t0 = time.time()
        res = [acceptance_gpu_trial(cfg, t, dev, leg, data)["final"] for t in range(a.trials)]
        out["legs"][leg] = [r["r2_rs"] for r in res]
        out["rmse"][leg] = [r["rmse_rs"] for r in res]
        print(json.dumps(dict(leg=leg, r2=out["legs"][leg], seconds=round(time.time() - t0, 1))), flush=True)
        for t in range(min(a.check_repeat, a.trials)):
            again = acceptance_gpu_trial(cfg, t, dev, leg, data)["final"]["r2_rs"]
            same = again == out["legs"][leg][t]
            print(json.dumps(dict(leg=leg, trial=t, repeat=again, identical=same)), flush=True)
            if not same:
                raise SystemExit(f"leg {leg} trial {t} is not reproducible: {again} vs {out['legs'][leg][t]}")
    if a.bias_grad_leg:
        t0 = time.time()
        res = [acceptance_gpu_trial(cfg, t, dev, "fp32", data, closed_form_bias_grad=False)["final"] for t in range(a.trials)]
        out["fp32_colsum_bias"] = [r["r2_rs"] for r in res]
        print(json.dumps(dict(leg="fp32, bias gradient by column sums", r2=out["fp32_colsum_bias"],
                              seconds=round(time.time() - t0, 1))), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
