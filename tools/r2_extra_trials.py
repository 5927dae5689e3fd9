"""Supplementary trials of the R2 acceptance schedule on the HIP path (seeds beyond the five of the protocol): prints one JSON
line per (leg, trial).  Usage (GPU box): python tools/r2_extra_trials.py 5 10 [fp32 bf16 bf16rows]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    from train_eval import acceptance_data, acceptance_gpu_trial
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    legs = sys.argv[3:] or ["fp32", "bf16", "bf16rows"]
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "r2_cpu_leg.json")))["config"]
    dev = torch.device("cuda", 0)
    data = acceptance_data(cfg, dev)
    for leg in legs:
        for t in range(lo, hi):
            r = acceptance_gpu_trial(cfg, t, dev, leg, data)
            print(json.dumps(dict(leg=leg, trial=t, r2_rs=r["final"]["r2_rs"])), flush=True)


if __name__ == "__main__":
    main()
