"""Markdown table of the R2 acceptance run: the committed CPU trials (tests/golden/r2_cpu_leg.json) beside the HIP legs of
`python tools/train_eval.py --acceptance` (its stdout, one JSON object per leg).  Usage: python tools/r2_report.py <log>"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(log):
    legs = [json.loads(ln) for ln in open(log) if ln.startswith("{") and '"leg"' in ln]
    cpu = next(l for l in legs if l["leg"].startswith("cpu"))
    rc = np.array(cpu["r2"])
    se = lambda r: 1.2533 * r.std(0, ddof=1) / math.sqrt(len(r))  # noqa: E731
    print("| leg | trials (R² biomass / volume) | median | std | gap of medians to CPU | in s.e. of the difference | bare ±0.005 |")
    print("|---|---|---|---|---|---|---|")
    for l in legs:
        r = np.array(l["r2"])
        med = np.median(r, 0)
        row = f"| {l['leg']} | " + ", ".join(f"{a:.4f} / {b:.4f}" for a, b in r) + f" | {med[0]:.4f} / {med[1]:.4f} | " \
              f"{r.std(0, ddof=1)[0]:.4f} / {r.std(0, ddof=1)[1]:.4f} | "
        if l is cpu:
            row += "— | — | — |"
        else:
            gap = med - np.median(rc, 0)
            s = np.sqrt(se(r) ** 2 + se(rc) ** 2)
            row += f"{gap[0]:+.4f} / {gap[1]:+.4f} | {abs(gap[0]) / s[0]:.2f} / {abs(gap[1]) / s[1]:.2f} | " \
                   f"{'met' if (np.abs(gap) <= 0.005).all() else 'not met'} |"
        print(row)


if __name__ == "__main__":
    main(sys.argv[1])
