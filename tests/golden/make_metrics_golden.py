"""Golden vectors for dpcr_agb_amd/metrics.py from the reference's own meters (importable with torch only):
torch_points3d/metrics/meters/r2meter.py and maemeter.py are imported from /root/reference and fed seeded batches
(with NaN targets removed per target, as instance_tracker.py:116-134 does before calling them).  torchnet (MSEMeter) is
not installed: its root-MSE is recomputed here from its definition (sum of squared errors / n, square root).
    python tests/golden/make_metrics_golden.py
"""
import importlib.util
import math
import os

import numpy as np
import torch

REF = "/root/reference/torch-points3d/torch_points3d/metrics/meters"


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    R2Meter, MAEMeter = _load("r2meter").R2Meter, _load("maemeter").MAEMeter
    rng = np.random.default_rng(7)
    n_batches, B = 6, 13
    areas = np.array(["north", "south", "east"])
    y = rng.normal(200, 80, size=(n_batches, B, 2)).astype(np.float32)
    out = (y + rng.normal(0, 30, size=y.shape)).astype(np.float32)
    y[rng.uniform(size=y.shape) < 0.15] = np.nan          # missing targets
    area = rng.integers(0, 3, size=(n_batches, B))
    means = {a: rng.normal(200, 10, size=2) for a in list(areas) + ["total"]}
    res = {}
    for ai, a in enumerate(list(areas) + ["total"]):
        for t in range(2):
            r2, mae, se, n = R2Meter(means[a][t]), MAEMeter(), 0.0, 0
            for b in range(n_batches):
                ok = ~np.isnan(y[b, :, t])
                if a != "total":
                    ok &= area[b] == ai
                if not ok.any():
                    continue
                o, tt = torch.from_numpy(out[b, ok, t]), torch.from_numpy(y[b, ok, t])
                r2.add(o, tt)
                mae.add(o, tt)
                se += torch.sum((o - tt) ** 2).item()
                n += int(ok.sum())
            res[f"{a}/{t}"] = np.array([math.sqrt(se / max(1, n)), mae.value(), r2.value()])
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_golden.npz"), y=y, out=out, area=area,
             areas=areas, **{f"mean/{a}": v for a, v in means.items()}, **{f"res/{k}": v for k, v in res.items()})
    print("written metrics_golden.npz", {k: v.round(4).tolist() for k, v in list(res.items())[:3]})


if __name__ == "__main__":
    main()
