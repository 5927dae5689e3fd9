"""dpcr_agb_amd/metrics.py against golden vectors from the reference's own R2Meter / MAEMeter
(tests/golden/make_metrics_golden.py) — per-area and total RMSE / MAE / R2 with missing (NaN) targets — and the
model <-> trainer contract pieces of InstanceBase (models/instance/base.py:86-185): target standardisation, masked
smooth-L1 loss, de-standardised report."""
import os
import types

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_golden.npz")


class _DS:
    has_reg_targets = True
    reg_targets_idx = np.array([True, True])
    reg_targets = ["BMag_ha", "V_ha"]

    def __init__(self, g):
        self.areas = {str(a): None for a in g["areas"]}
        self._means = {a: {"val": g[f"mean/{a}"], "train": g[f"mean/{a}"]} for a in list(self.areas) + ["total"]}

    def get_mean_targets(self):
        return self._means


def test_tracker_matches_reference_meters():
    from dpcr_agb_amd.metrics import InstanceTracker
    g = np.load(GOLD)
    ds = _DS(g)
    tr = InstanceTracker(ds, stage="val")
    for b in range(g["y"].shape[0]):
        model = types.SimpleNamespace(
            get_reg_output=lambda b=b: torch.from_numpy(g["out"][b]), get_reg_input=lambda b=b: torch.from_numpy(g["y"][b]),
            get_current_losses=lambda b=b: {"loss_reg": float(b)},
            data_visual={"area_name": [str(g["areas"][i]) for i in g["area"][b]]})
        tr.track(model)
    m = tr.get_metrics()
    assert m["val_loss_reg"] == pytest.approx(np.mean(range(g["y"].shape[0])))
    n = 0
    for key in g.files:
        if not key.startswith("res/"):
            continue
        area, t = key[4:].split("/")
        name = ds.reg_targets[int(t)]
        rmse, mae, r2 = g[key]
        assert m[f"val_{area}_{name}_rmse"] == pytest.approx(rmse, rel=1e-12)
        assert m[f"val_{area}_{name}_mae"] == pytest.approx(mae, rel=1e-12)
        assert m[f"val_{area}_{name}_r2"] == pytest.approx(r2, rel=1e-12)
        n += 1
    assert n == 8


def test_regression_meter_masks_missing_targets():
    from dpcr_agb_amd.metrics import RegressionMeter
    g = np.load(GOLD)
    meter = RegressionMeter(g["mean/total"])
    for b in range(g["y"].shape[0]):
        meter.add(torch.from_numpy(g["out"][b]), torch.from_numpy(g["y"][b]))
    v = meter.value()
    for t in range(2):
        rmse, mae, r2 = g[f"res/total/{t}"]
        assert v["rmse"][t] == pytest.approx(rmse, rel=1e-6)
        assert v["mae"][t] == pytest.approx(mae, rel=1e-6)
        assert v["r2"][t] == pytest.approx(r2, rel=1e-6)


class _Net(torch.nn.Module):
    def forward(self, x):
        return x


def test_instance_base_target_scaling_and_loss():
    """base.py:86-114 (centre / scale buffers from the dataset's train statistics, averaged over areas, overrides and
    scale_mult), :139-146 (output slice + activation), :154-179 (masked loss on standardised targets times the mean task
    weight), :181-185 (de-standardised report)."""
    import torch.nn.functional as F
    from dpcr_agb_amd.config import Opt
    from dpcr_agb_amd.instance.base import InstanceBase
    ds = types.SimpleNamespace(
        has_reg_targets=True, reg_targets_idx=np.array([True, True]), num_reg_classes=2, double_batch=False,
        targets=Opt(a=Opt(task="regression", weight=0.5), b=Opt(task="regression", weight=0.25, scale_mult=2.0,
                                                                center_override=7.0)),
        get_mean_targets=lambda: {"n": {"train": np.array([10.0, 20.0])}, "s": {"train": np.array([30.0, np.nan])},
                                  "t": {"val": np.array([0.0, 0.0])}},
        get_std_targets=lambda: {"n": {"train": np.array([2.0, 4.0])}, "s": {"train": np.array([4.0, 6.0])}},
        get_min_targets=lambda: {}, get_max_targets=lambda: {})
    m = InstanceBase(Opt(reg_loss_fn="smoothl1"), "x", ds)
    assert torch.allclose(m.reg_center_targets, torch.tensor([[20.0, 7.0]]))     # nanmean over the train areas; override
    assert torch.allclose(m.reg_scale_targets, torch.tensor([[3.0, 10.0]]))      # mean std; scale_mult 2
    assert torch.allclose(m.reg_weights, torch.tensor([0.5, 0.25]))
    out = torch.tensor([[0.5, -1.0, 9.0], [2.0, 0.25, 9.0], [-3.0, 1.0, 9.0]])
    y = torch.tensor([[23.0, 17.0], [26.0, 7.0], [11.0, -3.0]])
    mask = torch.tensor([[True, True], [True, False], [True, True]])
    m.reg_out = m.convert_outputs(out)
    assert m.reg_out.shape == (3, 2)
    m.reg_y, m.reg_y_mask, m._reg_mask_all = y, mask, False
    m.compute_loss()
    labels = (y - m.reg_center_targets) / m.reg_scale_targets
    want = 0.375 * F.smooth_l1_loss(out[:, :2][mask], labels[mask])
    assert float(m.loss) == pytest.approx(float(want), rel=1e-6)
    assert torch.allclose(m.get_reg_output(), out[:, :2] * m.reg_scale_targets + m.reg_center_targets)
    assert m.get_current_losses()["loss_reg"] == pytest.approx(float(want) / 0.375, rel=1e-6)
    # all targets present: the unmasked path gives the same value as a full mask
    m._reg_mask_all = True
    m.compute_loss()
    assert float(m.loss) == pytest.approx(float(0.375 * F.smooth_l1_loss(out[:, :2], labels)), rel=1e-6)
    with pytest.raises(NotImplementedError):
        InstanceBase(Opt(double_batch=True), "x", ds)
