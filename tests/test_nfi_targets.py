"""Target standardisation against the ONLY real data of the reference tree: nfi-data/{train,val,test}_split.csv (4 271 / 919 /
914 plots with BMag_ha, V_ha).  tests/golden/nfi_target_stats.json holds their statistics (numbers only; generator:
tests/golden/make_nfi_target_stats.py, run in the build container where /root/reference exists).

Reference: models/instance/base.py:86-114 (`get_task_weights_scale_center`: normalization "standard" -> center = train mean,
scale = train std per target; the regression loss compares (y - center) / scale, base.py:146-162) — restated in
dpcr_agb_amd/instance/base.py:_register_target_stats."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats():
    with open(os.path.join(ROOT, "tests", "golden", "nfi_target_stats.json")) as f:
        return json.load(f)


class _NFIDataset:
    """The dataset surface InstanceBase reads (datasets/instance/base_dataset.py: targets, per-area statistics)."""
    feature_dimension = 3
    has_reg_targets = True
    reg_targets_idx = np.array([True, True])
    num_reg_classes = 2
    double_batch = False
    targets = {"BMag_ha": {"task": "regression", "weight": 1}, "V_ha": {"task": "regression", "weight": 1}}

    def __init__(self, st):
        self._st = st

    def _get(self, stat):
        per_stage = {s: np.asarray(self._st[s][stat], dtype=np.float64) for s in ("train", "val", "test")}
        return {"NFI": per_stage, "total": per_stage}

    def get_mean_targets(self):
        return self._get("mean")

    def get_std_targets(self):
        return self._get("std")

    def get_min_targets(self):
        return self._get("min")

    def get_max_targets(self):
        return self._get("max")


def test_target_standardisation_on_the_nfi_split():
    from dpcr_agb_amd.config import Opt
    from dpcr_agb_amd.instance.base import InstanceBase
    st = _stats()
    assert st["train"]["rows"] == 4271 and st["val"]["rows"] == 919 and st["test"]["rows"] == 914
    model = InstanceBase(Opt(reg_loss_fn="smoothl1"), "minkowski", _NFIDataset(st))
    mean, std = np.asarray(st["train"]["mean"]), np.asarray(st["train"]["std"])
    # center / scale = the TRAIN split's mean / std of (BMag_ha, V_ha), fp32 buffers of shape [1, 2]
    assert model.reg_center_targets.shape == (1, 2) and model.reg_scale_targets.shape == (1, 2)
    assert np.allclose(model.reg_center_targets.numpy()[0], mean, rtol=1e-6)
    assert np.allclose(model.reg_scale_targets.numpy()[0], std, rtol=1e-6)
    assert torch.equal(model.reg_weights, torch.ones(2))
    # known answers: the first five plots of the train split, standardised
    y = torch.tensor(st["train"]["first_rows"], dtype=torch.float32)
    z = (y - model.reg_center_targets) / model.reg_scale_targets
    want = (np.asarray(st["train"]["first_rows"]) - mean) / std
    assert np.allclose(z.numpy(), want, rtol=1e-5, atol=1e-6)
    # ... and the report path undoes it (base.py:199-208: out * scale + center)
    back = z * model.reg_scale_targets + model.reg_center_targets
    assert torch.allclose(back, y, rtol=1e-6, atol=1e-4)


def test_synthetic_label_scale_against_the_nfi_targets():
    """The synthetic labels (synthetic.make_plot: biomass = a * sum(height^b) of the plot's trees, volume = ratio * biomass) keep the
    reference data's VOLUME / BIOMASS ratio and coefficient of variation regime; their absolute scale is a per-plot sum, about
    ten times the per-hectare NFI values — irrelevant after standardisation (R2 is scale-free), stated here so that nobody reads
    the bench line's val RMSE as tonnes per hectare."""
    from dpcr_agb_amd import synthetic
    st = _stats()["train"]
    ratio_nfi = st["mean"][1] / st["mean"][0]
    assert abs(synthetic.VOLUME_RATIO - ratio_nfi) / ratio_nfi < 0.01, (synthetic.VOLUME_RATIO, ratio_nfi)
    ys = np.stack([synthetic.make_plot(s, 256)[2] for s in range(200)]).astype(np.float64)
    assert abs(ys[:, 1].mean() / ys[:, 0].mean() - ratio_nfi) / ratio_nfi < 0.02
    scale = ys.mean(0) / np.asarray(st["mean"])
    assert 5.0 < scale[0] < 15.0 and 5.0 < scale[1] < 15.0, scale      # per-plot sums vs per-hectare values
    cv_syn, cv_nfi = ys.std(0) / ys.mean(0), np.asarray(st["std"]) / np.asarray(st["mean"])
    assert (cv_syn > 0.2).all() and (cv_syn < cv_nfi).all()            # a narrower, unimodal target distribution
