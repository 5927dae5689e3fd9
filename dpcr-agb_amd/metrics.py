"""Acceptance metrics and the per-area tracker of the reference, restated without torchnet / wandb / tensorboard.

Reference: torch_points3d/metrics/instance_tracker.py:17-178 (InstanceTracker: per-area and "total" meters per
regression target, NaN targets ignored :116-134, metric names ``{stage}_{area}_{target}_{rmse|mae|r2}`` :139-160),
metrics/meters/r2meter.py:4-26 (R2 against the mean of the stage's targets), meters/maemeter.py:4-22, and torchnet's
MSEMeter(root=True) (RMSE = sqrt(sum of squared errors / number of elements)).  The meters accumulate python floats
exactly like the reference's (``.item()`` of a torch sum per ``add``), so a run tracked batch by batch gives the same
numbers as the reference tracker would on the same predictions (golden vectors: tests/golden/metrics_golden.npz).
"""
import math
from collections import OrderedDict

import numpy as np
import torch


class RMSEMeter:
    """torchnet.meter.MSEMeter(root=True)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.n = 0
        self.sesum = 0.0

    def add(self, output, target):
        output, target = torch.as_tensor(output), torch.as_tensor(target)
        self.n += output.numel()
        self.sesum += torch.sum((output - target) ** 2).item()

    def value(self):
        return math.sqrt(self.sesum / max(1, self.n))


class MAEMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.n = 0
        self.abssum = 0.0

    def add(self, output, target):
        output, target = torch.as_tensor(output), torch.as_tensor(target)
        self.n += output.numel()
        self.abssum += torch.sum(abs(output - target)).item()

    def value(self):
        return self.abssum / max(1, self.n)


class R2Meter:
    def __init__(self, target_mean):
        self.target_mean = target_mean
        self.reset()

    def reset(self):
        self.n = 0
        self.ressum = 0.0
        self.totsum = 0.0

    def add(self, output, target):
        output, target = torch.as_tensor(output), torch.as_tensor(target)
        self.n += output.numel()
        self.ressum += torch.sum((output - target) ** 2).item()
        self.totsum += torch.sum((target - self.target_mean) ** 2).item()

    def value(self):
        return (1 - (self.ressum / self.totsum)) if self.n > 0 and self.totsum > 0 else 0.0


class RegressionMeter:
    """All targets at once (columns), fp64 sums; rows whose target is NaN (or masked out) are ignored per target,
    as the tracker does.  value(): dict of per-target lists rmse / mae / r2."""

    def __init__(self, target_mean):
        self.target_mean = torch.as_tensor(target_mean, dtype=torch.float64).reshape(1, -1)
        self.reset()

    def reset(self):
        k = self.target_mean.shape[1]
        self.n = torch.zeros(k, dtype=torch.float64)
        self.res, self.tot, self.abs = (torch.zeros(k, dtype=torch.float64) for _ in range(3))

    def add(self, output, target, mask=None):
        o, t = output.detach().double().cpu(), target.detach().double().cpu()
        ok = ~torch.isnan(t)
        if mask is not None:
            ok &= mask.detach().cpu().bool().reshape(ok.shape)
        d = torch.where(ok, o - t, torch.zeros_like(o))
        c = torch.where(ok, t - self.target_mean, torch.zeros_like(t))
        self.res += (d ** 2).sum(0)
        self.abs += d.abs().sum(0)
        self.tot += (c ** 2).sum(0)
        self.n += ok.sum(0).double()

    def value(self):
        n = self.n.clamp(min=1)
        r2 = torch.where((self.tot > 0) & (self.n > 0), 1 - self.res / self.tot.clamp(min=1e-300),
                         torch.zeros_like(self.tot))
        return dict(rmse=torch.sqrt(self.res / n).tolist(), mae=(self.abs / n).tolist(), r2=r2.tolist())


class InstanceTracker:
    """Per-area (+ "total") RMSE / MAE / R2 per regression target and the running average of the model's losses.

    dataset: needs ``has_reg_targets, reg_targets_idx, reg_targets (names), areas (mapping name -> anything),
    get_mean_targets() -> {area: {stage: array over all targets}}`` like the reference's LasDataset.
    ``track(model)`` reads ``model.get_reg_output() / get_reg_input() / get_current_losses()`` and
    ``model.data_visual.area_name`` (one name per sample)."""

    def __init__(self, dataset, stage="train", log_train_metrics=True):
        self.has_reg_targets = dataset.has_reg_targets
        self.reg_targets_idx = np.asarray(dataset.reg_targets_idx, dtype=bool)
        self.reg_targets = list(dataset.reg_targets)
        self.area_names = list(dataset.areas.keys())
        self.area_name_map = OrderedDict((name, i) for i, name in enumerate(self.area_names))
        self.target_means = dataset.get_mean_targets()
        self.log_train_metrics = log_train_metrics
        self._metric_func = {"loss": min}
        if self.has_reg_targets:
            self._metric_func.update({"_rmse": min, "loss_reg": min})
        self.reset(stage)

    @property
    def metric_func(self):
        return self._metric_func

    def _active(self):
        return self._stage != "train" or self.log_train_metrics

    def reset(self, stage="train"):
        self._stage = stage
        self._loss_sums, self._loss_counts = {}, {}
        self._rmse, self._mae, self._r2 = {}, {}, {}
        if not (self._active() and self.has_reg_targets):
            return
        areas = [a for a in self.area_names if self.target_means[a].get(stage, None) is not None] + ["total"]
        for a in areas:
            self._rmse[a], self._mae[a], self._r2[a] = {}, {}, {}
            means = np.asarray(self.target_means[a][stage], dtype=np.float64)[self.reg_targets_idx]
            for i, t in enumerate(self.reg_targets):
                if np.isnan(means[i]).all():
                    continue
                self._rmse[a][t], self._mae[a][t], self._r2[a][t] = RMSEMeter(), MAEMeter(), R2Meter(means[i])

    def track(self, model, **kwargs):
        for k, v in model.get_current_losses().items():
            if v is None:
                continue
            self._loss_sums[k] = self._loss_sums.get(k, 0.0) + float(v)
            self._loss_counts[k] = self._loss_counts.get(k, 0) + 1
        if not (self._active() and self.has_reg_targets):
            return
        names = model.data_visual["area_name"] if isinstance(model.data_visual, dict) else model.data_visual.area_name
        areas = torch.tensor([self.area_name_map[a] for a in names])
        outputs = model.get_reg_output().detach().cpu()
        targets = model.get_reg_input().detach().cpu()
        no_nans = ~torch.isnan(targets)
        for i, t in enumerate(self.reg_targets):
            ok = no_nans[:, i]
            if not ok.any():
                continue
            out, tgt, area = outputs[:, i][ok], targets[:, i][ok], areas[ok]
            for a in self.area_names:
                sel = area == self.area_name_map[a]
                if sel.any():
                    self._add(a, t, out[sel], tgt[sel])
            self._add("total", t, out, tgt)

    def _add(self, area, target, out, tgt):
        if target not in self._r2.get(area, {}):
            return
        self._rmse[area][target].add(out, tgt)
        self._mae[area][target].add(out, tgt)
        self._r2[area][target].add(out, tgt)

    def get_metrics(self, verbose=False):
        m = OrderedDict()
        for k, s in self._loss_sums.items():
            m[f"{self._stage}_{k}"] = s / self._loss_counts[k]
        if self._active() and self.has_reg_targets:
            for a in self.area_names + ["total"]:
                for t in self.reg_targets:
                    if t not in self._r2.get(a, {}):
                        continue
                    m[f"{self._stage}_{a}_{t}_rmse"] = self._rmse[a][t].value()
                    m[f"{self._stage}_{a}_{t}_mae"] = self._mae[a][t].value()
                    m[f"{self._stage}_{a}_{t}_r2"] = self._r2[a][t].value()
        return m
