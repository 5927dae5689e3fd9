"""Acceptance metrics of the reference's InstanceTracker, restated (the tracker/wandb machinery itself is out of
scope): RMSE = sqrt(mean((pred - y)^2)) (torchnet MSEMeter(root=True), metrics/instance_tracker.py:85), MAE, and
R2 = 1 - sum((pred - y)^2) / sum((y - mean_of_the_stage's_targets)^2) (metrics/meters/r2meter.py:15-26)."""
import torch


class RegressionMeter:
    def __init__(self, target_mean):
        self.target_mean = torch.as_tensor(target_mean, dtype=torch.float64).reshape(1, -1)
        self.reset()

    def reset(self):
        self.n = 0
        self.res = self.tot = self.abs = None

    def add(self, output, target):
        o, t = output.detach().double().cpu(), target.detach().double().cpu()
        res, tot, ab = ((o - t) ** 2).sum(0), ((t - self.target_mean) ** 2).sum(0), (o - t).abs().sum(0)
        if self.n == 0:
            self.res, self.tot, self.abs = res, tot, ab
        else:
            self.res, self.tot, self.abs = self.res + res, self.tot + tot, self.abs + ab
        self.n += o.shape[0]

    def value(self):
        """dict of per-target lists: rmse, mae, r2"""
        if self.n == 0:
            return dict(rmse=[], mae=[], r2=[])
        r2 = torch.where(self.tot > 0, 1 - self.res / self.tot, torch.zeros_like(self.tot))
        return dict(rmse=torch.sqrt(self.res / self.n).tolist(), mae=(self.abs / self.n).tolist(), r2=r2.tolist())
