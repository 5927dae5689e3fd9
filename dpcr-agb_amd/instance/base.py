"""Model <-> trainer contract of the reference, restated for the hot path:
``set_input(data, device)``, ``forward(epoch=)``, ``optimize_parameters(epoch, batch_size, num_batches)``,
``get_reg_output()``, ``get_reg_input()``, ``get_current_losses()``, ``loss_names``.

Follows torch_points3d/models/base_model.py:230-256 (train step: forward, backward, clip_grad_value_,
optimizer step, LR scheduler stepped per batch with a fractional epoch :219-226) and
torch_points3d/models/instance/base.py:54-208 (target standardisation buffers :86-114, output slice :139-146,
smooth-L1 on standardised targets weighted by mean(task weights) :154-179, de-standardised report :181-185).
Mixed precision: the reference's published sparse models run fp32; GradScaler is a no-op here.
"""
import os
from collections import OrderedDict
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

from ..optim import AdaBelief

REG_LOSSES = {"smoothl1": F.smooth_l1_loss, "l2": F.mse_loss, "l1": F.l1_loss}
OUT_ACT = {"linear": lambda x: x, "elu": F.elu, "relu": F.relu}


class InstanceBase(torch.nn.Module):
    def __init__(self, option, model_type, dataset, modules=None):
        super().__init__()
        self.opt = option
        self.loss_names: List[str] = []
        self.visual_names = ["data_visual"]
        self.output = None
        self.model = None
        self._conv_type = option.get("conv_type", None)
        self._optimizer = None
        self._lr_scheduler = None
        self._num_epochs = 0
        self._num_batches = 0
        self._num_samples = -1
        self._grad_clip = -1
        self._update_lr_scheduler_on = "on_epoch"
        self.grad_sync = None  # data-parallel hook: called between backward and clip/step

        self.loss_fns = {}
        self.has_reg_targets = dataset.has_reg_targets
        self.reg_targets_idx = dataset.reg_targets_idx
        if self.has_reg_targets:
            self.loss_names.append("loss_reg")
            self._register_target_stats(dataset)
            self.reg_out_act = OUT_ACT[option.get("reg_out_activation", "linear").lower()]
            self.reg_report_out_act = OUT_ACT[option.get("reg_out_report_activation", "linear").lower()]
            names = option.get("reg_loss_fn", "smoothl1")
            self.loss_fns["reg"] = [REG_LOSSES[n] for n in names.split(",")] if names else []
        self.num_reg_classes = dataset.num_reg_classes
        self.double_batch = option.get("double_batch", dataset.double_batch)
        if self.double_batch:
            raise NotImplementedError("double_batch (paired-sample batches) is not used by the NFI recipes and is not "
                                      "implemented")

    # ----------------------------------------------------------- target statistics (base.py:86-134)
    def _register_target_stats(self, dataset):
        n = int(sum(self.reg_targets_idx))
        center, scale, weights = np.zeros(n), np.ones(n), []
        i = 0
        for name in dataset.targets:
            t = dataset.targets[name]
            if t["task"] != "regression":
                continue
            weights.append(t.get("weight", 1))
            norm = t.get("normalization", "standard")
            if norm == "standard":
                center[i] = self._avg_stat(dataset, "mean", i)
                scale[i] = self._avg_stat(dataset, "std", i)
            elif norm == "min-max":
                center[i] = self._avg_stat(dataset, "min", i)
                scale[i] = self._avg_stat(dataset, "max", i) - center[i]
            center[i] = t.get("center_override", center[i])
            scale[i] = t.get("scale_override", scale[i])
            scale[i] *= t.get("scale_mult", 1.0)
            i += 1
        self.register_buffer("reg_scale_targets", torch.tensor(scale.reshape(1, -1), dtype=torch.float))
        self.register_buffer("reg_center_targets", torch.tensor(center.reshape(1, -1), dtype=torch.float))
        self.register_buffer("reg_weights", torch.tensor(weights, dtype=torch.float))

    @staticmethod
    def _avg_stat(dataset, stat, i):
        vals = np.array([np.asarray(area["train"])[i] for area in getattr(dataset, f"get_{stat}_targets")().values()
                         if "train" in area], dtype=np.float64)
        return float(np.nanmean(vals))

    def set_kernel_options(self, **kw):
        """Operand precision / kernel-choice knobs of THIS model (sparse_ops.KernelOptions fields, e.g.
        ``precision="bf16"``): carried by its backbone, applied to its forward pass and kept by its autograd nodes."""
        from ..sparse_ops import KernelOptions
        base = getattr(self.model, "kernel_options", None)
        opts = KernelOptions(base=base, **kw) if base is not None else KernelOptions(**kw)
        if opts.bf16_activations and not getattr(self.model, "supports_bf16_rows", False):
            # (bf16 ROW storage exists for the sparse ResNet / SENet backbones only: KPConv's gather / max-pool kernels and
            # the PointNet pooling take fp32 rows and would fail mid-network)
            raise ValueError(f"bf16_activations is implemented for the sparse ResNet/SENet backbones, not for "
                             f"{type(self.model).__name__}")
        self.model.kernel_options = opts
        return self.model.kernel_options

    # ----------------------------------------------------------- contract
    @property
    def conv_type(self):
        return self._conv_type

    def set_input(self, data, device):
        raise NotImplementedError

    def convert_outputs(self, outputs):
        if outputs is None or not self.has_reg_targets:
            return None
        return self.reg_out_act(outputs[:, :self.num_reg_classes])

    def compute_reg_loss(self):
        if not (self.has_reg_targets and self.loss_fns.get("reg")) or getattr(self, "reg_y", None) is None:
            return      # (no labels in this batch: a prediction-only forward pass)
        output = self.reg_out
        labels = (self.reg_y - self.reg_center_targets) / self.reg_scale_targets
        if self._reg_mask_all is False:
            if not bool(self.reg_y_mask.any()):
                return
            output, labels = output[self.reg_y_mask], labels[self.reg_y_mask]
        loss_reg = 0
        for fn in self.loss_fns["reg"]:
            loss_reg = loss_reg + fn(output, labels)
        d = self.__dict__      # (plain attributes: nn.Module.__setattr__ costs ~30 us per assignment)
        d["loss_reg"] = loss_reg
        d["loss"] = self.loss + self.reg_weights.mean() * loss_reg

    def compute_loss(self):
        self.__dict__["loss"] = 0
        self.compute_reg_loss()

    def get_reg_output(self):
        return self.reg_report_out_act(self.reg_out * self.reg_scale_targets + self.reg_center_targets)

    def get_reg_input(self):
        return self.reg_y

    def get_current_losses(self):
        out = OrderedDict()
        for name in self.loss_names:
            if hasattr(self, name):
                try:
                    out[name] = float(getattr(self, name))
                except Exception:
                    out[name] = None
        return out

    def get_parameter_list(self) -> List[dict]:
        return [{"params": list(self.parameters())}]

    # ----------------------------------------------------------- training objects (base_model.py:279-341)
    def init_train_objects(self, training):
        """training: Opt like config.TRAINING_NFI (optimizer class/params, lr_scheduler, grad_clip)."""
        opt = training.optim.optimizer
        name = opt.get("name", opt.get("class", "AdaBelief"))
        params = dict(opt.get("params", {}))
        cls = AdaBelief if name == "AdaBelief" else getattr(torch.optim, name)
        self._grad_clip = training.get("grad_clip", -1)
        if cls is AdaBelief and torch.cuda.is_available() and next(self.parameters()).is_cuda:
            # one fused HIP launch per parameter group, gradient clip folded in
            params.update(fused=True, clip_value=self._grad_clip if self._grad_clip > 0 else None)
        self._optimizer = cls(self.get_parameter_list(), **params)
        sch = training.get("lr_scheduler", None)
        if sch:
            self._update_lr_scheduler_on = sch.get("update_scheduler_on", "on_epoch")
            self._lr_scheduler = getattr(torch.optim.lr_scheduler, sch.name)(self._optimizer, **dict(sch.params))
        self._grad_clip = training.get("grad_clip", -1)

    @property
    def optimizer(self):
        return self._optimizer

    def _step_scheduler(self, epoch, batch_size, num_batches):
        if self._lr_scheduler is None:
            return
        mode = self._update_lr_scheduler_on
        if mode == "on_epoch":
            for _ in range(epoch - self._num_epochs):
                self._lr_scheduler.step(epoch)
        elif mode == "on_num_batch":
            self._lr_scheduler.step(self._num_batches / num_batches)
        elif mode == "on_num_sample":
            for _ in range(batch_size):
                self._lr_scheduler.step(epoch)

    def reserve_workspace(self, device, main_bytes=0, side_bytes=0, small_bytes=256 << 20):
        """Grow the caching allocator's pools of the compute stream and of the input pipeline's side stream up front (one
        big block each, handed straight back to the cache, which then carves every later request out of it): a
        hipMalloc in the middle of a step waits for the whole device, and the pools are per stream.  MI355X has 288 GB:
        a few GiB of head-room cost nothing.  Creates the side stream ``prefetch_input`` uses."""
        if not hasattr(self, "_side_stream"):
            self._side_stream = torch.cuda.Stream(device=device)
            self._staged = None
        for stream, nbytes in ((torch.cuda.current_stream(device), main_bytes), (self._side_stream, side_bytes)):
            if nbytes > 0:
                with torch.cuda.stream(stream):
                    block = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
                    del block
            if small_bytes > 0:
                # requests of up to 1 MiB come from a pool of their own (2 MiB segments) that the big block does not
                # feed: park the same head-room there (per-plot vectors, statistics, tile tables, drop-path rows)
                with torch.cuda.stream(stream):
                    blocks = [torch.empty(1 << 20, dtype=torch.uint8, device=device)
                              for _ in range(int(small_bytes) >> 20)]
                    del blocks
        torch.cuda.synchronize(device)

    @torch.no_grad()
    def calibrate_bn(self, batches, device, epochs=1):
        """Forward-only passes in train mode to refresh the BatchNorm running statistics — the reference's
        calibrate_bn.py flow (trainer.py:230-283: model.train() under torch.no_grad())."""
        was_training = self.training
        self.train()
        for _ in range(epochs):
            for data in batches:
                self.set_input(data, device)
                self.forward()
        self.train(was_training)

    @torch.no_grad()
    def evaluate(self, batches, device, target_mean=None):
        """eval.py flow (trainer.py:361-418): no-grad forward in eval mode, RMSE / MAE / R2 as the InstanceTracker
        defines them (metrics.py).  target_mean: mean of the stage's targets (default: of these batches)."""
        from ..metrics import RegressionMeter
        was_training = self.training
        self.eval()
        outs, ys, masks = [], [], []
        for data in batches:
            self.set_input(data, device)
            self.forward()
            outs.append(self.get_reg_output().detach().cpu())
            ys.append(self.get_reg_input().detach().cpu())
            masks.append(self.reg_y_mask.detach().cpu())
        self.train(was_training)
        outs, ys, masks = torch.cat(outs), torch.cat(ys), torch.cat(masks)
        # missing targets (masked out or NaN) are ignored per target, as the tracker does
        # (metrics/instance_tracker.py:116-134)
        valid = masks & ~torch.isnan(ys)
        if target_mean is None:
            yd = torch.where(valid, ys.double(), torch.zeros_like(ys, dtype=torch.float64))
            target_mean = yd.sum(0) / valid.sum(0).clamp(min=1)
        meter = RegressionMeter(target_mean)
        meter.add(outs, ys, valid)
        return meter.value()

    def poll_prefetch(self):
        """Input pipelines that read counts back asynchronously (KPConvModel) move on here; called between the phases of a step."""

    def optimize_parameters(self, epoch, batch_size, num_batches):
        self.poll_prefetch()
        self(epoch=epoch)
        self.poll_prefetch()
        # autograd assigns fresh gradient tensors (no zero-fill, no accumulate kernel per parameter); the data-parallel
        # hook packs them into its flat buckets with one multi-tensor copy per bucket (dist.GradAllReduce)
        self._optimizer.zero_grad(set_to_none=True)
        from ..sparse_ops import ZERO_ARENA
        ZERO_ARENA.new_step(self.loss.device)     # one fill for every zero-start gradient buffer of this backward pass
        self.loss.backward()
        self.poll_prefetch()
        if self.grad_sync is not None:
            self.grad_sync()
        if self._grad_clip > 0 and not getattr(self._optimizer, "fused", False):
            torch.nn.utils.clip_grad_value_(self.parameters(), self._grad_clip)
        self._optimizer.step()
        self._step_scheduler(epoch, batch_size, num_batches)
        self._num_epochs = epoch
        self._num_batches += 1
        self._num_samples += batch_size
        self.poll_prefetch()
        self._pace_host()
        self.poll_prefetch()

    # The host enqueues a step in about half the time the device needs for it and would run ahead until the hardware
    # queue is full — where the runtime SPINS for a free slot: one core per rank burnt for nothing, and under a CPU quota
    # shared by eight ranks a reason to be throttled.  Instead the host sleeps on a blocking event (interrupt-driven wait)
    # until the step issued PACE_DEPTH steps ago is done: the device queue always holds that many whole steps.
    PACE_DEPTH = int(os.environ.get("AGB_PACE_DEPTH", "3"))      # 0: no pacing (the host runs ahead until the queue is full)

    def _pace_host(self):
        if not torch.cuda.is_available() or self.PACE_DEPTH <= 0:
            return
        ring = self.__dict__.setdefault("_pace_ring", [])
        ev = torch.cuda.Event(blocking=True)
        ev.record()
        held = self.__dict__.get("_held_inputs")
        ring.append((ev, held))      # everything consumed on the compute stream so far is finished when ev is
        self.__dict__["_held_inputs"] = []
        if len(ring) > self.PACE_DEPTH:
            old, _inputs = ring.pop(0)
            old.synchronize()         # (_inputs dropped here: their memory returns to the side stream's pool)

    # Tensors built on the side stream and consumed on the compute stream.  ``Tensor.record_stream`` would make the caching
    # allocator record an event per freed block and poll it with hipEventQuery — and every query of an unfinished event
    # makes the runtime enqueue a marker with a completion callback on the other queue: ~100 callbacks per step kept the
    # runtime's signal-handler thread spinning (8.5-9.5 ms of CPU per 8.9 ms step, profiles/r03_host_cpu.txt).  Instead
    # the step keeps a reference until the pacing event recorded after it has been waited for.
    HOLD_LIMIT = 8

    def _hold_input(self, obj, tensors, stream):
        if self.PACE_DEPTH <= 0 or os.environ.get("AGB_INPUT_RECORD_STREAM", "0") != "0":
            for t in tensors:
                t.record_stream(stream)
            return
        held = self.__dict__.setdefault("_held_inputs", [])
        held.append((obj, list(tensors)))
        # a loop that never calls optimize_parameters (evaluation, calibrate_bn) has no pacing event to release on: it keeps
        # two batches (the one in flight and the one being built), the rest goes the allocator's way
        if len(held) > (self.HOLD_LIMIT if self.training and torch.is_grad_enabled() else 2):
            for t in held.pop(0)[1]:
                t.record_stream(stream)
