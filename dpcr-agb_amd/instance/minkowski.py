"""``MinkowskiBaselineModel`` — the reference's wrapper for MSENet14/50 and MPointNet
(torch_points3d/models/instance/minkowski.py:15-89): builds the backbone through
``initialize_minkowski_unet`` (:32-38), replaces ``model.final`` by one ``Linear(C, 1)`` per regression target
(:39-46), ``set_input`` builds ``SparseTensor(features, [batch,x,y,z] int32)`` (:67-80), head/backbone
parameter groups (:54-65).
"""
from typing import List

import torch

from .. import me_compat as ME
from ..backbones import initialize_minkowski_unet
from .base import InstanceBase


class SeparateLinear(torch.nn.Module):
    def __init__(self, in_channel, num_reg_classes):
        super().__init__()
        self.linears = torch.nn.ModuleList([torch.nn.Linear(in_channel, 1, bias=True)
                                            for _ in range(max(num_reg_classes, 0))])

    def forward(self, x):
        return torch.cat([lin(x.F) for lin in self.linears], 1)


class MinkowskiBaselineModel(InstanceBase):
    def __init__(self, option, model_type, dataset, modules=None):
        super().__init__(option, model_type, dataset, modules)
        self.model = initialize_minkowski_unet(
            option.model_name, dataset.feature_dimension, dataset.num_classes, activation=option.activation,
            first_stride=option.first_stride, global_pool=option.global_pool, bias=option.get("bias", True),
            bn_momentum=option.get("bn_momentum", 0.1), norm_type=option.get("norm_type", "bn"),
            dropout=option.get("dropout", 0.0), drop_path=option.get("drop_path", 0.0),
            **option.get("extra_options", {}))
        in_channel = self.model.final.linear.weight.shape[1]
        self.model.final = SeparateLinear(in_channel, self.num_reg_classes)
        for m in self.model.final.linears:
            torch.nn.init.trunc_normal_(m.weight, std=0.02)
            torch.nn.init.constant_(m.bias, 0)
        self.head_namespace = option.get("head_namespace", "final.linears")
        self.head_optim_settings = option.get("head_optim_settings", {})
        self.backbone_optim_settings = option.get("backbone_optim_settings", {})
        self.add_pos = option.get("add_pos", False)

    def get_parameter_list(self) -> List[dict]:
        head, backbone = [], []
        for name, p in self.model.named_parameters():
            (head if self.head_namespace in name else backbone).append(p)
        return [{"params": head, **self.head_optim_settings}, {"params": backbone, **self.backbone_optim_settings}]

    def _stage_a(self, data, device, defer):
        """SparseTensor + coordinate levels of a batch (device-side counts; host read-back deferred if asked)."""
        coords = torch.cat([data.batch.unsqueeze(-1).int(), data.coords.int()], -1)
        features = data.x
        if self.add_pos:
            features = torch.cat([data.pos, features], 1)
        inp = ME.SparseTensor(features=features, coordinates=coords, device=device, batch_size=len(data),
                              bounds=getattr(data, "coord_bounds", None))
        strides = getattr(self.model, "tensor_strides", None)
        if strides:
            # whole coordinate pyramid in one go: a single host read-back per batch instead of one per level
            inp.coordinate_manager.prefetch_strides(strides, defer=defer)
        return inp

    def _stage_b(self, inp):
        """Kernel maps of every layer of the model for a staged batch."""
        cm = inp.coordinate_manager
        cm.finish_prefetch()
        if hasattr(self.model, "plan_spec"):
            cm.prebuild(self.model.plan_spec(input_requires_grad=False), getattr(self.model, "kernel_options", None))
        return inp

    def _build_input(self, data, device):
        return self._stage_b(self._stage_a(data, device, defer=False))

    def input_stream(self, device):
        """The side stream the input pipeline runs on (created on first use): a caller that BUILDS its batches on the device
        (``train_transforms.SparseTrainPipeline``) does so under ``torch.cuda.stream(model.input_stream(device))`` before
        ``prefetch_input``; ``set_input`` keeps such a batch alive until the step that consumed it is done."""
        if not hasattr(self, "_side_stream"):
            self._side_stream = torch.cuda.Stream(device=device)
            self._staged = None
        return self._side_stream

    def prefetch_input(self, data, device):
        """Input pipeline on a side stream, two batches deep, with no host wait: this call enqueues stage A of `data`
        (coordinate levels; the row counts travel to pinned memory asynchronously) and stage B (kernel maps) of the
        batch staged by the PREVIOUS call, whose counts landed a whole step ago.  ``set_input`` picks a batch up at
        whatever stage it is in.  Call it after ``optimize_parameters`` with the batch two steps ahead (or one step
        ahead: stage B then runs inside ``set_input``, still on the side stream)."""
        side = self.input_stream(device)
        # NOTE: no wait on the compute stream here (that would serialise the plan behind the whole running step):
        # the batch tensors must already be materialised (data-loader output / device-resident pool).
        with torch.cuda.stream(side):
            prev = self._staged
            if prev is not None and prev is not data and getattr(prev, "_prefetched", None) is not None \
                    and prev._prefetched[2] == "A":
                inp_prev = self._stage_b(prev._prefetched[0])
                prev._prefetched = (inp_prev, side.record_event(), "B")
            inp = self._stage_a(data, device, defer=True)
            data._prefetched = (inp, side.record_event(), "A")
            self._staged = data

    def set_input(self, data, device):
        d = self.__dict__     # (plain attributes: nn.Module.__setattr__ costs ~20 us per tensor assignment)
        d["batch_idx"] = data.batch.squeeze()
        d["data_visual"] = data
        pre = getattr(data, "_prefetched", None)
        if pre is not None:
            inp, ev, stage = pre
            data._prefetched = None
            if getattr(self, "_staged", None) is data:
                self._staged = None
            if stage == "A":   # staged only one step ahead: finish on the side stream now
                with torch.cuda.stream(self._side_stream):
                    inp = self._stage_b(inp)
                    ev = self._side_stream.record_event()
            d["input"] = inp
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            # (data too: a batch built on the side stream — its targets are read by the loss on the compute stream)
            self._hold_input((inp, data), list(inp.coordinate_manager.tensors()) + [inp.F] +
                             [t for t in getattr(data, "__dict__", {}).values() if isinstance(t, torch.Tensor) and t.is_cuda], cur)
        else:
            d["input"] = self._build_input(data, device)
        d["reg_y"] = None          # (a batch without labels — prediction from a file: forward() then skips the loss)
        if len(self.loss_fns) > 0:
            bs = len(data)
            if self.has_reg_targets and getattr(data, "y_reg", None) is not None:
                mask_all = getattr(data, "y_reg_mask_all", None)  # host-side flag: no device sync in the loss
                d["_reg_mask_all"] = bool(data.y_reg_mask.all()) if mask_all is None else mask_all
                d["reg_y_mask"] = data.y_reg_mask.to(device, non_blocking=True).view(bs, -1)
                d["reg_y"] = data.y_reg.to(device, non_blocking=True).view(bs, -1)

    def _fused_head(self):
        """Loss-function mask of the one-launch head + loss (head_ops.py, csrc/head.hip) when this model / batch is one it
        takes — the plain configuration of the reference (linear output activation, smooth-L1 / L2 / L1, every target
        present) on a backbone with ``forward_features`` — else None."""
        from ..head_ops import MAX_TARGETS, loss_mask
        from ..sparse_ops import current
        m = self.model
        opts = getattr(m, "kernel_options", None) or current()
        if not (getattr(opts, "fused_head", True) and hasattr(m, "forward_features") and self.has_reg_targets
                and isinstance(m.final, SeparateLinear) and self.opt.get("reg_out_activation", "linear").lower() == "linear"
                and self.__dict__.get("reg_y") is not None and self.__dict__.get("_reg_mask_all") is True
                and 1 <= len(m.final.linears) <= MAX_TARGETS and len(m.final.linears) == self.num_reg_classes):
            return None
        return loss_mask(self.loss_fns.get("reg") or [])

    def forward(self, *args, **kwargs):
        mask = self._fused_head() if len(self.loss_fns) > 0 else None
        if mask is not None:
            pooled = self.model.forward_features(self.input).F
            if pooled.is_cuda and pooled.dtype == torch.float32 and self.reg_y.dtype == torch.float32:
                from ..head_ops import reg_head_loss
                d = self.__dict__      # (plain attributes: nn.Module.__setattr__ costs ~20 us per tensor assignment)
                d["output"], d["loss_reg"], d["loss"] = reg_head_loss(
                    pooled, self.model.final.linears, self.reg_y, self.reg_center_targets, self.reg_scale_targets,
                    self.reg_weights, mask)
                d["reg_out"] = self.output
                return
            self.output = torch.cat([lin(pooled) for lin in self.model.final.linears], 1)
        else:
            self.output = self.model(self.input)
        self.reg_out = self.convert_outputs(self.output)
        self.compute_loss()
