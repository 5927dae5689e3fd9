"""KPConv backbone (KPCNN) on the HIP kernels — same configuration attributes, ``forward(batch)`` contract and
state_dict keys as the reference (torch_points3d/modules/KPConv/architectures.py:67-151; blocks.py:414-738):

    block_ops.<i>.KPConv.{weights [K,Cin,Cout], kernel_points [K,3]}
    block_ops.<i>.{batch_norm | batch_norm_conv}.batch_norm.{weight,bias,running_mean,running_var,num_batches_tracked}
    block_ops.<i>.{unary1,unary2,unary_shortcut}.{mlp.weight, batch_norm.batch_norm.*}
    head_mlp.{mlp.weight, batch_norm.bias}

``batch`` carries ``features [N0,F]`` and the per-level lists ``points``, ``neighbors``, ``pools``, ``lengths``
(models/instance/kpconv.py:252-264).  Only the rigid (non-deformable) convolution with linear influence and sum
aggregation is implemented — what the AGB configuration uses (conf/models/instance/kpconv.yaml:15-75).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..kp_dispositions import kernel_disposition
from ..kpconv_ops import KPConvFusedFunction, KPConvSymmetricFunction, KPGatherFunction, KPMaxPoolFunction, as_index
from ..norm_ops import ACT_IDS, AddActFunction, batch_norm_act, batch_norm_add_act
from ..sparse_ops import DenseConvFunction, current as current_options, dense_linear, dense_linear_join, model_scope, \
    segment_reduce, take_bn_hint

ACTIVATION_NAMES = {"relu": "relu", "gelu": "gelu"}
# (Linear -> BatchNorm -> + shortcut -> activation of the bottleneck blocks as one node: KernelOptions.fused_tail)


def _act_module(name):
    return {"relu": nn.ReLU, "gelu": nn.GELU, "silu": nn.SiLU, "swish": nn.SiLU, "sigmoid": nn.Sigmoid,
            "tanh": nn.Tanh}[name]()


class KPConv(nn.Module):
    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius, fixed_kernel_points="center",
                 KP_influence="linear", aggregation_mode="sum", deformable=False, modulated=False):
        super().__init__()
        if deformable or modulated:
            raise NotImplementedError("deformable / modulated KPConv is not used by the AGB configuration")
        if KP_influence != "linear" or aggregation_mode != "sum":
            raise NotImplementedError("only KP_influence='linear' with aggregation_mode='sum' is implemented")
        self.K, self.p_dim = kernel_size, p_dim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.radius, self.KP_extent = radius, KP_extent
        self.weights = nn.Parameter(torch.zeros(kernel_size, in_channels, out_channels, dtype=torch.float32))
        nn.init.kaiming_uniform_(self.weights, a=math.sqrt(5))
        kp = kernel_disposition(radius, kernel_size, dimension=p_dim, fixed=fixed_kernel_points)
        self.kernel_points = nn.Parameter(torch.tensor(kp, dtype=torch.float32), requires_grad=False)

    def forward(self, q_pts, s_pts, neighb_inds, x):
        idx = as_index(neighb_inds)
        # same point set on both sides with a symmetric neighbour relation (marked by the input pyramid on uncropped
        # self-searches): the scatter-free backward
        symmetric = q_pts is s_pts and getattr(neighb_inds, "agb_symmetric", False) and x.is_cuda
        if symmetric and KPConvFusedFunction.supported(self.K, self.in_channels, self.out_channels, idx, current_options()):
            # gather + contraction in one kernel per direction (csrc/kpfused.hip)
            return KPConvFusedFunction.apply(x, q_pts, idx, self.kernel_points, self.KP_extent, self.weights)
        if symmetric and KPConvSymmetricFunction.supported(self.K, self.in_channels, self.out_channels):
            return take_bn_hint(KPConvSymmetricFunction.apply(x, q_pts, idx, self.kernel_points, self.KP_extent,
                                                              self.weights))
        wf = KPGatherFunction.apply(x, q_pts, s_pts, idx, self.kernel_points, self.KP_extent)
        # dense feature x kernel-weight contraction [N, K*Cin] @ [K*Cin, Cout] on the library's own MFMA kernels
        w2d = self.weights.view(-1, self.out_channels)
        if wf.is_cuda and DenseConvFunction.supported(w2d.shape[0], w2d.shape[1]):
            return take_bn_hint(DenseConvFunction.apply(wf.view(wf.shape[0], -1), w2d, None))
        # odd widths (the 3-feature input layer: K * Cin = 45): the zero-padding Linear form of the same kernels
        return dense_linear(wf.view(wf.shape[0], -1), w2d.t().contiguous())

    def __repr__(self):
        return f"KPConv(radius: {self.radius:.2f}, in_feat: {self.in_channels:d}, out_feat: {self.out_channels:d})"


class BatchNormBlock(nn.Module):
    def __init__(self, in_dim, use_bn, bn_momentum):
        super().__init__()
        self.bn_momentum, self.use_bn, self.in_dim = bn_momentum, use_bn, in_dim
        if use_bn:
            self.batch_norm = nn.BatchNorm1d(in_dim, momentum=bn_momentum)
        else:
            self.bias = nn.Parameter(torch.zeros(in_dim, dtype=torch.float32))

    def forward(self, x, act=None):
        """act: fused activation name ('relu'/'gelu') or None."""
        if self.use_bn:
            if x.shape[1] % 4 == 0:
                return batch_norm_act(x, self.batch_norm, act)
            y = self.batch_norm(x)
        else:
            y = x + self.bias
        if act == "relu":
            return torch.relu(y)
        if act == "gelu":
            return torch.nn.functional.gelu(y)
        return y


class UnaryBlock(nn.Module):
    def __init__(self, in_dim, out_dim, act_name, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.in_dim, self.out_dim, self.no_relu = in_dim, out_dim, no_relu
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self.batch_norm = BatchNormBlock(out_dim, use_bn, bn_momentum)
        self.act_name = act_name
        self.act = None if (no_relu or act_name in ACTIVATION_NAMES) else _act_module(act_name)

    def forward(self, x, batch=None):
        return self._norm_act(dense_linear(x, self.mlp.weight, self.mlp.bias))

    def forward_join(self, x):
        """(block output, x for the branch that bypasses this block): sparse_ops.dense_linear_join."""
        z, branch = dense_linear_join(x, self.mlp.weight, self.mlp.bias)
        return self._norm_act(z), branch

    def _norm_act(self, x):
        if self.no_relu:
            return self.batch_norm(x)
        if self.act is None:
            return self.batch_norm(x, ACTIVATION_NAMES[self.act_name])
        return self.act(self.batch_norm(x))


def _geometry(block, batch):
    li = block.layer_ind
    if "strided" in block.block_name:
        return batch.points[li + 1], batch.points[li], batch.pools[li]
    return batch.points[li], batch.points[li], batch.neighbors[li]


def _make_conv(config, in_dim, out_dim, radius):
    extent = radius * config.KP_extent / config.conv_radius
    return KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim, extent, radius,
                  fixed_kernel_points=config.fixed_kernel_points, KP_influence=config.KP_influence,
                  aggregation_mode=config.aggregation_mode, deformable=False,
                  modulated=getattr(config, "modulated", False))


def _post(bn_block, act_name, act_mod, x):
    if act_mod is None:
        return bn_block(x, ACTIVATION_NAMES[act_name])
    return act_mod(bn_block(x))


class SimpleBlock(nn.Module):
    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, act_name, config):
        super().__init__()
        if "deform" in block_name:
            raise NotImplementedError("deformable blocks are not used by the AGB configuration")
        self.block_name, self.layer_ind, self.in_dim, self.out_dim = block_name, layer_ind, in_dim, out_dim
        self.KPConv = _make_conv(config, in_dim, out_dim // 2, radius)
        self.batch_norm = BatchNormBlock(out_dim // 2, config.use_batch_norm, config.batch_norm_momentum)
        self.act_name = act_name
        self.act = None if act_name in ACTIVATION_NAMES else _act_module(act_name)

    def forward(self, x, batch):
        q, s, idx = _geometry(self, batch)
        return _post(self.batch_norm, self.act_name, self.act, self.KPConv(q, s, idx, x))


class ResnetBottleneckBlock(nn.Module):
    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, act_name, config):
        super().__init__()
        if "deform" in block_name:
            raise NotImplementedError("deformable blocks are not used by the AGB configuration")
        self.block_name, self.layer_ind, self.in_dim, self.out_dim = block_name, layer_ind, in_dim, out_dim
        bn, mom = config.use_batch_norm, config.batch_norm_momentum
        self.unary1 = UnaryBlock(in_dim, out_dim // 4, act_name, bn, mom) if in_dim != out_dim // 4 else nn.Identity()
        self.KPConv = _make_conv(config, out_dim // 4, out_dim // 4, radius)
        self.batch_norm_conv = BatchNormBlock(out_dim // 4, bn, mom)
        self.unary2 = UnaryBlock(out_dim // 4, out_dim, act_name, bn, mom, no_relu=True)
        self.unary_shortcut = UnaryBlock(in_dim, out_dim, act_name, bn, mom, no_relu=True) if in_dim != out_dim \
            else nn.Identity()
        self.act_name = act_name
        self.act = _act_module(act_name)
        self._fused_act = None if act_name not in ACTIVATION_NAMES else act_name

    def forward(self, features, batch):
        q, s, idx = _geometry(self, batch)
        if isinstance(self.unary1, UnaryBlock):
            # the block input feeds unary1 and the shortcut: the shortcut's gradient joins in unary1's data-gradient kernel
            x, features = self.unary1.forward_join(features)
        else:
            x = self.unary1(features)
        x = self.KPConv(q, s, idx, x)
        x = _post(self.batch_norm_conv, self.act_name, None if self._fused_act else self.act, x)
        if "strided" in self.block_name:
            shortcut = KPMaxPoolFunction.apply(features, as_index(idx))
        else:
            shortcut = features
        shortcut = self.unary_shortcut(shortcut)
        u2 = self.unary2
        if (current_options().fused_tail and self._fused_act and x.is_cuda and u2.no_relu and u2.batch_norm.use_bn
                and u2.out_dim % 4 == 0):
            # Linear -> BatchNorm -> (+ shortcut) -> activation without the BatchNorm output in memory
            z = dense_linear(x, u2.mlp.weight, u2.mlp.bias)
            return batch_norm_add_act(z, shortcut, u2.batch_norm.batch_norm, self._fused_act)
        x = u2(x)
        if self._fused_act and x.shape[1] % 4 == 0:
            return AddActFunction.apply(x, shortcut, None, None, ACT_IDS[self._fused_act])
        return self.act(x + shortcut)


def _ptr_from_lengths(lengths, device):
    l = lengths.detach().to("cpu", torch.int64) if isinstance(lengths, torch.Tensor) else torch.as_tensor(lengths)
    p = torch.zeros(len(l) + 1, dtype=torch.int32)
    p[1:] = torch.cumsum(l, 0).to(torch.int32)
    return p.to(device)


class _GlobalReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ptr, B, mode_id):
        x = x.contiguous()
        y, _ = segment_reduce(x, None, ptr, B, mode_id)
        ctx.save_for_backward(ptr)
        ctx.cfg = (x.shape[0], B, mode_id)
        return y

    @staticmethod
    def backward(ctx, dy):
        (ptr,) = ctx.saved_tensors
        n, B, mode_id = ctx.cfg
        lens = (ptr[1:] - ptr[:-1]).long()
        g = dy if mode_id == 0 else dy / lens.clamp(min=1).unsqueeze(1).to(dy.dtype)
        return torch.repeat_interleave(g, lens, dim=0, output_size=n), None, None, None


class GlobalSumBlock(nn.Module):
    MODE = 0

    def forward(self, x, batch):
        lengths = batch.lengths[-1]
        ptr = getattr(batch, "last_ptr", None)
        if ptr is None:
            ptr = _ptr_from_lengths(lengths, x.device)
        return _GlobalReduce.apply(x, ptr, len(lengths), self.MODE)


class GlobalAverageBlock(GlobalSumBlock):
    MODE = 1


class MaxPoolBlock(nn.Module):
    def __init__(self, layer_ind):
        super().__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return KPMaxPoolFunction.apply(x, as_index(batch.pools[self.layer_ind + 1]))


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, act_name, config):
    if block_name == "unary":
        return UnaryBlock(in_dim, out_dim, act_name, config.use_batch_norm, config.batch_norm_momentum)
    if block_name.startswith("simple"):
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, act_name, config)
    if block_name.startswith("resnetb"):
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, act_name, config)
    if block_name in ("max_pool", "max_pool_wide"):
        return MaxPoolBlock(layer_ind)
    if block_name == "global_average":
        return GlobalAverageBlock()
    if block_name == "global_sum":
        return GlobalSumBlock()
    raise ValueError("Unknown block name in the architecture definition : " + block_name)


class KPCNN(nn.Module):
    def __init__(self, config):
        super().__init__()
        layer = 0
        r = config.first_subsampling_dl * config.conv_radius
        in_dim, out_dim = config.in_features_dim, config.first_features_dim
        self.K = config.num_kernel_points
        self.block_ops = nn.ModuleList()
        for block in config.architecture:
            if "equivariant" in block and out_dim % 3 != 0:
                raise ValueError("Equivariant block but features dimension is not a factor of 3")
            if "upsample" in block:
                break
            self.block_ops.append(block_decider(block, r, in_dim, out_dim, layer, config.activation, config))
            in_dim = out_dim // 2 if "simple" in block else out_dim
            if "pool" in block or "strided" in block:
                layer += 1
                r *= 2
                out_dim *= 2
        self.head_mlp = UnaryBlock(out_dim, 1024, config.activation, False, 0)

    kernel_options = None      # sparse_ops.KernelOptions of this model (None: the ones in force / the defaults)

    def forward(self, batch):
        with model_scope(self):
            x = batch.features.clone().detach()
            if x.is_cuda and not current_options().low_precision:
                # W^T of all unary blocks' Linear layers from one launch per optimiser step (fused_blocks.LinearTransposes)
                from ..fused_blocks import linear_transposes
                lt = linear_transposes(self, lambda: [m for m in self.modules() if isinstance(m, nn.Linear)])
                if lt is not None:
                    lt.ensure()
            for op in self.block_ops:
                x = op(x, batch)
            return self.head_mlp(x, batch)


__all__ = ["KPConv", "KPCNN", "UnaryBlock", "BatchNormBlock", "SimpleBlock", "ResnetBottleneckBlock",
           "GlobalSumBlock", "GlobalAverageBlock", "MaxPoolBlock", "block_decider"]
