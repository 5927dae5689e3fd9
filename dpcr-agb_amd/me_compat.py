"""The subset of the MinkowskiEngine Python API that the reference's sparse-voxel backbones use
(torch_points3d/modules/MinkowskiEngine/{SENet,PointNet,resnet_block,senet_block,common}.py and
torch_points3d/models/instance/minkowski.py:67-80), implemented on libagbhip's HIP kernels.

Usage mirrors the reference:  ``import dpcr_agb_amd.me_compat as ME``  then ``ME.SparseTensor(features=...,
coordinates=int32[N,1+3], device=...)``, ``ME.MinkowskiConvolution(...)`` etc.  Semantics follow ME 0.5.x's
documented behaviour (the engine itself is not vendored by the reference: parity is pinned by this repo's own
known-answer tests, see DESIGN.md "parity unpinned"):
  * output coordinates of a stride-s op: unique(floor(c / (s*ts)) * s*ts)
  * kernel offsets: odd k -> {-(k//2) .. k//2} * ts * dilation, even k -> {0 .. k-1}; x fastest
  * kernel [K^3, Cin, Cout] ([Cin, Cout] when K^3 == 1 and stride == 1), bias [1, Cout]
  * max pooling over present inputs only; global pooling per batch index; MinkowskiGlobalPooling = average
"""
import enum
import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from .coords import CoordinateManager, CoordinateMapKey, _as_int
from .norm_ops import ACT_IDS, AddActFunction, batch_norm_act
from .sparse_ops import (BroadcastMulFunction, DenseConvFunction, take_bn_hint, GlobalPoolFunction, MaxPoolFunction,
                         SparseConvFunction, dense_conv_join, dense_linear)


class SparseTensor:
    def __init__(self, features: torch.Tensor, coordinates: Optional[torch.Tensor] = None,
                 coordinate_map_key: Optional[CoordinateMapKey] = None,
                 coordinate_manager: Optional[CoordinateManager] = None, tensor_stride: int = 1, device=None,
                 batch_size: Optional[int] = None, bounds=None, coordinate_mode: str = "auto", **kwargs):
        if coordinates is not None:
            if device is None:
                device = features.device
            ts = _as_int(tensor_stride)
            coordinate_manager = CoordinateManager(coordinates, device=device, tensor_stride=ts,
                                                   batch_size=batch_size, bounds=bounds, mode=coordinate_mode)
            coordinate_map_key = CoordinateMapKey(ts)
            features = features.to(device=coordinate_manager.device, dtype=torch.float32, non_blocking=True)
        if coordinate_manager is None or coordinate_map_key is None:
            raise ValueError("SparseTensor needs either coordinates or (coordinate_map_key, coordinate_manager)")
        n = coordinate_manager.num_rows(coordinate_map_key)
        if features.shape[0] != n:
            raise ValueError(f"features have {features.shape[0]} rows but the coordinate map has {n}")
        self._F = features
        self.coordinate_map_key = coordinate_map_key
        self.coordinate_manager = coordinate_manager

    # --- ME attribute surface used by the reference
    @property
    def F(self):
        return self._F

    @property
    def features(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.coords_of(self.coordinate_map_key)

    @property
    def coordinates(self):
        return self.C

    @property
    def tensor_stride(self):
        return [self.coordinate_map_key.tensor_stride] * 3

    @property
    def device(self):
        return self._F.device

    @property
    def dtype(self):
        return self._F.dtype

    @property
    def shape(self):
        return self._F.shape

    @property
    def D(self):
        return 3

    def size(self):
        return self._F.size()

    def __len__(self):
        return self._F.shape[0]

    @property
    def _ts(self):
        return self.coordinate_map_key.tensor_stride

    def _ptr_list(self) -> List[int]:
        cm = self.coordinate_manager
        if self._ts == 0:
            return list(range(cm.batch_size + 1))
        return cm.batch_ptr(self._ts).tolist()

    @property
    def decomposed_coordinates(self):
        p, c = self._ptr_list(), self.C
        return [c[p[b]:p[b + 1], 1:] for b in range(len(p) - 1)]

    @property
    def decomposed_features(self):
        p = self._ptr_list()
        return [self._F[p[b]:p[b + 1]] for b in range(len(p) - 1)]

    def _like(self, feats):
        return SparseTensor(feats, coordinate_map_key=self.coordinate_map_key,
                            coordinate_manager=self.coordinate_manager)

    def _check_same_map(self, other):
        if not isinstance(other, SparseTensor):
            return
        if other.coordinate_manager is not self.coordinate_manager or \
                other.coordinate_map_key != self.coordinate_map_key:
            raise ValueError("binary operations need both SparseTensors on the same coordinate map")

    def __add__(self, other):
        self._check_same_map(other)
        return self._like(self._F + (other._F if isinstance(other, SparseTensor) else other))

    def __sub__(self, other):
        self._check_same_map(other)
        return self._like(self._F - (other._F if isinstance(other, SparseTensor) else other))

    def __mul__(self, other):
        self._check_same_map(other)
        return self._like(self._F * (other._F if isinstance(other, SparseTensor) else other))

    def __repr__(self):
        return f"SparseTensor(F={tuple(self._F.shape)}, tensor_stride={self._ts}, device={self.device})"


# --------------------------------------------------------------------------- convolution / pooling
class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, dimension=None, **kwargs):
        super().__init__()
        if dimension is not None and dimension != 3:
            raise NotImplementedError("only D=3 sparse convolutions are implemented")
        if expand_coordinates:
            raise NotImplementedError("expand_coordinates is not supported")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = _as_int(kernel_size), _as_int(stride), _as_int(dilation)
        if self.kernel_size < 1:
            raise ValueError("kernel_size must be >= 1")
        self.kernel_volume = self.kernel_size ** 3
        self.use_mm = self.kernel_volume == 1 and self.stride == 1
        shape = (in_channels, out_channels) if self.use_mm else (self.kernel_volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(*shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty(1, out_channels, dtype=torch.float32)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward_join(self, input: SparseTensor):
        """(self(input), input for the branch that bypasses this layer): for the 1x1 first convolution of a bottleneck block,
        whose input also feeds the shortcut — the shortcut's gradient then joins in this layer's data-gradient kernel
        (sparse_ops.dense_conv_join).  Any other layer: (self(input), input)."""
        if self.use_mm and input.F.is_cuda and DenseConvFunction.supported(self.in_channels, self.out_channels):
            y, branch = dense_conv_join(input.F, self.kernel, self.bias)
            return input._like(y), (input if branch is input.F else input._like(branch))
        return self.forward(input), input

    def forward(self, input: SparseTensor, coordinates=None) -> SparseTensor:
        if coordinates is not None:
            raise NotImplementedError("explicit output coordinates are not supported")
        cm, ts = input.coordinate_manager, input._ts
        if self.use_mm:
            if input.F.is_cuda and DenseConvFunction.supported(self.in_channels, self.out_channels):
                # 1x1 stride-1: this library's own MFMA kernels on the identity map (csrc/spconv.hip), not a BLAS call
                return input._like(take_bn_hint(DenseConvFunction.apply(input.F, self.kernel, self.bias)))
            out = input.F @ self.kernel     # odd channel counts (not used by the NFI models)
            if self.bias is not None:
                out = out + self.bias
            return input._like(out)
        ts_out = ts * self.stride
        needs_dx = input.F.requires_grad and torch.is_grad_enabled()
        if self.in_channels == 3 and not needs_dx and ("fwd", ts, self.kernel_size, self.stride, self.dilation) \
                not in cm.kernel_maps:
            # three-channel stem: the forward kernel probes the level's dense grid itself and writes the kernel map out
            # for its weight gradient (no separate 7^3 map pass)
            probe = cm.grid_probe(ts, self.kernel_size, self.stride, self.dilation)
            if probe is not None:
                n = cm.level(ts).n
                out = SparseConvFunction.apply(input.F, self.kernel, self.bias, None, None, n, n, None,
                                               (probe, self.kernel_size))
                return SparseTensor(out, coordinate_map_key=CoordinateMapKey(ts_out), coordinate_manager=cm)
        nbr = cm.kernel_map(ts, self.kernel_size, self.stride, self.dilation)
        nbrT = plan = None
        if needs_dx and not (self.stride == 1 and self.kernel_size % 2 == 1):
            nbrT = cm.transposed_map(ts, self.kernel_size, self.stride, self.dilation)
            if self.stride > 1:
                plan = cm.transposed_plan(ts, self.kernel_size, self.stride, self.dilation)
        n_in, n_out = cm.level(ts).n, cm.level(ts_out).n
        out = SparseConvFunction.apply(input.F, self.kernel, self.bias, nbr, nbrT, n_in, n_out, plan)
        return SparseTensor(out, coordinate_map_key=CoordinateMapKey(ts_out), coordinate_manager=cm)

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, dilation={self.dilation}")


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=None, **kwargs):
        super().__init__()
        self.kernel_size, self.stride, self.dilation = _as_int(kernel_size), _as_int(stride), _as_int(dilation)

    def forward(self, input: SparseTensor, coordinates=None) -> SparseTensor:
        cm, ts = input.coordinate_manager, input._ts
        nbr = cm.kernel_map(ts, self.kernel_size, self.stride, self.dilation)
        ts_out = ts * self.stride
        if input.F.requires_grad and torch.is_grad_enabled():
            nbrT = cm.transposed_map(ts, self.kernel_size, self.stride, self.dilation)
        else:
            nbrT = nbr  # placeholder, never used without a backward pass
        n_in, n_out = cm.level(ts).n, cm.level(ts_out).n
        out = MaxPoolFunction.apply(input.F, nbr, nbrT, n_in, n_out)
        return SparseTensor(out, coordinate_map_key=CoordinateMapKey(ts_out), coordinate_manager=cm)


class _GlobalPoolBase(nn.Module):
    MODE = "sum"

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, input: SparseTensor) -> SparseTensor:
        cm, ts = input.coordinate_manager, input._ts
        if ts == 0:
            return input
        lvl = cm.level(ts)
        out = GlobalPoolFunction.apply(input.F, lvl.coords, cm.batch_ptr(ts), cm.batch_size, self.MODE)
        return SparseTensor(out, coordinate_map_key=CoordinateMapKey(0), coordinate_manager=cm)


class MinkowskiGlobalSumPooling(_GlobalPoolBase):
    MODE = "sum"


class MinkowskiGlobalAvgPooling(_GlobalPoolBase):
    MODE = "avg"


class MinkowskiGlobalMaxPooling(_GlobalPoolBase):
    MODE = "max"


class MinkowskiGlobalPooling(MinkowskiGlobalAvgPooling):
    """ME 0.5.x keeps this name as the average pooling."""


class MinkowskiBroadcastMultiplication(nn.Module):
    def forward(self, input: SparseTensor, input_glob: SparseTensor) -> SparseTensor:
        cm, ts = input.coordinate_manager, input._ts
        if input_glob._ts != 0:
            raise ValueError("the second operand must be a globally pooled tensor (one row per batch)")
        lvl = cm.level(ts)
        out = BroadcastMulFunction.apply(input.F, input_glob.F, lvl.coords, cm.batch_ptr(ts))
        return input._like(out)


class MinkowskiBroadcastAddition(nn.Module):
    def forward(self, input: SparseTensor, input_glob: SparseTensor) -> SparseTensor:
        cm, ts = input.coordinate_manager, input._ts
        lvl = cm.level(ts)
        ones = torch.ones_like(input.F)
        return input._like(input.F + BroadcastMulFunction.apply(ones, input_glob.F, lvl.coords, cm.batch_ptr(ts)))


# --------------------------------------------------------------------------- per-row (dense) layers
class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, input: SparseTensor) -> SparseTensor:
        # per-row dense layer on the library's own MFMA kernels (no BLAS call on the shared-MLP path)
        return input._like(dense_linear(input.F, self.linear.weight, self.linear.bias))


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, input: SparseTensor) -> SparseTensor:
        if input.F.shape[1] % 4 == 0:
            return input._like(batch_norm_act(input.F, self.bn, None))
        return input._like(self.bn(input.F))


class MinkowskiInstanceNorm(nn.Module):
    """Per-batch-element, per-channel normalisation (ME.MinkowskiInstanceNorm), eps = 1e-6."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.eps = 1e-6
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))
        self._avg = MinkowskiGlobalAvgPooling()
        self._bmul = MinkowskiBroadcastMultiplication()

    def forward(self, input: SparseTensor) -> SparseTensor:
        if input.F.dtype == torch.bfloat16:
            # bf16 row storage (KernelOptions.bf16_activations): this composite normalisation computes on fp32 rows and hands
            # bf16 rows on (the fused BatchNorm kernels are the ones with a bf16-row form)
            out = self.forward(input._like(input.F.float()))
            return out._like(out.F.to(torch.bfloat16))
        ones = input._like(torch.ones_like(input.F))
        mean = self._bmul(ones, self._avg(input))
        centred = input - mean
        var = self._avg(centred * centred)
        inv = SparseTensor(torch.rsqrt(var.F + self.eps), coordinate_map_key=var.coordinate_map_key,
                           coordinate_manager=var.coordinate_manager)
        out = self._bmul(centred, inv)
        return input._like(out.F * self.weight + self.bias)


class MinkowskiDropout(nn.Module):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.dropout = nn.Dropout(p, inplace=False)

    def forward(self, input: SparseTensor) -> SparseTensor:
        return input._like(self.dropout(input.F))


def _pointwise(name, torch_cls, fused_name=None):
    class _Act(nn.Module):
        def __init__(self, *args, **kwargs):
            super().__init__()
            kwargs.pop("inplace", None)  # features may alias saved activations of the HIP ops
            self.module = torch_cls(*args, **kwargs)
            # name understood by the fused BN/residual kernels (only for the plain default form)
            self.act_name = fused_name if not args and not kwargs else None

        def forward(self, input: SparseTensor) -> SparseTensor:
            return input._like(self.module(input.F))

    _Act.__name__ = _Act.__qualname__ = name
    return _Act


MinkowskiReLU = _pointwise("MinkowskiReLU", nn.ReLU, "relu")
MinkowskiGELU = _pointwise("MinkowskiGELU", nn.GELU, "gelu")
MinkowskiCELU = _pointwise("MinkowskiCELU", nn.CELU)
MinkowskiSiLU = _pointwise("MinkowskiSiLU", nn.SiLU)
MinkowskiELU = _pointwise("MinkowskiELU", nn.ELU)
MinkowskiSigmoid = _pointwise("MinkowskiSigmoid", nn.Sigmoid)
MinkowskiTanh = _pointwise("MinkowskiTanh", nn.Tanh)
MinkowskiLeakyReLU = _pointwise("MinkowskiLeakyReLU", nn.LeakyReLU)


class MinkowskiSinusoidal(nn.Module):
    """ME.MinkowskiSinusoidal (the "siren" entry of the reference's ACTIVATIONS table, common.py:40; no AGB
    configuration selects it): coef * sin(F @ kernel + bias)."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        self.kernel = nn.Parameter(torch.rand(in_channel, out_channel))
        self.bias = nn.Parameter(torch.rand(1, out_channel))
        self.coef = nn.Parameter(torch.rand(1, out_channel))

    def forward(self, input: SparseTensor) -> SparseTensor:
        return input._like(self.coef * torch.sin(input.F.mm(self.kernel) + self.bias))


class RegionType(enum.Enum):
    """ME.RegionType: the reference's common.py:75-86 builds lookup tables from these at import time.  Only HYPER_CUBE
    kernels exist in this library (the AGB models use no other)."""
    HYPER_CUBE = 0
    HYPER_CROSS = 1
    CUSTOM = 2


class KernelGenerator:
    """ME.KernelGenerator as far as the reference's helper constructors (common.py:135-212) use it: carries the
    geometry; anything but a hypercube region is refused."""

    def __init__(self, kernel_size=-1, stride=1, dilation=1, is_transpose=False, region_type=RegionType.HYPER_CUBE,
                 region_offsets=None, expand_coordinates=False, axis_types=None, dimension=-1):
        if region_type != RegionType.HYPER_CUBE or axis_types is not None:
            raise NotImplementedError("only HYPER_CUBE kernel regions are implemented")
        self.kernel_size, self.kernel_stride, self.kernel_dilation = kernel_size, stride, dilation
        self.region_type, self.dimension = region_type, dimension


def _unsupported(name):
    class _Unsupported(nn.Module):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(f"ME.{name} is not used by the AGB encoder path and is not implemented")

    _Unsupported.__name__ = _Unsupported.__qualname__ = name
    return _Unsupported


# named by helper functions of the reference's common.py that no AGB model calls (U-Net decoders, average / sum pooling)
MinkowskiConvolutionTranspose = _unsupported("MinkowskiConvolutionTranspose")
MinkowskiAvgPooling = _unsupported("MinkowskiAvgPooling")
MinkowskiSumPooling = _unsupported("MinkowskiSumPooling")
MinkowskiAvgUnpooling = _unsupported("MinkowskiAvgUnpooling")


def fused_norm_act(norm, act, x: SparseTensor) -> SparseTensor:
    """act(norm(x)) in one fused BatchNorm+activation kernel pair when norm is a MinkowskiBatchNorm and act a plain
    ReLU/GELU (the reference's ConvNormActivation / block wiring); otherwise the modules are applied one by one."""
    name = getattr(act, "act_name", None) if act is not None else "none"
    if isinstance(norm, MinkowskiBatchNorm) and name is not None and x.F.shape[1] % 4 == 0:
        return x._like(batch_norm_act(x.F, norm.bn, name))
    x = norm(x)
    return act(x) if act is not None else x


def fused_residual(out: SparseTensor, residual: SparseTensor, act, drop_scale=None) -> SparseTensor:
    """act(out * drop_scale[batch] + residual): the tail of every residual block in one kernel."""
    out._check_same_map(residual)
    name = getattr(act, "act_name", None)
    if name is not None and out.F.shape[1] % 4 == 0:
        lvl = out.coordinate_manager.level(out._ts)
        return out._like(AddActFunction.apply(out.F, residual.F, drop_scale, lvl.coords, ACT_IDS[name]))
    if drop_scale is not None:
        lvl = out.coordinate_manager.level(out._ts)
        s = drop_scale.view(-1, 1).expand(-1, out.F.shape[1]).contiguous()
        f = BroadcastMulFunction.apply(out.F, s, lvl.coords, out.coordinate_manager.batch_ptr(out._ts))
        out = out._like(f)
    return act(out + residual)


class _Namespace:
    pass


# ``from MinkowskiEngine import MinkowskiNormalization as N`` / ``MinkowskiNonlinearity as NL`` look-alikes
MinkowskiNormalization = _Namespace()
MinkowskiNormalization.MinkowskiBatchNorm = MinkowskiBatchNorm
MinkowskiNormalization.MinkowskiInstanceNorm = MinkowskiInstanceNorm
MinkowskiNonlinearity = _Namespace()
for _n in ("ReLU", "GELU", "CELU", "SiLU", "ELU", "Sigmoid", "Tanh", "LeakyReLU", "Sinusoidal"):
    setattr(MinkowskiNonlinearity, "Minkowski" + _n, globals()["Minkowski" + _n])

__all__ = [n for n in dir() if n.startswith("Minkowski")] + ["SparseTensor", "CoordinateManager", "CoordinateMapKey",
                                                              "RegionType", "KernelGenerator", "fused_norm_act",
                                                              "fused_residual"]
