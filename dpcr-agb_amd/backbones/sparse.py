"""Sparse-voxel ResNet / SENet backbones (MSENet14 / MSENet50 of the reference) on the HIP sparse ops.

Same constructor arguments, module attribute names and state_dict keys as the reference
(torch_points3d/modules/MinkowskiEngine/SENet.py:14-118 ResNetBase, :151 SENet14, :185 SENet50;
resnet_block.py:31-133 BasicBlock/Bottleneck; senet_block.py:33-147 SELayer/SEBasicBlock/SEBottleneck;
common.py:215-226 ConvNormActivation, :344-366 MinkowskiDropPath), so checkpoints trained with the reference
load unchanged:  blocks.0.0.{conv,norm.bn}, blocks.<s>.<i>.{conv1,norm1.bn,conv2,norm2.bn,se.fc.{0,2}.linear,
downsample.{0,1.bn}}, final.linear.
"""
import random
from functools import partial

import torch
import torch.nn as nn

from .. import me_compat as ME
from ..se_ops import MAX_HIDDEN as MAX_SE_HIDDEN, se_layer
from ..sparse_ops import current as current_options, model_scope

ACTIVATIONS = {
    "relu": ME.MinkowskiReLU,
    "celu": partial(ME.MinkowskiCELU, alpha=0.54),
    "silu": ME.MinkowskiSiLU,
    "swish": ME.MinkowskiSiLU,
    "elu": partial(ME.MinkowskiELU, alpha=0.54),
    "sigmoid": ME.MinkowskiSigmoid,
    "tanh": ME.MinkowskiTanh,
    "gelu": ME.MinkowskiGELU,
}

GLOBAL_POOL = {
    "max": ME.MinkowskiGlobalMaxPooling,
    "mean": ME.MinkowskiGlobalAvgPooling,
    "sum": ME.MinkowskiGlobalSumPooling,
}


class MinkowskiDropPath(nn.Module):
    """Stochastic depth per batch element; draws one ``random.uniform(0, 1)`` per element in batch order,
    like the reference (common.py:355-361), so a seeded run drops the same samples."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    # pinned staging rows: the per-step H2D copies stay asynchronous; a row is rewritten only after the device has
    # executed the copy that read it (one event per row)
    _RING, _ring, _slot, _done = 64, None, 0, None

    def _draw(self, B):
        keep_prob = 1 - self.drop_prob
        keep = [1.0 if random.uniform(0, 1) > self.drop_prob else 0.0 for _ in range(B)]
        if keep_prob > 0.0 and self.scale_by_keep:
            keep = [k / keep_prob for k in keep]
        return keep

    @staticmethod
    def _upload(values, device):
        """float[len(values)] on the device through a ring of pinned staging rows (asynchronous copy)."""
        cls = MinkowskiDropPath
        n = len(values)
        if cls._ring is None or cls._ring.shape[1] < n:
            if cls._done is not None:
                for ev in cls._done:
                    if ev is not None:
                        ev.synchronize()
            cls._ring = torch.empty(cls._RING, max(n, 1024), dtype=torch.float32).pin_memory()
            cls._done = [None] * cls._RING
        slot = cls._slot % cls._RING
        cls._slot += 1
        if cls._done[slot] is not None:
            cls._done[slot].synchronize()
        row = cls._ring[slot, :n]
        row.copy_(torch.tensor(values, dtype=torch.float32))
        out = row.to(device, non_blocking=True)
        if out.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            cls._done[slot] = ev
        return out

    def scale_vector(self, x):
        """float[B] on the device: keep/(1-p) per batch element, or None when nothing is to be applied."""
        if not self.training:
            return None
        preset = self.__dict__.pop("_preset", None)
        if preset is not None:          # (drawn at the start of this forward pass together with the other blocks' vectors)
            return preset
        return self._upload(self._draw(x.coordinate_manager.batch_size), x.device)

    @staticmethod
    def predraw(blocks, B, device):
        """The drop-path vectors of all `blocks` (modules with a ``drop_path``) of one forward pass, drawn NOW in block order
        — the same ``random`` sequence as one draw per block while it runs, provided nothing else draws in between — and
        uploaded with ONE copy; every block's ``scale_vector`` then returns its slice."""
        dps = [b.drop_path for b in blocks if isinstance(b.drop_path, MinkowskiDropPath) and b.drop_path.training]
        if not dps:
            return
        values = []
        for dp in dps:
            values += dp._draw(B)
        flat = MinkowskiDropPath._upload(values, device)
        for i, dp in enumerate(dps):
            dp.__dict__["_preset"] = flat[i * B:(i + 1) * B]

    def forward(self, x):
        scale = self.scale_vector(x)
        if scale is None:
            return x
        cm = x.coordinate_manager
        s = scale.view(-1, 1).expand(-1, x.F.shape[1]).contiguous()
        glob = ME.SparseTensor(s, coordinate_map_key=ME.CoordinateMapKey(0), coordinate_manager=cm)
        return ME.MinkowskiBroadcastMultiplication()(x, glob)


class MinkowskiLayerNorm(nn.Module):
    """Channel-wise layer normalisation of a sparse tensor's rows (common.py:369-386: nn.LayerNorm(C, eps=1e-6) on .F;
    state_dict keys ``ln.weight`` / ``ln.bias``), on csrc/layernorm.hip."""

    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.ln = nn.LayerNorm(normalized_shape, eps=eps)

    def forward(self, input):
        from ..norm_ops import layer_norm
        if input.F.dtype == torch.bfloat16:     # bf16 row storage: csrc/layernorm.hip takes fp32 rows
            return input._like(layer_norm(input.F.float(), self.ln).to(torch.bfloat16))
        return input._like(layer_norm(input.F, self.ln))


def _residual_tail(block, out, residual):
    """relu(drop_path(out) + residual) (resnet_block.py:70-73) as one fused kernel."""
    dp = block.drop_path
    scale = dp.scale_vector(out) if isinstance(dp, MinkowskiDropPath) else None
    if not isinstance(dp, (MinkowskiDropPath, nn.Identity)):
        out = dp(out)
    return ME.fused_residual(out, residual, block.relu, scale)


class ConvNormActivation(nn.Module):
    def __init__(self, input_channels, out_channels, kernel_size, stride, norm_layer, activation_layer, bias, D):
        super().__init__()
        self.conv = ME.MinkowskiConvolution(input_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                            dimension=D, bias=bias)
        self.norm = norm_layer(out_channels)
        self.act = nn.Identity() if activation_layer is None else activation_layer

    def forward(self, x):
        return ME.fused_norm_act(self.norm, self.act if not isinstance(self.act, nn.Identity) else None, self.conv(x))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, act_fn, norm_layer, stride=1, dilation=1, downsample=None, drop_path=0.0,
                 bias=True, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                             dimension=dimension, bias=bias)
        self.norm1 = norm_layer(planes)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                             dimension=dimension, bias=bias)
        self.norm2 = norm_layer(planes)
        self.relu = act_fn
        self.downsample = downsample if downsample is not None else nn.Identity()
        self.drop_path = MinkowskiDropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def _main(self, x):
        out = ME.fused_norm_act(self.norm1, self.relu, self.conv1(x))
        return self.norm2(self.conv2(out))

    def forward(self, x):
        out = self._main(x)
        residual = self.downsample(x)
        return _residual_tail(self, out, residual)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, act_fn, norm_layer, stride=1, dilation=1, downsample=None, drop_path=0.0,
                 bias=True, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension, bias=bias)
        self.norm1 = norm_layer(planes)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                             dimension=dimension, bias=bias)
        self.norm2 = norm_layer(planes)
        self.conv3 = ME.MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension,
                                             bias=bias)
        self.norm3 = norm_layer(planes * self.expansion)
        self.relu = act_fn
        self.downsample = downsample if downsample is not None else nn.Identity()
        self.drop_path = MinkowskiDropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def _main(self, x):
        out = ME.fused_norm_act(self.norm1, self.relu, self.conv1(x))
        out = ME.fused_norm_act(self.norm2, self.relu, self.conv2(out))
        return self.norm3(self.conv3(out))

    def forward(self, x):
        # (x feeds conv1 and the shortcut: the shortcut's gradient joins in conv1's data-gradient kernel)
        out, x = self.conv1.forward_join(x)
        out = ME.fused_norm_act(self.norm1, self.relu, out)
        out = ME.fused_norm_act(self.norm2, self.relu, self.conv2(out))
        out = self.norm3(self.conv3(out))
        residual = self.downsample(x)
        return _residual_tail(self, out, residual)


class SELayer(nn.Module):
    """Squeeze-excite: global AVERAGE pool -> Linear(C, C/r) -> act -> Linear(C/r, C) -> sigmoid -> broadcast mul."""

    def __init__(self, channel, act_fn, reduction=16, dimension=-1):
        super().__init__()
        self.fc = nn.Sequential(
            ME.MinkowskiLinear(channel, channel // reduction),
            act_fn,
            ME.MinkowskiLinear(channel // reduction, channel),
            ME.MinkowskiSigmoid(),
        )
        self.pooling = ME.MinkowskiGlobalPooling()
        self.broadcast_mul = ME.MinkowskiBroadcastMultiplication()

    def forward(self, x):
        lin1, act, lin2, gate = self.fc[0], self.fc[1], self.fc[2], self.fc[3]
        name = getattr(act, "act_name", None)
        if (name in ("relu", "gelu") and isinstance(gate, ME.MinkowskiSigmoid) and x.F.is_cuda
                and lin1.linear.out_features <= MAX_SE_HIDDEN and x._ts != 0 and x.F.shape[1] % 4 == 0):
            # pooling + excitation MLP + broadcast multiplication as one autograd node (csrc/se.hip, pool.hip):
            # 3 launches forward, 4 backward, and no gradient addition where x feeds both branches
            cm, ts = x.coordinate_manager, x._ts
            return x._like(se_layer(x.F, cm.level(ts).coords, cm.batch_ptr(ts), cm.batch_size, lin1.linear, name,
                                    lin2.linear))
        pooled = self.pooling(x)
        return self.broadcast_mul(x, self.fc(pooled))


def _se_block_forward(block, x, z, last_norm):
    """Everything behind the last convolution of an SE block (its output z before `last_norm`): the one-node fused tail
    (se_ops.SEBlockTailFunction) when the pieces are the plain ones, the module-by-module path otherwise."""
    import dpcr_agb_amd.se_ops as se_ops
    se, dp = block.se, block.drop_path
    lin1, act, lin2, gate = se.fc[0], se.fc[1], se.fc[2], se.fc[3]
    se_name, act_name = getattr(act, "act_name", None), getattr(block.relu, "act_name", None)
    residual = block.downsample(x)
    if (current_options().fused_tail and isinstance(last_norm, ME.MinkowskiBatchNorm) and z.F.is_cuda and z.F.shape[1] % 4 == 0
            and se_name in ("relu", "gelu") and act_name in ("relu", "gelu") and isinstance(gate, ME.MinkowskiSigmoid)
            and lin1.linear.out_features <= MAX_SE_HIDDEN and z._ts != 0
            and isinstance(dp, (MinkowskiDropPath, nn.Identity))):
        z._check_same_map(residual)
        cm, ts = z.coordinate_manager, z._ts
        keep = dp.scale_vector(z) if isinstance(dp, MinkowskiDropPath) else None
        return z._like(se_ops.se_block_tail(z.F, residual.F, last_norm.bn, cm.level(ts).coords, cm.batch_ptr(ts),
                                            cm.batch_size, lin1.linear, se_name, lin2.linear, keep, act_name))
    return _residual_tail(block, se(last_norm(z)), residual)


class SEBasicBlock(BasicBlock):
    def __init__(self, inplanes, planes, act_fn, norm_layer, stride=1, dilation=1, downsample=None, reduction=16,
                 drop_path=0.0, bias=True, dimension=-1):
        super().__init__(inplanes, planes, act_fn, norm_layer, stride=stride, dilation=dilation,
                         downsample=downsample, drop_path=drop_path, bias=bias, dimension=dimension)
        self.se = SELayer(planes, act_fn, reduction=reduction, dimension=dimension)

    def _main(self, x):
        return self.se(super()._main(x))

    def forward(self, x):
        out = ME.fused_norm_act(self.norm1, self.relu, self.conv1(x))
        return _se_block_forward(self, x, self.conv2(out), self.norm2)


class SEBottleneck(Bottleneck):
    def __init__(self, inplanes, planes, act_fn, norm_layer, stride=1, dilation=1, downsample=None, dimension=-1,
                 drop_path=0.0, bias=True, reduction=16):
        super().__init__(inplanes, planes, act_fn, norm_layer, stride=stride, dilation=dilation,
                         downsample=downsample, drop_path=drop_path, bias=bias, dimension=dimension)
        self.se = SELayer(planes * self.expansion, act_fn, reduction=reduction, dimension=dimension)

    def _main(self, x):
        return self.se(super()._main(x))

    def forward(self, x):
        out, x = self.conv1.forward_join(x)
        out = ME.fused_norm_act(self.norm1, self.relu, out)
        out = ME.fused_norm_act(self.norm2, self.relu, self.conv2(out))
        return _se_block_forward(self, x, self.conv3(out), self.norm3)


class ResNetBase(nn.Module):
    BLOCK = None
    LAYERS = ()
    STRIDES = None
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)

    def __init__(self, in_channels, out_channels, activation="relu", D=3, first_stride=2, dropout=0.0, drop_path=0.0,
                 bn_momentum=0.1, norm_type="bn", global_pool="mean", use_gn=False, bias=True, **kwargs):
        super().__init__()
        assert self.BLOCK is not None and self.STRIDES is not None
        self.D, self.bias, self.bn_momentum, self.drop_path = D, bias, bn_momentum, drop_path
        self.act_fn = ACTIVATIONS[activation]()
        self.norm_type = norm_type
        if norm_type == "bn":
            self.norm_layer = partial(ME.MinkowskiBatchNorm, momentum=bn_momentum)
        elif norm_type == "bn_no_affine":
            self.norm_layer = partial(ME.MinkowskiBatchNorm, momentum=bn_momentum, affine=False)
        elif norm_type == "in":
            self.norm_layer = ME.MinkowskiInstanceNorm
        elif norm_type == "ln":
            self.norm_layer = MinkowskiLayerNorm
        else:
            raise NotImplementedError(f"Choose either 'bn', 'in', or 'ln'. Given: {norm_type}")

        # tensor strides the forward pass will visit (lets set_input build the whole coordinate pyramid at once)
        ts, self.tensor_strides = first_stride * 2, [first_stride, first_stride * 2]
        for s in self.STRIDES:
            ts *= s
            self.tensor_strides.append(ts)

        self.inplanes = self.INIT_DIM
        stem = nn.Sequential(
            ConvNormActivation(in_channels, self.inplanes, kernel_size=7, stride=first_stride, D=D, bias=bias,
                               activation_layer=self.act_fn, norm_layer=self.norm_layer),
            ME.MinkowskiMaxPooling(kernel_size=3, stride=2, dimension=D),
        )
        stages = [stem]
        for planes, layers, stride in zip(self.PLANES, self.LAYERS, self.STRIDES):
            stages.append(self._make_layer(self.BLOCK, planes, layers, stride=stride))
        self.blocks = nn.ModuleList(stages)

        self.glob_avg = GLOBAL_POOL[global_pool]()
        if dropout > 0:
            self.glob_avg = nn.Sequential(self.glob_avg, ME.MinkowskiDropout(dropout))
        self.final = ME.MinkowskiLinear(self.inplanes, out_channels, bias=True)
        self.apply(self.init_weights)

    @staticmethod
    def init_weights(m):
        if isinstance(m, ME.MinkowskiBatchNorm) and m.bn.affine:
            nn.init.constant_(m.bn.weight, 1)
            nn.init.constant_(m.bn.bias, 0)
        if isinstance(m, ME.MinkowskiConvolution):
            nn.init.trunc_normal_(m.kernel, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        if isinstance(m, ME.MinkowskiLinear):
            nn.init.trunc_normal_(m.linear.weight, std=0.02)
            if m.linear.bias is not None:
                nn.init.constant_(m.linear.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                ME.MinkowskiConvolution(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride,
                                        dimension=self.D, dilation=1, bias=self.bias),
                self.norm_layer(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, self.act_fn, stride=stride, dilation=dilation, downsample=downsample,
                        dimension=self.D, drop_path=self.drop_path, bias=self.bias, norm_layer=self.norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, self.act_fn, stride=1, dilation=dilation, dimension=self.D,
                                drop_path=self.drop_path, bias=self.bias, norm_layer=self.norm_layer))
        return nn.Sequential(*layers)

    def plan_spec(self, input_requires_grad=False):
        """Kernel maps the forward/backward pass will ask for: (ts_in, K, stride, dilation, need_transposed).
        Lets the input pipeline build them ahead of time (on a side stream) instead of lazily inside forward."""
        specs = []

        def visit(conv, ts, needs_dx):
            if getattr(conv, "use_mm", False):
                return ts
            k, s, d = conv.kernel_size, conv.stride, conv.dilation
            # a 3-channel layer whose input needs no gradient probes the dense grid itself when it can
            widths = None
            if conv.in_channels >= 12:      # padded widths as SparseConvFunction hands them over
                widths = (-(-conv.in_channels // 4) * 4, -(-conv.out_channels // 4) * 4)
            specs.append((ts, k, s, d, needs_dx and not (s == 1 and k % 2 == 1), conv.in_channels == 3 and not needs_dx,
                          widths))
            return ts * s

        stem, pool = self.blocks[0][0], self.blocks[0][1]
        ts = visit(stem.conv, 1, input_requires_grad)
        specs.append((ts, pool.kernel_size, pool.stride, pool.dilation, True))
        ts *= pool.stride
        for stage in list(self.blocks)[1:]:
            for blk in stage:
                ts_in = ts
                convs = [blk.conv1, blk.conv2] + ([blk.conv3] if hasattr(blk, "conv3") else [])
                for c in convs:
                    ts = visit(c, ts, True)
                if not isinstance(blk.downsample, nn.Identity):
                    visit(blk.downsample[0], ts_in, True)
        return specs

    kernel_options = None      # sparse_ops.KernelOptions of this model (None: the ones in force / the defaults)
    supports_bf16_rows = True  # KernelOptions.bf16_activations: every row kernel of this backbone has a bf16-row form

    def _fused_plan(self):
        """Which stages the one-call-per-block path (fused_blocks.py) can take: (stem, [[block, ...] per stage]); a matter of
        module structure, decided once."""
        plan = self.__dict__.get("_agb_fused_plan")
        if plan is None:
            from .. import fused_blocks as FB
            stages = list(self.blocks)
            plan = (FB.stem_supported(stages[0]), [[FB.block_supported(b) for b in st] for st in stages[1:]])
            self.__dict__["_agb_fused_plan"] = plan
        return plan

    def forward(self, x):
        return self.final(self.forward_features(x))

    def forward_features(self, x):
        """Everything in front of the head: stem, stages, global pooling (SENet.py:113-117) -> SparseTensor [B, C]."""
        with model_scope(self):
            from .. import fused_blocks as FB
            opts = current_options()
            if not FB.options_allow(opts):
                if self.training:      # every block's drop-path vector: one draw in block order, one upload
                    MinkowskiDropPath.predraw([b for st in list(self.blocks)[1:] for b in st if hasattr(b, "drop_path")],
                                              x.coordinate_manager.batch_size, x.F.device)
                for block in self.blocks:
                    x = block(x)
            else:
                stem_ok, stages_ok = self._fused_plan()
                stages = list(self.blocks)
                fused = [blk for stage, oks in zip(stages[1:], stages_ok) for blk, ok in zip(stage, oks) if ok]
                wt = None
                if fused and torch.is_grad_enabled():
                    wt = FB.weight_transposes(self, fused)
                    wt.ensure()       # one launch per step: W^T of every fused convolution, for the data gradients
                if fused and self.training and all(all(oks) for oks in stages_ok):
                    # every residual block runs fused: their drop-path vectors in one draw + one upload
                    MinkowskiDropPath.predraw(fused, x.coordinate_manager.batch_size, x.F.device)
                out = FB.run_stem(stages[0], x, opts) if stem_ok else None
                x = out if out is not None else stages[0](x)
                for stage, oks in zip(stages[1:], stages_ok):
                    for blk, ok in zip(stage, oks):
                        out = FB.run_block(blk, x, opts, wt) if ok else None
                        x = out if out is not None else blk(x)
            return self.glob_avg(x)


def _variant(name, block, layers, strides=(1, 2, 2, 2), init_dim=64, planes=(64, 128, 256, 512)):
    return type(name, (ResNetBase,), dict(BLOCK=block, LAYERS=layers, STRIDES=strides, INIT_DIM=init_dim,
                                          PLANES=planes))


ResNet14_ = _variant("ResNet14_", BasicBlock, (1, 1, 1, 1))
ResNet18_ = _variant("ResNet18_", BasicBlock, (2, 2, 2, 2))
ResNet34_ = _variant("ResNet34_", BasicBlock, (3, 4, 6, 3))
ResNet50_ = _variant("ResNet50_", Bottleneck, (3, 4, 6, 3))
ResNet101_ = _variant("ResNet101_", Bottleneck, (3, 4, 23, 3))
SENet14 = _variant("SENet14", SEBasicBlock, (1, 1, 1, 1))
SENet18 = _variant("SENet18", SEBasicBlock, (2, 2, 2, 2))
SENet34 = _variant("SENet34", SEBasicBlock, (3, 4, 6, 3))
SENet50 = _variant("SENet50", SEBottleneck, (3, 4, 6, 3))
SENet101 = _variant("SENet101", SEBottleneck, (3, 4, 23, 3))
SENet17_6deep = _variant("SENet17_6deep", SEBasicBlock, (1, 1, 1, 1, 2, 1), (1, 2, 2, 2, 2, 2), 32,
                         (32, 64, 128, 256, 512, 1024))
SENet17_5deep = _variant("SENet17_5deep", SEBasicBlock, (1, 1, 1, 2, 2), (1, 2, 2, 2, 2), 64,
                         (64, 128, 256, 512, 1024))

__all__ = ["ResNetBase", "ResNet14_", "ResNet18_", "ResNet34_", "ResNet50_", "ResNet101_", "SENet14", "SENet18",
           "SENet34", "SENet50", "SENet101", "SENet17_6deep", "SENet17_5deep", "BasicBlock", "Bottleneck",
           "SEBasicBlock", "SEBottleneck", "SELayer", "ConvNormActivation", "MinkowskiDropPath", "MinkowskiLayerNorm",
           "ACTIVATIONS",
           "GLOBAL_POOL"]
