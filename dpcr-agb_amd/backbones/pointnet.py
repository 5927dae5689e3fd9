"""MinkowskiPointNet (the reference's published "PointNet"): shared per-point MLP + per-plot pool + head MLP.

Same constructor arguments, attribute names and state_dict keys as the reference
(torch_points3d/modules/MinkowskiEngine/PointNet.py:9-49): blocks.{0,3,6}.linear, blocks.{1,4,7}.bn,
mlp.{0,3}.linear, mlp.{1,4}.bn, final.linear.  The whole forward is fp32 (PointNet.py:43).

MI355X shape of the forward pass: every Linear runs on the library's own MFMA kernels (identity-map convolution kernels,
csrc/spconv.hip), BatchNorm + activation are one fused kernel pair (csrc/norm.hip), and the LAST shared layer's
BatchNorm + activation is fused into the per-plot pooling (csrc/pointnet.hip): the [N, 1024] activation — 3.6 GB at
B = 64 — and, in the backward pass, the broadcast pooled gradient are never written to HBM.
"""
import torch.nn as nn

from .. import me_compat as ME
import torch

from ..norm_ops import POOL_MODES, batch_norm_act_pool, pointnet_mlp_forward
from ..sparse_ops import model_scope
from .sparse import ACTIVATIONS, GLOBAL_POOL

# (Sequential name, layer widths relative to the constructor arguments); each width adds Linear(no bias) + BN + act
_STACKS = (("blocks", lambda cin, emb: (cin, 64, 128, emb)), ("mlp", lambda cin, emb: (emb, 512, 256)))


class MinkowskiPointNet(nn.Module):
    def __init__(self, in_channels, out_channels, activation="relu", global_pool="max", embedding_channel=1024, D=3,
                 dropout=0.0, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.act_fn = ACTIVATIONS[activation]()
        for name, widths in _STACKS:
            w = widths(D + in_channels, embedding_channel)
            layers = []
            for cin, cout in zip(w[:-1], w[1:]):
                layers += [ME.MinkowskiLinear(cin, cout, bias=False), ME.MinkowskiBatchNorm(cout, momentum=bn_momentum),
                           self.act_fn]
            setattr(self, name, nn.Sequential(*layers))
        self.global_pool_name = global_pool
        self.global_pool = GLOBAL_POOL[global_pool]()
        self.dp1 = ME.MinkowskiDropout(dropout)
        self.final = ME.MinkowskiLinear(256, out_channels, bias=True)
        # (global_pool holds no parameters: parameter / state_dict order = blocks, mlp, final, as in the reference)

    @staticmethod
    def _run(mods, x):
        """(Linear, BatchNorm, act) triples with BatchNorm + activation in one fused kernel pair."""
        for i in range(0, len(mods), 3):
            x = ME.fused_norm_act(mods[i + 1], mods[i + 2], mods[i](x))
        return x

    def _embed(self, x):
        """Shared MLP + per-plot pooling; the last BatchNorm + activation runs inside the pooling kernel when it can."""
        mods = list(self.blocks)
        name = getattr(mods[2], "act_name", None)
        pool = "avg" if self.global_pool_name == "mean" else self.global_pool_name
        fusable = (x.F.is_cuda and name is not None and pool in POOL_MODES and x._ts != 0 and len(mods) == 9
                   and all(isinstance(mods[i], ME.MinkowskiBatchNorm) and mods[i].bn.track_running_stats and
                           mods[i].bn.affine and mods[i].bn.num_features % 4 == 0 for i in (1, 4, 7)))
        if fusable and not self.training and not torch.is_grad_enabled():
            # inference: the whole shared MLP in one library call (agb_pointnet_mlp_fwd), running statistics
            cm, ts = x.coordinate_manager, x._ts
            pooled, _ = pointnet_mlp_forward(x.F, [(mods[i].linear, mods[i + 1].bn) for i in (0, 3, 6)], name,
                                             cm.batch_ptr(ts), cm.batch_size, pool)
            return ME.SparseTensor(pooled, coordinate_map_key=ME.CoordinateMapKey(0), coordinate_manager=cm)
        x = self._run(mods[:-3], x)
        lin, norm, act = mods[-3:]
        z = lin(x)
        if (z.F.is_cuda and isinstance(norm, ME.MinkowskiBatchNorm) and name is not None and z.F.shape[1] % 4 == 0
                and pool in POOL_MODES and z._ts != 0):
            cm, ts = z.coordinate_manager, z._ts
            pooled = batch_norm_act_pool(z.F, norm.bn, name, cm.level(ts).coords, cm.batch_ptr(ts), cm.batch_size, pool)
            return ME.SparseTensor(pooled, coordinate_map_key=ME.CoordinateMapKey(0), coordinate_manager=cm)
        return self.global_pool(ME.fused_norm_act(norm, act, z))

    kernel_options = None      # sparse_ops.KernelOptions of this model (None: the ones in force / the defaults)

    def forward(self, x):
        with model_scope(self):
            x = self._embed(x)
            x = self._run(list(self.mlp), x)
            x = self.dp1(x)
            return self.final(x)
