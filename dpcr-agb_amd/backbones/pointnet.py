"""MinkowskiPointNet (the reference's published "PointNet"): shared per-point MLP + per-plot pool + head MLP.

Same constructor, attribute names and state_dict keys as the reference
(torch_points3d/modules/MinkowskiEngine/PointNet.py:9-49): blocks.{0,3,6}.linear, blocks.{1,4,7}.bn,
mlp.{0,3}.linear, mlp.{1,4}.bn, final.linear.  The whole forward is fp32 (PointNet.py:43).
"""
import torch.nn as nn

from .. import me_compat as ME
from .sparse import ACTIVATIONS, GLOBAL_POOL


class MinkowskiPointNet(nn.Module):
    def __init__(self, in_channels, out_channels, activation="relu", global_pool="max", embedding_channel=1024, D=3,
                 dropout=0.0, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.act_fn = ACTIVATIONS[activation]()
        self.blocks = nn.Sequential(
            ME.MinkowskiLinear(D + in_channels, 64, bias=False),
            ME.MinkowskiBatchNorm(64, momentum=bn_momentum),
            self.act_fn,
            ME.MinkowskiLinear(64, 128, bias=False),
            ME.MinkowskiBatchNorm(128, momentum=bn_momentum),
            self.act_fn,
            ME.MinkowskiLinear(128, embedding_channel, bias=False),
            ME.MinkowskiBatchNorm(embedding_channel, momentum=bn_momentum),
            self.act_fn,
        )
        self.global_pool = GLOBAL_POOL[global_pool]()
        self.mlp = nn.Sequential(
            ME.MinkowskiLinear(embedding_channel, 512, bias=False),
            ME.MinkowskiBatchNorm(512, momentum=bn_momentum),
            self.act_fn,
            ME.MinkowskiLinear(512, 256, bias=False),
            ME.MinkowskiBatchNorm(256, momentum=bn_momentum),
            self.act_fn,
        )
        self.dp1 = ME.MinkowskiDropout(dropout)
        self.final = ME.MinkowskiLinear(256, out_channels, bias=True)

    @staticmethod
    def _run(seq, x):
        """(Linear, BatchNorm, act) triples of the Sequential with BatchNorm+activation in one fused kernel pair."""
        mods = list(seq)
        for i in range(0, len(mods), 3):
            x = ME.fused_norm_act(mods[i + 1], mods[i + 2], mods[i](x))
        return x

    def forward(self, x):
        x = self._run(self.blocks, x)
        x = self.global_pool(x)
        x = self._run(self.mlp, x)
        x = self.dp1(x)
        return self.final(x)
