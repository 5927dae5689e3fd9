"""ORACLE (test infrastructure only). Per-sample CPU restatement of the deterministic part of the NFI sparse transform
chain (torch-points3d/conf/data/instance/NFI/transforms/sparse-xy.yaml:106-150, test_transform) and of the two
train-time coordinate augmentations that follow GridSampling3D (:100-104).

The reference classes live in torch_points3d/core/data_transform/{transforms,features,sparse_transforms}.py, which do
not import here (torch_geometric, numba, dbscan1d, omegaconf are absent and not installable offline), so each function
restates the reference lines it cites with the same torch / numpy / matplotlib calls:
  scale_pos                transforms.py:590-598      torch.div / torch.mul by a [1,3] tensor
  move_center              transforms.py:722-739      pos += [cx, cy, cz]         (cz defaults to 0.5: :734)
  start_z_from_zero        transforms.py:766-769      pos[:, 2] -= pos[:, 2].min()
  polygon_extend           transforms.py:1461-1496    matplotlib.path.Path(polygon).contains_points(pos[:, :2])
                                                      (matplotlib IS installed: the third-party call is the real one)
  max_points / min_points  transforms.py:1312-1358, 1742-1800   torch.randperm choice (injected here)
  features                 features.py:307-334,353-383; AddFeatsByKeys -> x = cat([ones, pos_z, xy_distance])
                           xy_distance = torch.nn.PairwiseDistance()(pos[:, :2], centre)   (eps 1e-6)
  random_coords_flip       sparse_transforms.py:49-55 one random.random() per non-ignored axis (set order x, y)
  shift_voxels             transforms.py:1046-1054    one random.random(), then (torch.rand(3) * 100) cast to int
Pinned by: the real matplotlib call, torch's own PairwiseDistance, and known-answer tests (tests/test_transforms.py).
"""
import math
import random

import numpy as np
import torch
from matplotlib.path import Path

HEXAGON = [[0.0, 0.5], [0.25, 0.9330127], [0.75, 0.9330127], [1.0, 0.5], [0.75, 0.0669873], [0.25, 0.0669873]]


def scale_pos(pos, scale, op="div"):
    s = torch.tensor([scale[0], scale[1], scale[2]]).unsqueeze(0)
    return (torch.div if op == "div" else torch.mul)(pos, s)


def move_center(pos, center_x=0.5, center_y=0.5, center_z=0.5):
    return pos + torch.FloatTensor([[center_x, center_y, center_z]])


def start_z_from_zero(pos):
    pos = pos.clone()
    pos[:, 2] -= pos[:, 2].min()
    return pos


def polygon_mask(pos, polygon=HEXAGON):
    return torch.from_numpy(Path(polygon).contains_points(pos[:, [0, 1]].numpy()))


def fixed_points_choice(num_nodes, num, allow_duplicates):
    """FixedPointsOwn.__call__ with replace=False (transforms.py:1337-1350); consumes torch's global RNG."""
    if not allow_duplicates:
        return torch.randperm(num_nodes)[:num]
    return torch.cat([torch.randperm(num_nodes) for _ in range(math.ceil(num / num_nodes))], dim=0)[:num]


def features(pos, center_x=0.5, center_y=0.5):
    ones = torch.ones((pos.shape[0], 1)).float()
    pos_z = pos[:, 2].clone()
    centre = torch.tensor([[center_x, center_y]])
    dist = torch.nn.PairwiseDistance()(pos[:, :2], centre.repeat_interleave(pos.shape[0], dim=0))
    return torch.cat([ones, pos_z.unsqueeze(-1), dist.unsqueeze(-1)], -1)


def test_transform_sample(pos, scale=(30.0, 30.0, 40.0), center=(0.5, 0.5), polygon=HEXAGON, max_points=16000,
                          min_points=500):
    """One plot through ScalePos .. AddFeatsByKeys.  Returns (pos [M,3], x [M,3], src [M] rows of the input).
    MaxPoints / MinPoints draw from torch's global RNG exactly when the reference does."""
    p = scale_pos(pos.float(), scale, "div")
    p = move_center(p, center[0], center[1])
    p = start_z_from_zero(p)
    src = torch.arange(p.shape[0])
    m = polygon_mask(p, polygon)
    p, src = p[m], src[m]
    if p.shape[0] > max_points:
        ch = fixed_points_choice(p.shape[0], max_points, allow_duplicates=False)
        p, src = p[ch], src[ch]
    if 0 < p.shape[0] < min_points:
        ch = fixed_points_choice(p.shape[0], min_points, allow_duplicates=True)
        p, src = p[ch], src[ch]
    return p, features(p, center[0], center[1]), src


def random_coords_flip(coords, p=0.5, ignored_axis=("z",)):
    """In place on an int tensor [M,3]; returns the flags drawn (per axis, in the reference's set iteration order)."""
    mapping = {"x": 0, "y": 1, "z": 2}
    axes = set(range(3)) - {mapping[a] for a in ignored_axis}
    flags = [0, 0, 0]
    for ax in axes:
        if random.random() < p:
            flags[ax] = 1
            coords[:, ax] = torch.max(coords[:, ax]) - coords[:, ax]
    return flags


def shift_voxels(coords, p=0.5):
    """In place; returns the shift applied (zeros when the coin flip says no)."""
    shift = torch.zeros(3, dtype=coords.dtype)
    if random.random() < p:
        shift = (torch.rand(3) * 100).type_as(coords)
        coords[:, :3] += shift
    return shift
