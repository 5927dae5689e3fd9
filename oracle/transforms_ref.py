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


# ------------------------------------------------------------------------------------------------ train chain
def euler_angles_to_rotation_matrix(theta, random_order=False):
    """torch_points3d/utils/geometry.py:5-22, restated line by line."""
    R_x = torch.tensor(
        [[1, 0, 0], [0, torch.cos(theta[0]), -torch.sin(theta[0])], [0, torch.sin(theta[0]), torch.cos(theta[0])]])
    R_y = torch.tensor(
        [[torch.cos(theta[1]), 0, torch.sin(theta[1])], [0, 1, 0], [-torch.sin(theta[1]), 0, torch.cos(theta[1])]])
    R_z = torch.tensor(
        [[torch.cos(theta[2]), -torch.sin(theta[2]), 0], [torch.sin(theta[2]), torch.cos(theta[2]), 0], [0, 0, 1]])
    matrices = [R_x, R_y, R_z]
    if random_order:
        random.shuffle(matrices)
    return torch.mm(matrices[2], torch.mm(matrices[1], matrices[0]))


def train_transform_sample(raw, min_v=0.05, max_v=0.5, p_ground=0.1, min_points=500, dropout_ratio=0.2,
                           dropout_application_ratio=0.5, scale=(30.0, 30.0, 40.0), sigma=0.0025, clip=0.05,
                           rot=(0, 0, 180), shift_p=0.5, shift_max=(0.01, 0.01, 0.0), center=(0.5, 0.5),
                           n_max_points=12000, add_ratio=(0.01, 0.2), add_p=0.25, cj_sigma=0.005, cj_clip=0.015,
                           polygons=(HEXAGON,), rotate=180):
    """One sample through sparse-xy.yaml:5-69 (RandomGroundRemoval .. RandomPolygon2dExtend), drawing from the global
    ``random`` / ``numpy.random`` / ``torch`` generators exactly where the reference classes do (file:line in the
    module docstring and below).  Returns (pos after the crop, src: row of every kept point in the pre-crop cloud =
    [kept originals, added points, jittered copies], pre-crop cloud, index of the kept originals in `raw`)."""
    from matplotlib.transforms import Affine2D
    pos = raw.float().clone()
    orig = torch.arange(len(pos))
    # RandomGroundRemoval.__call__  transforms.py:1140-1150
    if random.random() < p_ground:
        remove_v = random.random() * (max_v - min_v) + min_v
        cond = pos[:, 2] > remove_v
        if not cond.sum() < min_points:
            pos[:, 2] -= remove_v
            pos, orig = pos[cond], orig[cond]
    # RandomDropout.__call__  transforms.py:1078-1082
    N = len(pos)
    if N > min_points and random.random() < dropout_application_ratio:
        ch = fixed_points_choice(N, int(N * (1 - dropout_ratio)), allow_duplicates=True)
        pos, orig = pos[ch], orig[ch]
    pos = scale_pos(pos, scale, "div")                                       # ScalePos
    if random.random() < 1:                                                  # RandomNoise :498-503 (p None -> 1)
        noise = sigma * torch.randn(pos.shape)
        pos = pos + noise.clamp(-clip, clip)
    thetas = torch.zeros(3, dtype=torch.float)                               # Random3AxisRotation features.py:44-57
    for axis_ind, deg_angle in enumerate([abs(min(r, 180)) if r else 0 for r in rot]):
        if deg_angle > 0 and random.random() < 1:
            rand_deg_angle = random.random() * 2 * deg_angle - deg_angle
            thetas[axis_ind] = float(rand_deg_angle * np.pi) / 180.0
    M = euler_angles_to_rotation_matrix(thetas, random_order=True)
    pos = pos.float() @ M.T
    if random.random() > shift_p:                                            # RandomShiftPos :755-758
        max_ = torch.FloatTensor([[shift_max[0], shift_max[1], shift_max[1]]])
        pos += (torch.rand(1, 3) * 2 * max_) - max_
    pos = move_center(pos, center[0], center[1])                             # MoveCenterPosPerSample (cz = 0.5)
    pos = start_z_from_zero(pos)                                             # StartZFromZero
    n_ori_points = len(pos)                                                  # AddRandomPoints :795-811
    if not n_ori_points >= n_max_points and add_p > random.random():
        ratio = random.random() * (add_ratio[1] - add_ratio[0]) + add_ratio[0]
        n_points = int(ratio * n_ori_points)
        n_points += np.amin([0, n_max_points - (n_ori_points + n_points)])
        min_ = pos.amin(0, keepdim=True)
        max_ = pos.amin(0, keepdim=True)
        random_points = (torch.rand(n_points, pos.shape[1]) * (max_ - min_) + min_)
        pos = torch.cat([pos, random_points], 0)
    n_ori_points = len(pos)                                                  # CopyJitterRandomPoints :845-869
    if not n_ori_points >= n_max_points and add_p > random.random():
        ratio = random.random() * (add_ratio[1] - add_ratio[0]) + add_ratio[0]
        n_points = int(ratio * n_ori_points)
        n_points += np.amin([0, n_max_points - (n_ori_points + n_points)])
        idx = np.random.choice(n_ori_points, size=n_points, replace=True)
        random_points = pos[idx].clone()
        noise = cj_sigma * torch.randn(random_points.shape)
        random_points += noise.clamp(-cj_clip, cj_clip)
        pos = torch.cat([pos, random_points], 0)
    pre_crop = pos
    polygon = list(polygons)[np.random.choice(len(polygons))]                # RandomPolygon2dExtend :1531-1543
    rand_scale = np.random.rand() * (1 - 1) + 1
    trans = (1 - rand_scale) / 2
    rand_rotate = np.random.rand() * rotate * np.sign(np.random.rand() - .5)
    A = Affine2D().scale(rand_scale).translate(trans, trans).rotate_deg_around(0.5, 0.5, rand_rotate)
    mask = torch.from_numpy(Path(polygon).transformed(A).contains_points(pos[:, [0, 1]].numpy()))
    src = torch.arange(len(pos))
    if mask.sum() > 0:
        pos, src = pos[mask], src[mask]
    return pos, src, pre_crop, orig
