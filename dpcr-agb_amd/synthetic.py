"""Seeded synthetic LiDAR forest plots and the batch container handed to ``set_input``.

The NFI point clouds are not available offline, so benches and parity tests run on plots drawn here:
normalised frame of the reference (x, y in the hexagon inscribed in [0,1]^2 — vertices
conf/data/instance/NFI/transforms/sparse-xy.yaml:60-64 —, z = height / 40 m), 30 % ground returns,
70 % canopy returns from 15-60 trees, labels biomass = a * sum(h^b) (+ noise) and volume = 1.87 * biomass.
Features are the reference's ``[ones, pos_z, xy_distance]`` (sparse-xy.yaml:72-94).

``PlotBatch`` carries the fields ``MinkowskiBaselineModel.set_input`` reads from a torch_geometric Batch
(models/instance/minkowski.py:67-80): batch, coords, x, pos, y_reg, y_reg_mask, len() = number of plots.
"""
from typing import List, Optional

import numpy as np
import torch

HEXAGON = np.array([[0.0, 0.5], [0.25, 0.9330127], [0.75, 0.9330127], [1.0, 0.5], [0.75, 0.0669873],
                    [0.25, 0.0669873]], dtype=np.float64)
BIOMASS_A, BIOMASS_B, VOLUME_RATIO = 0.035, 2.2, 1.87


def _in_hexagon(xy):
    inside = np.ones(len(xy), dtype=bool)
    for i in range(6):
        a, b = HEXAGON[i], HEXAGON[(i + 1) % 6]
        cross = (b[0] - a[0]) * (xy[:, 1] - a[1]) - (b[1] - a[1]) * (xy[:, 0] - a[0])
        inside &= cross <= 0  # vertices are listed clockwise
    return inside


def make_plot(seed: int, n_points: int = 16000, extra_feature: bool = False):
    """One plot: pos float32 [n,3] (normalised), x float32 [n,F], y float32 [2] = (biomass, volume)."""
    rng = np.random.default_rng(seed)
    n_draw = int(n_points * 1.6) + 64
    n_trees = int(rng.integers(15, 61))
    centres = rng.uniform(0, 1, size=(n_trees, 2))
    heights = rng.uniform(8.0, 30.0, size=n_trees)       # m
    radii = rng.uniform(1.5, 4.0, size=n_trees) / 30.0   # normalised xy units
    is_ground = rng.uniform(size=n_draw) < 0.3
    pos = np.empty((n_draw, 3), dtype=np.float64)
    ng = int(is_ground.sum())
    pos[is_ground, :2] = rng.uniform(0, 1, size=(ng, 2))
    pos[is_ground, 2] = np.abs(rng.normal(0, 0.3, size=ng)) / 40.0
    nc = n_draw - ng
    tree = rng.integers(0, n_trees, size=nc)
    pos[~is_ground, :2] = centres[tree] + rng.normal(0, 1, size=(nc, 2)) * (radii[tree] / 2)[:, None]
    pos[~is_ground, 2] = heights[tree] * (1 - 0.5 * rng.beta(2, 3, size=nc)) / 40.0
    keep = _in_hexagon(pos[:, :2])
    pos = pos[keep][:n_points].astype(np.float32)
    biomass = BIOMASS_A * float(np.sum(heights ** BIOMASS_B)) * (1 + 0.05 * rng.normal())
    volume = VOLUME_RATIO * biomass * (1 + 0.03 * rng.normal())
    feats = [np.ones(len(pos), np.float32), pos[:, 2], np.sqrt((pos[:, 0] - 0.5) ** 2 + (pos[:, 1] - 0.5) ** 2)]
    if extra_feature:
        feats.append(rng.uniform(0, 1, size=len(pos)).astype(np.float32))
    x = np.stack(feats, 1).astype(np.float32)
    return pos, x, np.array([biomass, volume], dtype=np.float32)


def voxelize_host(pos: np.ndarray, perm: np.ndarray, size: float):
    """Host-side (numpy) GridSampling3D(mode='last', quantize_coords=True) used only to PREPARE synthetic
    batches — the device kernel for this step is agb_voxelize_last.  Follows
    core/data_transform/grid_transform.py:112-128: shuffle by ``perm``, coords = round(pos / size) in fp32
    (half-to-even), voxel key with x fastest relative to the per-plot minimum, sorted unique, keep the LAST
    shuffled point of every voxel.  Returns (coords int32 [M,3], keep: indices into the ORIGINAL point order)."""
    p = pos[perm].astype(np.float32)
    c = np.rint(p / np.float32(size)).astype(np.float32)
    lo = c.min(0)
    span = (np.floor(c.max(0) - lo) + 1).astype(np.int64)
    rel = np.floor(c - lo).astype(np.int64)
    key = rel[:, 0] + span[0] * (rel[:, 1] + span[1] * rel[:, 2])
    uniq, inv = np.unique(key, return_inverse=True)
    last = np.empty(len(uniq), dtype=np.int64)
    last[inv] = np.arange(len(key))  # later writes win -> last occurrence
    return c[last].astype(np.int32), perm[last]


class PlotBatch:
    """Stand-in for the torch_geometric ``Batch`` the reference feeds to ``model.set_input``."""

    def __init__(self, batch, coords, x, pos, y_reg=None, y_reg_mask=None, num_plots=None, coord_bounds=None):
        self.batch, self.coords, self.x, self.pos = batch, coords, x, pos
        self.y_reg, self.y_reg_mask = y_reg, y_reg_mask
        self._n = int(num_plots) if num_plots is not None else int(batch.max().item()) + 1
        # (min_x, min_y, min_z, max_x, max_y, max_z) of the voxel coordinates: a by-product of voxelisation
        if coord_bounds is None and coords is not None and len(coords) > 0:
            coord_bounds = tuple(coords.min(0).values.tolist()) + tuple(coords.max(0).values.tolist())
        self.coord_bounds = coord_bounds
        self.y_reg_mask_all = bool(y_reg_mask.all()) if y_reg_mask is not None else None
        self._prefetched = None

    def __len__(self):
        return self._n

    def __contains__(self, k):
        return getattr(self, k, None) is not None

    def __getitem__(self, k):
        return getattr(self, k)

    def to(self, device, non_blocking=True):
        # (a device -> host copy that returns before it has landed hands the caller garbage: only uploads are asynchronous)
        non_blocking = bool(non_blocking) and torch.device(device).type != "cpu"
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)  # noqa: E731
        out = PlotBatch(mv(self.batch), mv(self.coords), mv(self.x), mv(self.pos), None, None, self._n,
                        self.coord_bounds)
        out.y_reg, out.y_reg_mask, out.y_reg_mask_all = mv(self.y_reg), mv(self.y_reg_mask), self.y_reg_mask_all
        for extra in ("pos_bounds", "area_name", "host_ptr"):
            if hasattr(self, extra):
                setattr(out, extra, getattr(self, extra))
        return out

    @property
    def ptr(self):
        if getattr(self, "host_ptr", None) is not None:     # (the device pipelines know the row counts on the host)
            return self.host_ptr
        counts = torch.bincount(self.batch.cpu(), minlength=self._n)
        return torch.cat([counts.new_zeros(1), counts.cumsum(0)])


def plot_points(seed: int, density) -> int:
    """Returns per plot that follow the stand: base + per_tree * n_trees (n_trees = the generator's first draw)."""
    base, per_tree = density
    return int(base) + int(per_tree) * int(np.random.default_rng(seed).integers(15, 61))


def make_sparse_batch(seeds: List[int], n_points: int = 16000, size: float = 0.0125, extra_feature: bool = False,
                      perm_seed: Optional[int] = None, density=None) -> PlotBatch:
    """Voxelised batch (test-time transform chain: no augmentation) for the sparse models.
    density = (base, per_tree): every plot gets base + per_tree * n_trees points instead of n_points (denser stands
    return more canopy echoes; NFI plots differ in size the same way)."""
    bs, cs, xs, ps, ys = [], [], [], [], []
    for b, seed in enumerate(seeds):
        pos, x, y = make_plot(seed, n_points if density is None else plot_points(seed, density), extra_feature)
        prng = np.random.default_rng((seed if perm_seed is None else perm_seed) + 7919)
        perm = prng.permutation(len(pos))
        coords, keep = voxelize_host(pos, perm, size)
        bs.append(np.full(len(keep), b, dtype=np.int64))
        cs.append(coords)
        xs.append(x[keep])
        ps.append(pos[keep])
        ys.append(y)
    return PlotBatch(torch.from_numpy(np.concatenate(bs)), torch.from_numpy(np.concatenate(cs)),
                     torch.from_numpy(np.concatenate(xs)), torch.from_numpy(np.concatenate(ps)),
                     torch.from_numpy(np.stack(ys)), torch.ones(len(seeds), 2, dtype=torch.bool), len(seeds))


def make_point_batch(seeds: List[int], n_points: int = 6144, extra_feature: bool = False) -> PlotBatch:
    """Raw (un-voxelised) batch for the KPConv path (xy.yaml caps plots at 6144 points)."""
    bs, xs, ps, ys = [], [], [], []
    for b, seed in enumerate(seeds):
        pos, x, y = make_plot(seed, n_points, extra_feature)
        bs.append(np.full(len(pos), b, dtype=np.int64))
        xs.append(x)
        ps.append(pos)
        ys.append(y)
    allp = np.concatenate(ps)
    out = PlotBatch(torch.from_numpy(np.concatenate(bs)), None, torch.from_numpy(np.concatenate(xs)),
                    torch.from_numpy(allp), torch.from_numpy(np.stack(ys)),
                    torch.ones(len(seeds), 2, dtype=torch.bool), len(seeds))
    # bounding box of the positions, known for free where the batch is assembled (saves the KPConv input pyramid its one
    # bounding-box read-back); slightly widened so that float32 round trips cannot put a point outside
    lo, hi = allp.min(0).astype(np.float64), allp.max(0).astype(np.float64)
    cloud_diag = max(float(np.linalg.norm(p.max(0).astype(np.float64) - p.min(0).astype(np.float64))) for p in ps)
    out.pos_bounds = tuple(lo - 1e-5) + tuple(hi + 1e-5) + (cloud_diag + 1e-4,)   # (7th: largest single-plot diagonal)
    return out


class SyntheticDataset:
    """The dataset attributes the reference's models read (models/instance/base.py:55-134,
    minkowski.py:33, kpconv.py:50) with target statistics estimated from the generator."""

    def __init__(self, feature_dimension: int = 3, num_points: int = 16000, stat_seeds=range(10_000, 10_256)):
        from .config import NFI_TARGETS, Opt
        self.feature_dimension = feature_dimension
        self.num_classes = 2
        self.num_reg_classes = 2
        self.has_reg_targets = True
        self.reg_targets_idx = np.array([True, True])
        self.targets = NFI_TARGETS
        self.reg_targets = list(NFI_TARGETS.keys())
        self.areas = {"synthetic": None}
        self.double_batch = False
        self.dataset_opt = Opt(fixed=Opt(num_points=num_points))
        ys = np.stack([self._labels(s) for s in stat_seeds])
        self._stats = {"mean": ys.mean(0), "std": ys.std(0), "min": ys.min(0), "max": ys.max(0)}

    @staticmethod
    def _labels(seed):
        # labels depend only on the first draws of make_plot's generator: replay them cheaply
        rng = np.random.default_rng(seed)
        n_trees = int(rng.integers(15, 61))
        rng.uniform(0, 1, size=(n_trees, 2))
        heights = rng.uniform(8.0, 30.0, size=n_trees)
        return np.array([BIOMASS_A * float(np.sum(heights ** BIOMASS_B)),
                         VOLUME_RATIO * BIOMASS_A * float(np.sum(heights ** BIOMASS_B))], dtype=np.float64)

    def _get(self, stat):
        # one area, the same statistics for every stage (the tracker reads [area][stage], instance_tracker.py:69-87)
        v = self._stats[stat]
        per_stage = {"train": v, "val": v, "test": v}
        return {"synthetic": dict(per_stage), "total": dict(per_stage)}

    def get_mean_targets(self):
        return self._get("mean")

    def get_std_targets(self):
        return self._get("std")

    def get_min_targets(self):
        return self._get("min")

    def get_max_targets(self):
        return self._get("max")
