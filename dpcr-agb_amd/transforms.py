"""The NFI sparse transform chain on the device, for a whole batch of plots at once.

The reference applies its transforms per sample on the CPU inside DataLoader workers
(torch-points3d/conf/data/instance/NFI/transforms/sparse-xy.yaml; classes in
torch_points3d/core/data_transform/{transforms,features,sparse_transforms,grid_transform}.py).  Here the same classes
(same names, same constructor arguments) are descriptors; ``SparsePlotPipeline`` fuses the per-point ones into
``agb_plot_prepare`` (csrc/transform.hip: scale, centre, z from zero, polygon crop, features) followed by
``agb_voxelize_last`` (GridSampling3D, csrc/voxelize.hip) and ``agb_coords_augment`` (RandomCoordsFlip, ShiftVoxels),
and returns the ``PlotBatch`` that ``MinkowskiBaselineModel.set_input`` consumes.

Randomness is drawn on the host with the same generators, in the same per-sample order as the reference
(``torch.randperm`` for MaxPoints / MinPoints / the GridSampling3D shuffle, ``random.random`` / ``torch.rand`` for the
coordinate flips and shifts), so a seeded run picks the same points and flips.  The float augmentations of the training
chain (ground removal, dropout, jitter, z-rotation, random points, random polygon) are not part of this module.
"""
import math
import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .kp_index import _elem_of_row, _ptr_tensor
from .voxelize import voxelize_last_begin, voxelize_last_end, device_permutations, draw_permutations, voxelize_last

_P = _lib.ptr
_V, _I = _lib.c_void_p, _lib.c_int
_lib.declare("agb_plot_workspace_bytes", [_I, _I])
_lib.declare("agb_plot_prepare_ws", [_V, _V, _V, _I, _I, _V, _I, _I, _V, _I, _V, _V, _V, _V, _V, _V, _V])
_lib.declare("agb_coords_augment", [_V, _V, _I, _I, _V, _V, _V, _V])

HEXAGON = [[0.0, 0.5], [0.25, 0.9330127], [0.75, 0.9330127], [1.0, 0.5], [0.75, 0.0669873], [0.25, 0.0669873]]


# ------------------------------------------------------------------------------------------------ descriptors
class ScalePos:
    def __init__(self, scale_x=1.0, scale_y=1.0, scale_z=1.0, op="mul"):
        self.scale, self.op_str = (float(scale_x), float(scale_y), float(scale_z)), op


class MoveCenterPosPerSample:
    def __init__(self, center_x: float = 0.5, center_y: float = 0.5, center_z: float = 0.5):
        self.center = (float(center_x), float(center_y), float(center_z))


class StartZFromZero:
    pass


class Polygon2dExtend:
    def __init__(self, polygon, skip_list: Optional[list] = None):
        self.polygon = [[float(a), float(b)] for a, b in polygon]
        self.skip_list = list(skip_list or [])


class MaxPoints:
    def __init__(self, num, skip_list: Optional[list] = None):
        self.num = int(num)


class MinPoints:
    def __init__(self, num, skip_list: Optional[list] = None):
        self.num = int(num)


class XYZFeature:
    def __init__(self, add_x=False, add_y=False, add_z=True):
        if add_x or add_y or not add_z:
            raise NotImplementedError("the NFI chains add pos_z only")


class AddOnes:
    pass


class AddXYDistanceToCenter:
    def __init__(self, center_x: float, center_y: float):
        self.center = (float(center_x), float(center_y))


class AddFeatsByKeys:
    def __init__(self, list_add_to_x, feat_names, input_nc_feats=None, stricts=None, delete_feats=None):
        if list(feat_names) != ["ones", "pos_z", "xy_distance"] or not all(list_add_to_x):
            raise NotImplementedError("the NFI chains build x = [ones, pos_z, xy_distance]")


class GridSampling3D:
    def __init__(self, size, quantize_coords=True, mode="last", verbose=False):
        if mode != "last" or not quantize_coords:
            raise NotImplementedError("only GridSampling3D(quantize_coords=True, mode='last') is implemented")
        self.size = float(size)


class RandomCoordsFlip:
    def __init__(self, ignored_axis, is_temporal=False, p=0.95):
        assert 0 <= p <= 1
        mapping = {"x": 0, "y": 1, "z": 2}
        self.axes = sorted(set(range(3)) - {mapping[a] for a in ignored_axis})   # CPython iterates {0,1,2} ascending
        self.p = p


class ShiftVoxels:
    def __init__(self, apply_shift=True, p=0.5):
        self.apply_shift, self.p = apply_shift, p


def nfi_test_transform(scale=(30.0, 30.0, 40.0), center=(0.5, 0.5), size=0.0125, max_points=16000, min_points=500):
    """sparse-xy.yaml test_transform with the values of conf/data/instance/NFI/default.yaml:18-23.  size = None: the point
    chain of the KPConv / PointNet models (xy.yaml:76-113: the same list without GridSampling3D, MaxPoints 6144)."""
    out = [ScalePos(*scale, op="div"), MoveCenterPosPerSample(*center), StartZFromZero(), Polygon2dExtend(HEXAGON),
           MaxPoints(max_points), MinPoints(min_points), XYZFeature(False, False, True), AddOnes(),
           AddXYDistanceToCenter(*center), AddFeatsByKeys([True] * 3, ["ones", "pos_z", "xy_distance"])]
    if size is not None:
        out.append(GridSampling3D(size, quantize_coords=True, mode="last"))
    return out


def nfi_coord_augmentation():
    """The two coordinate augmentations that end sparse-xy.yaml train_transform (:100-104)."""
    return [RandomCoordsFlip("z", p=0.5), ShiftVoxels()]


# ------------------------------------------------------------------------------------------------ pipeline
class SparsePlotPipeline:
    """Executes a transform list of the classes above on a batch of raw plots with fused device kernels."""

    def __init__(self, transforms: Sequence):
        self.scale, self.div = (1.0, 1.0, 1.0), 0
        self.center = (0.0, 0.0, 0.0)
        self.z0, self.polygon = False, None
        self.max_points = self.min_points = None
        self.feat_center, self.grid = None, None
        self.flip, self.shift = None, None
        stage = 0
        order = [ScalePos, MoveCenterPosPerSample, StartZFromZero, Polygon2dExtend, MaxPoints, MinPoints, XYZFeature,
                 AddOnes, AddXYDistanceToCenter, AddFeatsByKeys, GridSampling3D, RandomCoordsFlip, ShiftVoxels]
        for t in transforms:
            if type(t) not in order:
                raise NotImplementedError(f"{type(t).__name__} has no device implementation in this pipeline")
            if order.index(type(t)) < stage:
                raise NotImplementedError("transforms must come in the order of the NFI sparse chains")
            stage = order.index(type(t))
            if isinstance(t, ScalePos):
                self.scale, self.div = t.scale, int(t.op_str == "div")
            elif isinstance(t, MoveCenterPosPerSample):
                self.center = t.center
            elif isinstance(t, StartZFromZero):
                self.z0 = True
            elif isinstance(t, Polygon2dExtend):
                self.polygon = t.polygon
            elif isinstance(t, MaxPoints):
                self.max_points = t.num
            elif isinstance(t, MinPoints):
                self.min_points = t.num
            elif isinstance(t, AddXYDistanceToCenter):
                self.feat_center = t.center
            elif isinstance(t, GridSampling3D):
                self.grid = t
            elif isinstance(t, RandomCoordsFlip):
                self.flip = t
            elif isinstance(t, ShiftVoxels):
                self.shift = t
        if self.feat_center is None:
            raise NotImplementedError("the pipeline builds x = [ones, pos_z, xy_distance]: AddXYDistanceToCenter missing")

    # -- per-point stage -----------------------------------------------------------------------------------------
    def prepare(self, plots: List, device):
        """plots: list of float [n_i, 3] (numpy or torch).  Returns (pos [M,3], x [M,3], src int64 [M] rows of the
        stacked input, lengths int64 [B]) on the device, after the crop and MaxPoints / MinPoints."""
        lens = np.asarray([int(p.shape[0]) for p in plots], dtype=np.int64)
        B, n = len(plots), int(lens.sum())
        dev = torch.device(device)
        stacked = torch.cat([torch.as_tensor(p, dtype=torch.float32).reshape(-1, 3) for p in plots]).to(dev)
        ptr = _ptr_tensor(lens, dev)
        elem = _elem_of_row(ptr, B, n, dev)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)   # noqa: E731
        i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)      # noqa: E731
        nn_ = max(n, 1)
        pos_o, x_o = f32(nn_, 3), f32(nn_, 3)
        ws = torch.empty(_lib.size_call("agb_plot_workspace_bytes", n, B), dtype=torch.uint8, device=dev)
        src = torch.empty(nn_, dtype=torch.int64, device=dev)
        out_ptr, n_out = i32(B + 1), i32(1)
        xform = (_lib.c_float * 8)(*self.scale, *self.center, *self.feat_center)
        poly = None
        nv = 0
        if self.polygon is not None:
            poly = torch.tensor(self.polygon, dtype=torch.float64).reshape(-1).to(dev)
            nv = len(self.polygon)
        _lib.call("agb_plot_prepare_ws", _P(stacked), _P(ptr), _P(elem), B, n, xform, self.div, int(self.z0), _P(poly), nv,
                  _P(ws), _P(pos_o), _P(x_o), _P(src), _P(out_ptr), _P(n_out), _lib.stream())
        return self.fix_counts(pos_o, x_o, src, out_ptr)

    def fix_counts_begin(self, pos_o, out_ptr, with_extent=False):
        """The device tensor ``fix_counts_end`` needs on the host: the row offsets after the crop and — with_extent — the
        per-axis extent of the widest cloud in voxel cells (what ``voxelize_last`` sizes its cell grid from), so that ONE host
        read serves both."""
        if with_extent and self.grid is not None:
            from .kp_index import elem_bbox
            B = int(out_ptr.shape[0]) - 1
            bb = elem_bbox(pos_o, out_ptr, B)
            ext = (bb[:, 3:] - bb[:, :3]).max(0).values / float(np.float32(self.grid.size))
            return torch.cat([out_ptr.double(), ext.double()])
        return out_ptr

    def fix_counts_end(self, pos_o, x_o, src, host, B):
        """MaxPoints / MinPoints on the cropped rows; host: the values of ``fix_counts_begin``'s tensor (a list)."""
        dev = pos_o.device
        self.grid_extent = None
        optr = np.asarray(host[:B + 1], dtype=np.int64)
        if len(host) > B + 1 and all(np.isfinite(host[B + 1:])):
            self.grid_extent = [float(v) for v in host[B + 1:]]
        new_lens = np.diff(optr)
        m = int(optr[-1])
        pos_o, x_o, src = pos_o[:m], x_o[:m], src[:m]
        # MaxPoints / MinPoints: per plot, only when a plot is out of range (torch.randperm like FixedPointsOwn)
        need = [(self.max_points is not None and c > self.max_points) or
                (self.min_points is not None and 0 < c < self.min_points) for c in new_lens]
        if any(need):
            choice, out_lens = [], []
            for b, c in enumerate(new_lens):
                c = int(c)
                idx = torch.arange(c)
                if self.max_points is not None and c > self.max_points:
                    idx = torch.randperm(c)[:self.max_points]
                if self.min_points is not None and 0 < len(idx) < self.min_points:
                    k = len(idx)
                    idx = idx[torch.cat([torch.randperm(k) for _ in range(math.ceil(self.min_points / k))])
                              [:self.min_points]]
                choice.append(idx + int(optr[b]))
                out_lens.append(len(idx))
            choice = torch.cat(choice).to(dev)
            pos_o, x_o, src = pos_o[choice], x_o[choice], src[choice]
            new_lens = np.asarray(out_lens, dtype=np.int64)
        return pos_o, x_o, src, new_lens

    def fix_counts(self, pos_o, x_o, src, out_ptr, with_extent=False):
        """MaxPoints / MinPoints on the cropped rows (out_ptr: device int32 [B+1] offsets).
        with_extent: the per-axis extent of the widest cloud in voxel cells — what ``voxelize_last`` sizes its cell grid
        from — is computed on the device and comes back in the SAME host read as the lengths (self.grid_extent; MaxPoints /
        MinPoints keep a subset / repeat rows: the extent stays an upper bound): one synchronisation less per batch."""
        B = int(out_ptr.shape[0]) - 1
        host = self.fix_counts_begin(pos_o, out_ptr, with_extent).tolist()      # ONE host read: lengths (+ extent)
        return self.fix_counts_end(pos_o, x_o, src, host, B)

    # -- whole chain ---------------------------------------------------------------------------------------------
    def __call__(self, plots: List, device, y_reg=None, perms=None):
        """Returns a PlotBatch (batch, coords, x, pos, y_reg, ...) resident on `device`."""
        pos, x, src, lens = self.prepare(plots, device)
        return self.finish(pos, x, src, lens, len(plots), y_reg=y_reg, perms=perms)

    def finish(self, pos, x, src, lens, B, y_reg=None, perms=None, extent_hint=None):
        """GridSampling3D + coordinate augmentation + batch assembly on prepared rows."""
        gen = self.finish_staged(pos, x, src, lens, B, y_reg=y_reg, perms=perms, extent_hint=extent_hint)
        try:
            next(gen)
        except StopIteration as done:
            return done.value
        raise RuntimeError("finish_staged yielded without a reader")

    def finish_staged(self, pos, x, src, lens, B, y_reg=None, perms=None, extent_hint=None, reader=None):
        """``finish`` as a generator: with a `reader` (device tensor -> pending read) it YIELDS the pending read of the
        voxel counts instead of waiting for it, and is resumed with the values (``gen.send(values)``); the PlotBatch is the
        generator's return value.  Without a reader it never yields (what ``finish`` runs)."""
        from .synthetic import PlotBatch
        dev = pos.device
        if self.grid is None:
            # a point batch (xy.yaml: KPConv / PointNet models): rows, features, plot of every row — built on the device
            lens_np = np.asarray(lens, dtype=np.int64)
            ptr = _ptr_tensor(lens_np, dev)
            m = int(lens_np.sum())
            batch = _elem_of_row(ptr, B, m, dev)[:m].to(torch.int64)
            out = PlotBatch(batch, None, x, pos, None, None, B, None)
            out.src = src
            out.host_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens_np)]))   # (data.ptr without a read-back)
        else:
            perm = seed = None
            if perms is not None:
                perm = perms
            elif getattr(self, "device_shuffle", False):
                # the voxel shuffle drawn INSIDE the voxeliser from one seed of torch's host generator (a keyed bijection per
                # cloud: csrc/voxelize.hip vox_perm) — no permutation tensor, no sort (round 5: voxelize.device_permutations,
                # 2 ms of host time per batch of 32 plots in torch.argsort / randint / cumsum launches)
                seed = int(torch.randint(0, 1 << 62, (1,)).item())
            else:
                perm = draw_permutations(lens)
            augmented = self.flip is not None or self.shift is not None
            state, to_read = voxelize_last_begin(pos, lens, self.grid.size, perm=perm, extent_hint=extent_hint, seed=seed,
                                                 cloud_boxes=augmented)
            if to_read is None:
                host = None
            elif reader is not None:       # a pipelined caller: the read-back is started here and finished a step later
                host = yield reader(to_read)
            else:
                host = to_read.tolist()    # one host read: new lengths + coordinate bounds
            if augmented:
                coords, keep, vlens, bounds, boxes = voxelize_last_end(state, host, cloud_boxes=True)
            else:
                coords, keep, vlens, bounds = voxelize_last_end(state, host)
            coords = coords.contiguous()
            ptr = _ptr_tensor(vlens, dev)
            m = int(coords.shape[0])
            elem = _elem_of_row(ptr, B, m, dev)
            batch = elem[:m].to(torch.int64)           # (on the device: no 8-byte-per-row upload)
            if self.flip is not None or self.shift is not None:
                flips, shifts = np.zeros((B, 3), np.int32), np.zeros((B, 3), np.int32)
                for b in range(B):   # the reference draws per sample: flip axes first, then the shift
                    if self.flip is not None:
                        for ax in self.flip.axes:
                            if random.random() < self.flip.p:
                                flips[b, ax] = 1
                    if self.shift is not None and self.shift.apply_shift and random.random() < self.shift.p:
                        shifts[b] = (torch.rand(3) * 100).to(torch.int32).numpy()
                cmax = torch.empty(3 * B, dtype=torch.int32, device=dev)
                # named tensors: a temporary would be freed (and its block reused) as soon as its pointer is taken
                # (through the pinned ring: a copy from pageable memory makes the runtime wait for the whole device)
                from .kp_index import h2d_small
                flips_d, shifts_d = h2d_small(flips, dev), h2d_small(shifts, dev)
                _lib.call("agb_coords_augment", _P(coords), _P(elem), B, m, _P(flips_d), _P(shifts_d), _P(cmax),
                          _lib.stream())
                # The EXACT box of the augmented coordinates without reading them back: every cloud's own box came with the
                # voxeliser's read; a flipped axis of cloud b becomes max_b - c, i.e. [0, max_b - min_b]; the shift is known
                # here.  (Round 5 used the batch-wide box for every cloud: up to 1.5x wider per flipped axis, and the
                # coordinate manager sizes the dense lookup grid of every level from it.)
                live = np.asarray(vlens) > 0
                lo_c, hi_c = boxes[:, :3], boxes[:, 3:]
                lo_b = np.where(flips != 0, 0, lo_c) + shifts
                hi_b = np.where(flips != 0, hi_c - lo_c, hi_c) + shifts
                if live.any():
                    bounds = tuple(int(v) for v in lo_b[live].min(0)) + tuple(int(v) for v in hi_b[live].max(0))
            out = PlotBatch(batch, coords, x[keep], pos[keep], None, None, B, bounds)
            out.src = src[keep]
        if y_reg is not None:
            from .kp_index import h2d_small
            yr = y_reg.detach().cpu().numpy() if torch.is_tensor(y_reg) else np.asarray(y_reg)
            out.y_reg = h2d_small(np.ascontiguousarray(yr, dtype=np.float32), dev)
            out.y_reg_mask = torch.ones_like(out.y_reg, dtype=torch.bool)
            out.y_reg_mask_all = True
        return out
