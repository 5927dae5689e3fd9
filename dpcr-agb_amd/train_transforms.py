"""The TRAIN transform chain of the NFI sparse models on the device (sparse-xy.yaml:4-104), for a whole batch.

Every random decision is drawn on the host with the generators the reference uses (``random``, ``numpy.random``,
``torch``), per sample and in the reference's call order inside a sample (``draw_sample``); the device then applies the
draws to all plots at once (csrc/transform.hip: ``agb_plot_augment``, ``agb_plot_extend``, ``agb_plot_crop``) and hands over
to the deterministic tail shared with the test chain (MaxPoints / MinPoints, feature build, GridSampling3D,
RandomCoordsFlip, ShiftVoxels: ``transforms.SparsePlotPipeline``).

Across samples the draw order differs from a single-process reference run: the reference finishes the whole chain of
sample i (including MaxPoints, the voxel shuffle, flips and shifts) before it touches sample i+1, here the
count-dependent tail of all samples is drawn after the batch's crop has been counted on the device.  Each transform's
distribution is the reference's; a globally seeded run is reproducible but not draw-for-draw the reference's (with
DataLoader workers the reference's own stream order depends on the worker count, too).
"""
import math
import random
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch
from matplotlib.path import Path
from matplotlib.transforms import Affine2D

from . import _lib
from .kp_index import _elem_of_row, _ptr_tensor
from .transforms import HEXAGON, SparsePlotPipeline, nfi_coord_augmentation, nfi_test_transform

_P = _lib.ptr
_V, _I, _F = _lib.c_void_p, _lib.c_int, _lib.c_float
_lib.declare("agb_plot_augment", [_V, _V, _V, _V, _I, _I, _V, _V, _V, _V, _V])
_lib.declare("agb_plot_extend", [_V, _V, _V, _V, _V, _I, _I, _V, _V, _V, _V, _V, _V])
_lib.declare("agb_plot_crop_ws", [_V, _V, _V, _I, _I, _V, _I, _F, _F, _V, _V, _V, _V, _V, _V, _V])


@dataclass
class NFITrainConfig:
    """Values of conf/data/instance/NFI/transforms/sparse-xy.yaml:4-69 and default.yaml:18-23."""
    ground_min_v: float = 0.05
    ground_max_v: float = 0.5
    ground_p: float = 0.1
    ground_min_points: int = 500
    dropout_ratio: float = 0.2
    dropout_application_ratio: float = 0.5
    dropout_min_points: int = 500
    scale: tuple = (30.0, 30.0, 40.0)
    noise_sigma: float = 0.0025
    noise_clip: float = 0.05
    rot_deg: tuple = (0.0, 0.0, 180.0)
    shift_p: float = 0.5
    shift_max: tuple = (0.01, 0.01, 0.0)
    center: tuple = (0.5, 0.5, 0.5)
    add_n_max: int = 12000
    add_ratio: tuple = (0.01, 0.2)
    add_p: float = 0.25
    cj_n_max: int = 12000
    cj_ratio: tuple = (0.01, 0.2)
    cj_p: float = 0.25
    cj_sigma: float = 0.005
    cj_clip: float = 0.015
    polygons: list = field(default_factory=lambda: [HEXAGON])
    polygon_rotate: float = 180.0
    polygon_size: tuple = (1.0, 1.0)
    voxel: Optional[float] = 0.0125       # None: the point chain (xy.yaml: no GridSampling3D / flip / shift tail)
    max_points: int = 16000               # sparse-xy.yaml:70 (xy.yaml:70: 6144)
    min_points: int = 500


def rotation_matrix(thetas: torch.Tensor) -> torch.Tensor:
    """torch_points3d/utils/geometry.py:5-22 with random_order=True (one ``random.shuffle`` of [R_x, R_y, R_z])."""
    c, s = torch.cos, torch.sin
    r_x = torch.tensor([[1, 0, 0], [0, c(thetas[0]), -s(thetas[0])], [0, s(thetas[0]), c(thetas[0])]])
    r_y = torch.tensor([[c(thetas[1]), 0, s(thetas[1])], [0, 1, 0], [-s(thetas[1]), 0, c(thetas[1])]])
    r_z = torch.tensor([[c(thetas[2]), -s(thetas[2]), 0], [s(thetas[2]), c(thetas[2]), 0], [0, 0, 1]])
    mats = [r_x, r_y, r_z]
    random.shuffle(mats)
    return torch.mm(mats[2], torch.mm(mats[1], mats[0]))


def draw_sample(raw: torch.Tensor, cfg: NFITrainConfig) -> dict:
    """All random draws of one sample for RandomGroundRemoval .. RandomPolygon2dExtend, in the reference's order.
    raw: float32 [n, 3] on the host (the ground-removal decision needs the raw heights)."""
    d = dict(zsub=0.0, n_raw=len(raw))
    sel = None
    # RandomGroundRemoval (transforms.py:1140-1150)
    if random.random() < cfg.ground_p:
        remove_v = random.random() * (cfg.ground_max_v - cfg.ground_min_v) + cfg.ground_min_v
        cond = raw[:, 2] > remove_v
        if int(cond.sum()) >= cfg.ground_min_points:
            d["zsub"] = remove_v
            sel = torch.nonzero(cond).reshape(-1)
    # RandomDropout (transforms.py:1078-1082 -> FixedPointsOwn :1337-1350)
    n = len(sel) if sel is not None else len(raw)
    if n > cfg.dropout_min_points and random.random() < cfg.dropout_application_ratio:
        num = int(n * (1 - cfg.dropout_ratio))
        choice = torch.cat([torch.randperm(n) for _ in range(math.ceil(num / n))], dim=0)[:num]
        sel = choice if sel is None else sel[choice]
    d["sel"] = torch.arange(len(raw)) if sel is None else sel
    n1 = len(d["sel"])
    # RandomNoise (transforms.py:498-503; p = 1 still consumes one random.random())
    d["noise"] = None
    if random.random() < 1:
        d["noise"] = (cfg.noise_sigma * torch.randn(n1, 3)).clamp(-cfg.noise_clip, cfg.noise_clip)
    # Random3AxisRotation (features.py:44-51)
    thetas = torch.zeros(3, dtype=torch.float)
    for axis, deg in enumerate(cfg.rot_deg):
        deg = abs(min(deg, 180)) if deg else 0
        if deg > 0 and random.random() < 1:
            rand_deg = random.random() * 2 * deg - deg
            thetas[axis] = float(rand_deg * np.pi) / 180.0
    d["M"] = rotation_matrix(thetas)
    # RandomShiftPos (transforms.py:755-758; upstream uses max_y for the z component)
    d["shift"] = None
    if random.random() > cfg.shift_p:
        max_ = torch.FloatTensor([[cfg.shift_max[0], cfg.shift_max[1], cfg.shift_max[1]]])
        d["shift"] = (torch.rand(1, 3) * 2 * max_) - max_
    # AddRandomPoints (transforms.py:795-811)
    d["n_add"] = 0
    if n1 < cfg.add_n_max and cfg.add_p > random.random():
        ratio = random.random() * (cfg.add_ratio[1] - cfg.add_ratio[0]) + cfg.add_ratio[0]
        n_points = int(ratio * n1)
        n_points += int(np.amin([0, cfg.add_n_max - (n1 + n_points)]))
        torch.rand(n_points, 3)          # consumed; multiplied by (max_ - min_) == 0 upstream
        d["n_add"] = n_points
    # CopyJitterRandomPoints (transforms.py:845-869)
    n2 = n1 + d["n_add"]
    d["cj_idx"], d["cj_noise"] = None, None
    if n2 < cfg.cj_n_max and cfg.cj_p > random.random():
        ratio = random.random() * (cfg.cj_ratio[1] - cfg.cj_ratio[0]) + cfg.cj_ratio[0]
        n_points = int(ratio * n2)
        n_points += int(np.amin([0, cfg.cj_n_max - (n2 + n_points)]))
        d["cj_idx"] = torch.from_numpy(np.random.choice(n2, size=n_points, replace=True)).long()
        d["cj_noise"] = (cfg.cj_sigma * torch.randn(n_points, 3)).clamp(-cfg.cj_clip, cfg.cj_clip)
    # RandomPolygon2dExtend (transforms.py:1531-1541)
    polygon = cfg.polygons[np.random.choice(len(cfg.polygons))]
    rand_scale = np.random.rand() * (cfg.polygon_size[1] - cfg.polygon_size[0]) + cfg.polygon_size[0]
    trans = (1 - rand_scale) / 2
    rand_rotate = np.random.rand() * cfg.polygon_rotate * np.sign(np.random.rand() - .5)
    A = Affine2D().scale(rand_scale).translate(trans, trans).rotate_deg_around(0.5, 0.5, rand_rotate)
    d["polygon"] = np.asarray(Path(polygon).transformed(A).vertices, dtype=np.float64)
    return d


def collate_draws(draws: List[dict], cfg: Optional[NFITrainConfig] = None) -> dict:
    """The draws of a batch as a handful of stacked HOST tensors (what ``SparseTrainPipeline.augment`` uploads): row
    selections as indices into the stacked raw plots, per-sample parameter rows, noise tables, counts.  Runs wherever the
    draws were made — as the ``collate_fn`` of a DataLoader it runs in the worker process, and the batch crosses to the
    training process as ten tensors instead of a hundred small ones."""
    c = cfg or NFITrainConfig()
    B = len(draws)
    n_raw = np.asarray([int(d["n_raw"]) for d in draws], dtype=np.int64)
    raw_off = np.concatenate([[0], np.cumsum(n_raw)])
    n1s = np.asarray([len(d["sel"]) for d in draws], dtype=np.int64)
    aug = torch.zeros(B, 24, dtype=torch.float32)
    for b, d in enumerate(draws):
        sh = d["shift"].reshape(-1).tolist() if d["shift"] is not None else [0.0, 0.0, 0.0]
        aug[b, :19] = torch.tensor([np.float32(d["zsub"]), *c.scale, *d["M"].reshape(-1).tolist(), *sh, *c.center],
                                   dtype=torch.float32)
    nv = len(draws[0]["polygon"])
    if any(len(d["polygon"]) != nv for d in draws):
        raise NotImplementedError("polygons of one batch must have the same number of vertices")
    n_cj = np.asarray([0 if d["cj_idx"] is None else len(d["cj_idx"]) for d in draws], dtype=np.int64)
    return dict(
        B=B, nv=nv, n_raw=torch.from_numpy(n_raw), n1s=torch.from_numpy(n1s),
        sel=torch.cat([d["sel"].long() + int(raw_off[b]) for b, d in enumerate(draws)]), aug=aug,
        noise=torch.cat([d["noise"] if d["noise"] is not None else torch.zeros(len(d["sel"]), 3) for d in draws]),
        n_add=torch.from_numpy(np.asarray([d["n_add"] for d in draws], dtype=np.int32)), n_cj=torch.from_numpy(n_cj),
        cj_idx=torch.cat([d["cj_idx"] if d["cj_idx"] is not None else torch.zeros(0, dtype=torch.long)
                          for d in draws] + [torch.zeros(1, dtype=torch.long)]),
        cj_noise=torch.cat([d["cj_noise"] if d["cj_noise"] is not None else torch.zeros(0, 3)
                            for d in draws] + [torch.zeros(1, 3)]),
        polys=torch.from_numpy(np.stack([d["polygon"] for d in draws]).reshape(B, -1)))


class SampleDraws(torch.utils.data.Dataset):
    """The per-sample random draws as a map-style dataset, so that they run where the reference runs its transforms: in the
    worker processes of a ``torch.utils.data.DataLoader`` (torch_points3d/datasets/base_dataset.py: the loaders are built
    with ``num_workers`` from the training config; every worker has its own ``random`` / numpy / torch generator state, seeded
    by the loader).  Item i = ``draw_sample`` of raw plot ``i % len(raws)``; collate with ``collate_draws`` (in the worker):

        loader = DataLoader(SampleDraws(raws), batch_size=B, num_workers=4, collate_fn=collate_draws, pin_memory=True,
                            worker_init_fn=SampleDraws.seed_worker, persistent_workers=True)
        for draws in loader: batch = pipeline(raws_on_device[...], device, draws=draws)
    """

    def __init__(self, raws, cfg: Optional[NFITrainConfig] = None, length: Optional[int] = None):
        self.raws = [torch.as_tensor(r, dtype=torch.float32).reshape(-1, 3) for r in raws]
        self.cfg = cfg or NFITrainConfig()
        self.length = len(self.raws) if length is None else int(length)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        return draw_sample(self.raws[i % len(self.raws)], self.cfg)

    @staticmethod
    def seed_worker(worker_id):
        # the loader seeds ``torch`` and ``random`` per worker; numpy's global generator (np.random.choice / rand in
        # CopyJitterRandomPoints and RandomPolygon2dExtend) would otherwise be a copy of the parent's in every worker
        np.random.seed((torch.initial_seed() + worker_id) % (1 << 32))
        torch.set_num_threads(1)


class SparseTrainPipeline:
    """sparse-xy.yaml train_transform for a batch of raw plots -> PlotBatch on the device."""

    def __init__(self, cfg: Optional[NFITrainConfig] = None, device_shuffle: bool = False):
        """device_shuffle: draw the voxel shuffle of GridSampling3D on the device (``voxelize.device_permutations``) instead
        of one host ``torch.randperm`` per plot."""
        self.cfg = cfg or NFITrainConfig()
        c = self.cfg
        # the deterministic tail (counts, features, voxelisation, coordinate augmentation) is the test pipeline's
        self.tail = SparsePlotPipeline(nfi_test_transform(c.scale, c.center[:2], c.voxel, c.max_points, c.min_points) +
                                       (nfi_coord_augmentation() if c.voxel is not None else []))
        self.tail.device_shuffle = bool(device_shuffle)

    def augment(self, plots: List, draws, device):
        """Applies the per-sample draws (a list of ``draw_sample`` dicts or their ``collate_draws``) on the device.  Returns
        (pos [M,3], x [M,3], src [M], out_ptr int32 [B+1]): the rows kept by RandomPolygon2dExtend, their features, their
        position in the pre-crop stacking."""
        c = self.cfg
        dev = torch.device(device)
        B = len(plots)
        if not isinstance(draws, dict):
            draws = collate_draws(draws, c)
        ready = all(torch.is_tensor(p) and p.device == dev and p.dtype == torch.float32 and p.dim() == 2 for p in plots)
        if not ready:
            plots = [torch.as_tensor(p, dtype=torch.float32).reshape(-1, 3).to(dev) for p in plots]
        if draws["B"] != B or draws["n_raw"].tolist() != [p.shape[0] for p in plots]:
            raise ValueError("the draws were made for other plots than the ones passed")
        up = lambda t: t.to(dev, non_blocking=True)     # noqa: E731  (pinned by the loader: asynchronous; else blocking)
        raw = torch.cat(plots)
        sel, aug, noise = up(draws["sel"]), up(draws["aug"]), up(draws["noise"])
        n1s = draws["n1s"].numpy()
        n1 = int(n1s.sum())
        ptr1 = _ptr_tensor(n1s, dev)
        elem1 = _elem_of_row(ptr1, B, n1, dev)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)   # noqa: E731
        i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)      # noqa: E731
        pos1, mins = f32(max(n1, 1), 3), f32(3 * B)
        _lib.call("agb_plot_augment", _P(raw), _P(sel), _P(elem1), _P(ptr1), B, n1, _P(aug), _P(noise), _P(pos1),
                  _P(mins), _lib.stream())
        # StartZFromZero + AddRandomPoints + CopyJitterRandomPoints: all counts are known to the host
        n_cj = draws["n_cj"].numpy()
        n2s = n1s + draws["n_add"].numpy() + n_cj
        n2 = int(n2s.sum())
        ptr2 = _ptr_tensor(n2s, dev)
        elem2 = _elem_of_row(ptr2, B, n2, dev)
        cj_ptr = _ptr_tensor(n_cj, dev)
        cj_idx, cj_noise, n_add_d = up(draws["cj_idx"]), up(draws["cj_noise"]), up(draws["n_add"])
        pos2 = f32(max(n2, 1), 3)
        _lib.call("agb_plot_extend", _P(pos1), _P(ptr1), _P(mins), _P(ptr2), _P(elem2), B, n2, _P(n_add_d), _P(cj_ptr),
                  _P(cj_idx), _P(cj_noise), _P(pos2), _lib.stream())
        # RandomPolygon2dExtend + features
        nv = int(draws["nv"])
        polys = up(draws["polys"])
        nn_ = max(n2, 1)
        ws = torch.empty(_lib.size_call("agb_plot_workspace_bytes", n2, B), dtype=torch.uint8, device=dev)
        pos_o, x_o = f32(nn_, 3), f32(nn_, 3)
        src = torch.empty(nn_, dtype=torch.int64, device=dev)
        out_ptr, n_out = i32(B + 1), i32(1)
        _lib.call("agb_plot_crop_ws", _P(pos2), _P(ptr2), _P(elem2), B, n2, _P(polys), nv, float(c.center[0]),
                  float(c.center[1]), _P(ws), _P(pos_o), _P(x_o), _P(src), _P(out_ptr), _P(n_out), _lib.stream())
        return pos_o, x_o, src, out_ptr

    def __call__(self, plots: List, device, y_reg=None, draws: Optional[List[dict]] = None, perms=None):
        if draws is None:
            draws = [draw_sample(torch.as_tensor(p, dtype=torch.float32).reshape(-1, 3), self.cfg) for p in plots]
        pos, x, src, out_ptr = self.augment(plots, draws, device)
        pos, x, src, lens = self.tail.fix_counts(pos, x, src, out_ptr, with_extent=True)
        return self.tail.finish(pos, x, src, lens, len(plots), y_reg=y_reg, perms=perms, extent_hint=self.tail.grid_extent)

    def staged(self, plots: List, device, y_reg=None, draws: Optional[List[dict]] = None, perms=None):
        """``__call__`` as a generator that never waits for the device: at each of the chain's two count read-backs (rows after
        the crop; voxels after GridSampling3D) it starts an asynchronous copy into pinned memory and YIELDS the pending read
        (voxelize.AsyncRead); resumed later (``gen.send(read.value())``) — a step later in a pipelined loop, when the copy has
        long landed — it continues with the values.  The PlotBatch is the generator's return value (StopIteration.value).
        Same kernels, same draws, same batch as ``__call__``."""
        from .voxelize import AsyncRead
        if draws is None:
            draws = [draw_sample(torch.as_tensor(p, dtype=torch.float32).reshape(-1, 3), self.cfg) for p in plots]
        B = len(plots)
        pos, x, src, out_ptr = self.augment(plots, draws, device)
        host = yield AsyncRead(self.tail.fix_counts_begin(pos, out_ptr, with_extent=True))
        pos, x, src, lens = self.tail.fix_counts_end(pos, x, src, host, B)
        return (yield from self.tail.finish_staged(pos, x, src, lens, B, y_reg=y_reg, perms=perms,
                                                   extent_hint=self.tail.grid_extent, reader=AsyncRead))


class PointTrainPipeline(SparseTrainPipeline):
    """xy.yaml train_transform (conf/data/instance/NFI/transforms/xy.yaml:4-75: the chain of the KPConv and PointNet models)
    for a batch of raw plots -> a point batch on the device (pos, x = [ones, pos_z, xy_distance], batch; ``ptr`` on the host).
    The same draws and device kernels as the sparse chain up to the features; MaxPoints 6144, no voxel tail."""

    def __init__(self, cfg: Optional[NFITrainConfig] = None):
        import dataclasses
        cfg = dataclasses.replace(cfg or NFITrainConfig(), voxel=None) if cfg is not None else NFITrainConfig(voxel=None, max_points=6144)
        super().__init__(cfg, device_shuffle=False)


class StagedBatches:
    """A software pipeline over ``SparseTrainPipeline.staged``: every ``advance()`` moves each batch in flight one stage
    further on the given stream (oldest first) and returns the batches that completed — the host thread never waits for a
    count read-back, it picks each one up a step after it was started.  A batch needs three advances from ``submit``."""

    def __init__(self, pipeline, device, stream=None):
        self.pipe, self.device, self.stream = pipeline, device, stream
        self.flight = []        # [generator, pending AsyncRead]

    def submit(self, plots, y_reg=None, draws=None, perms=None):
        with torch.cuda.stream(self.stream) if self.stream is not None else _null():
            gen = self.pipe.staged(plots, self.device, y_reg=y_reg, draws=draws, perms=perms)
            self.flight.append([gen, next(gen)])

    def advance(self):
        done, keep = [], []
        with torch.cuda.stream(self.stream) if self.stream is not None else _null():
            for item in self.flight:
                gen, pending = item
                try:
                    item[1] = gen.send(pending.value())
                    keep.append(item)
                except StopIteration as fin:
                    done.append(fin.value)
        self.flight = keep
        return done

    def __len__(self):
        return len(self.flight)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

