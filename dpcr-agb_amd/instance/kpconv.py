"""``KPConv`` model wrapper of the reference (torch_points3d/models/instance/kpconv.py:38-276) with the input
pyramid built ON THE GPU: ``prepare_inputs`` (:145-264) runs 5 radius searches, 4 grid subsamplings and 4 pooled
radius searches per batch — single-threaded C++ in the reference's training main loop (4.0 s per 32 x 6144-point
batch measured in SURVEY.md), HIP kernels here (csrc/kpindex.hip), with the index matrices staying on the device
as int32 instead of travelling host -> device as int64.

Level radii / grid sizes follow the reference: r_0 = first_subsampling_dl * conv_radius, pool grid dl = 2 r / conv_radius,
r doubles per level (:148,197,234).  Head: one Linear(1024, 1) per regression target (:17-35).
The reference's ``has_mol_targets`` / ``has_cls_targets`` reads (:126,129) are defects (never defined): treated
as False.
"""
from typing import List

import numpy as np
import torch
from torch import nn

from .. import _lib, kp_index
from ..backbones.kpconv import KPCNN
from ..config import Opt
from .base import InstanceBase


class SeparateLinear(nn.Module):
    def __init__(self, in_channel, num_reg_classes):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(in_channel, 1, bias=True) for _ in range(max(num_reg_classes, 0))])

    def forward(self, x):
        return torch.cat([lin(x) for lin in self.linears], 1)


class KPConvModel(InstanceBase):
    def __init__(self, option, model_type, dataset, modules=None):
        super().__init__(option, model_type, dataset, modules)
        self.config = config = option.config
        self.model = KPCNN(config)
        self.neighborhood_limits: List[int] = []
        in_channel = self.model.head_mlp.mlp.weight.shape[0]
        self.head = SeparateLinear(in_channel, self.num_reg_classes)
        self.head_optim_settings = option.get("head_optim_settings", {})
        self.backbone_optim_settings = option.get("backbone_optim_settings", {})
        self.random_grid_orient = option.get("random_grid_orient", True)

    def get_parameter_list(self) -> List[dict]:
        return [{"params": list(self.head.parameters()), **self.head_optim_settings},
                {"params": list(self.model.parameters()), **self.backbone_optim_settings}]

    # ------------------------------------------------------------------ input pyramid (kpconv.py:145-264)
    def _crop(self, mat, layer):
        if len(self.neighborhood_limits) > 0:
            if hasattr(mat, "cropped"):          # ragged rows: the limit travels with them, nothing is copied
                return mat.cropped(self.neighborhood_limits[layer])
            return mat[:, :self.neighborhood_limits[layer]].contiguous()
        return mat

    # Neighbour lists stay RAGGED inside the pyramid (kp_index.Neighbors; `.padded()` gives the reference's matrix): set to
    # False for the padded int32 matrices of earlier rounds (A/B measurements)
    ragged_neighbors = True

    def prepare_inputs(self, stacked_points, stacked_features, stack_lengths, device, rotations=None, bounds=None):
        """stacked_points [N,3] / stacked_features [N,F] (numpy or tensors), stack_lengths int[B].
        rotations: optional list (one float32 [B,3,3] per strided level) replacing the np.random grid orientations.
        bounds: optional (min xyz, max xyz) of all points (saves the one bounding-box read-back)."""
        gen = self.prepare_inputs_staged(stacked_points, stacked_features, stack_lengths, device, rotations, bounds)
        try:
            want = next(gen)
            while True:
                want = gen.send(kp_index.read_back(*want) if want else [])     # one host synchronisation per level
        except StopIteration as done:
            return done.value

    def prepare_inputs_staged(self, stacked_points, stacked_features, stack_lengths, device, rotations=None, bounds=None):
        """``prepare_inputs`` as a generator: at each of its count read-backs (one per level) it YIELDS the list of small device
        tensors it needs on the host and is resumed with their values (``gen.send(values)``, the format of
        ``kp_index.read_back``); the pyramid is the generator's return value.  A pipelined caller (``prefetch_input``) starts
        the copy, goes on enqueuing the training step, and comes back when it has landed: the host never waits."""
        cfg = self.config
        pts = torch.as_tensor(stacked_points, dtype=torch.float32).to(device).contiguous()
        feats = torch.as_tensor(stacked_features, dtype=torch.float32).to(device).contiguous()
        lens = np.asarray(stack_lengths, dtype=np.int64).reshape(-1)
        r_normal = cfg.first_subsampling_dl * cfg.conv_radius
        layer_blocks, points, neighbors, pools, lengths = [], [], [], [], []
        empty_i = torch.zeros(0, 1, dtype=torch.int32, device=device)
        # Bounding box of the whole batch: ONE host read for the pyramid (barycentres never leave their parent's box, and
        # a cloud rotated about the origin keeps an extent below the box's diagonal), or none when the caller knows it.
        # The grid-subsampling cells of a randomly oriented cloud are sized from the largest SINGLE-cloud diagonal (a seventh
        # value of `bounds`; plots at different world positions in one batch would otherwise inflate every cloud's cell
        # budget to the batch's extent); a caller that hands in six values vouches that the batch diagonal will do.
        if bounds is None:
            bounds = kp_index.support_bounds(pts, lens, cloud_diag=self.random_grid_orient)      # (one waiting read)
        if len(bounds) > 6:
            diag, bounds = float(bounds[6]) * 1.0001 + 1e-6, tuple(bounds[:6])
        else:
            diag = float(np.linalg.norm(np.asarray(bounds[3:]) - np.asarray(bounds[:3]))) * 1.0001 + 1e-6
        level = 0
        pool_job = None          # pooled radius search of the previous level: its width is read with this level's counts
        for block in cfg.architecture:
            if not ("pool" in block or "strided" in block or "global" in block or "upsample" in block):
                layer_blocks.append(block)
                continue
            strided = "pool" in block or "strided" in block
            # enqueue everything of this level that needs no host value, then read all counts back at once: one
            # synchronisation per level instead of seven (each one waits for the side stream behind a running step)
            conv_job = kp_index.neighbors_begin(pts, pts, lens, lens, r_normal, bounds) if layer_blocks else None
            sub_job = rot = None
            if strided:
                dl = 2 * r_normal / cfg.conv_radius
                src = pts
                if self.random_grid_orient:
                    rot = kp_index.random_grid_rotations(len(lens)) if rotations is None else \
                        np.asarray(rotations[level], dtype=np.float32)
                    src = kp_index.rotate_points(pts, lens, rot, False)
                ext = (diag,) * 3 if self.random_grid_orient else tuple(np.asarray(bounds[3:]) - np.asarray(bounds[:3]))
                sub_job = kp_index.subsample_begin(src, None, lens, dl, ext)
            rag = self.ragged_neighbors
            sizes = lambda j: [j.max_count, j.row_ptr[-1:]] if rag else [j.max_count]      # noqa: E731
            finish = (lambda j, g: kp_index.neighbors_finish_csr(j, g.pop(0)[0], g.pop(0)[0])) if rag else \
                (lambda j, g: kp_index.neighbors_finish(j, g.pop(0)[0]))
            want = (sizes(conv_job) if conv_job else []) + (sizes(pool_job) if pool_job else []) + \
                ([sub_job.out_ptr, sub_job.status[:1]] if sub_job else [])
            got = (yield want) if want else []
            conv_i = finish(conv_job, got) if conv_job else empty_i
            if pool_job is not None:
                pools[-1] = self._crop(finish(pool_job, got), len(points) - 1)
                pool_job = None
            if strided:
                optr, st = got
                pool_p, _, pool_b, _ = kp_index.subsample_finish(sub_job, optr, st[0])
                pool_b = pool_b.astype(np.int64)
                if rot is not None:
                    pool_p = kp_index.rotate_points(pool_p, pool_b, rot, True)
                pool_job = kp_index.neighbors_begin(pool_p, pts, pool_b, lens, r_normal, bounds)
                pool_i = None        # filled in when the next level's counts are read
            else:
                pool_i, pool_p, pool_b = empty_i, torch.zeros(0, 3, device=device), np.zeros(0, dtype=np.int64)
            points.append(pts)
            if hasattr(conv_i, "cropped"):
                conv_i.agb_symmetric = True      # (kept by cropped() only while the limit cuts nothing)
            nb = self._crop(conv_i, len(points) - 1)
            # an uncropped radius search of a point set against itself is symmetric (d2(a, b) is computed from the same
            # differences either way): KPConv layers on it take the scatter-free backward (KPConvSymmetricFunction)
            if not hasattr(nb, "cropped"):
                nb.agb_symmetric = conv_job is not None and nb.shape[1] == conv_i.shape[1]
            neighbors.append(nb)
            pools.append(pool_i if pool_i is None else self._crop(pool_i, len(points) - 1))
            lengths.append(torch.from_numpy(lens.copy()))
            pts, lens = pool_p, pool_b
            r_normal *= 2
            layer_blocks = []
            level += 1
            if "global" in block or "upsample" in block:
                break
        if pool_job is not None:     # (an architecture that ends on a strided block)
            if self.ragged_neighbors:
                got = yield [pool_job.max_count, pool_job.row_ptr[-1:]]
                pools[-1] = self._crop(kp_index.neighbors_finish_csr(pool_job, got[0][0], got[1][0]), len(points) - 1)
            else:
                got = yield [pool_job.max_count]
                pools[-1] = self._crop(kp_index.neighbors_finish(pool_job, got[0][0]), len(points) - 1)
        ptr = np.zeros(len(lengths[-1]) + 1, dtype=np.int32)
        np.cumsum(lengths[-1].numpy(), out=ptr[1:])
        return dict(points=points, neighbors=neighbors, pools=pools, lengths=lengths, features=feats,
                    last_ptr=kp_index.h2d_small(ptr, device))

    def _pyramid(self, data, device):
        ptr = data.ptr
        lens = (ptr[1:] - ptr[:-1]).cpu().numpy().astype(np.int64)
        return self.prepare_inputs(data.pos.view(-1, 3), data.x.view(-1, data.x.shape[-1]), lens, device,
                                   bounds=getattr(data, "pos_bounds", None))

    def prefetch_input(self, data, device):
        """Build the NEXT batch's input pyramid on a side stream WITHOUT waiting for it: ``prepare_inputs`` reads counts back
        to the host after every level; here each of those reads is an asynchronous copy that the training loop picks up
        later (``poll_prefetch``: between the forward pass, the backward pass and the optimiser step of the running step,
        and from ``set_input``) — the host thread keeps enqueuing the step instead of sleeping on the side stream (round 5:
        7.6 ms of a 17.9 ms step spent in here, most of it waiting).  Grid orientations are still drawn from ``np.random`` in
        batch order while one batch is in flight."""
        if not hasattr(self, "_side_stream"):
            self._side_stream = torch.cuda.Stream(device=device)
        job = _PyramidJob(self, data, device)
        data._prefetched = job
        self.__dict__.setdefault("_jobs", []).append(job)
        job.poll()
        # while a pyramid is in flight the library gives it a turn every few calls (forward and backward pass alike)
        _lib.POLL_HOOK = self.poll_prefetch

    def poll_prefetch(self):
        """Move every pyramid in flight on as far as its read-backs have landed (never waits)."""
        if self.__dict__.get("_polling"):
            return                      # (the pyramid's own library calls come through the hook again)
        jobs = self.__dict__.get("_jobs")
        if jobs:
            self.__dict__["_polling"] = True
            try:
                for job in jobs:
                    job.poll()
            finally:
                self.__dict__["_polling"] = False
            jobs = self.__dict__["_jobs"] = [j for j in jobs if not j.done]
        if not jobs and _lib.POLL_HOOK == self.poll_prefetch:
            _lib.POLL_HOOK = None

    def set_input(self, data, device):
        # (plain attributes through __dict__: nn.Module.__setattr__ costs ~30 us per assignment, twelve of them per step)
        d = self.__dict__
        d["data_visual"] = data
        d["batch_idx"] = data.batch
        pre = getattr(data, "_prefetched", None)
        if pre is not None:
            data._prefetched = None
            self.__dict__["_polling"] = True
            try:
                inp, ev = pre.finish()          # (waits only for what has not landed yet)
            finally:
                self.__dict__["_polling"] = False
            jobs = self.__dict__.get("_jobs")
            if jobs and pre in jobs:
                jobs.remove(pre)
            if not jobs and _lib.POLL_HOOK == self.poll_prefetch:
                _lib.POLL_HOOK = None
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            used = []                   # built on the side stream, consumed on the compute stream
            for v in inp.values():
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    for u in (t.tensors() if hasattr(t, "tensors") else [t]):
                        if isinstance(u, torch.Tensor) and u.is_cuda:
                            used.append(u)
            self._hold_input(inp, used, cur)
            d["input"] = Opt(inp)
        else:
            d["input"] = Opt(self._pyramid(data, device))
        if len(self.loss_fns) > 0:
            bs = len(data)
            if self.has_reg_targets and data.y_reg is not None:
                mask_all = getattr(data, "y_reg_mask_all", None)
                d["_reg_mask_all"] = bool(data.y_reg_mask.all()) if mask_all is None else mask_all
                d["reg_y_mask"] = data.y_reg_mask.to(device, non_blocking=True).view(bs, -1)
                d["reg_y"] = data.y_reg.to(device, non_blocking=True).view(bs, -1)
            else:
                d["reg_y"] = None      # (a label-less batch: prediction only, base.compute_reg_loss skips the loss)

    def forward(self, *args, **kwargs):
        out = self.model(self.input)
        d = self.__dict__
        d["output"] = self.head(out)
        d["reg_out"] = self.convert_outputs(d["output"])
        self.compute_loss()


class _PyramidJob:
    """One batch's input pyramid in flight on the model's side stream (``KPConvModel.prepare_inputs_staged`` driven by
    asynchronous read-backs)."""

    def __init__(self, model, data, device):
        self.model, self.done, self.result, self.event = model, False, None, None
        self.side = model._side_stream
        ptr = data.ptr
        lens = (ptr[1:] - ptr[:-1]).cpu().numpy().astype(np.int64)
        with torch.cuda.stream(self.side):
            self.gen = model.prepare_inputs_staged(data.pos.view(-1, 3), data.x.view(-1, data.x.shape[-1]), lens, device,
                                                   bounds=getattr(data, "pos_bounds", None))
            self.pending = None
            self._step(None)

    def _step(self, values):
        """Resume the generator (inside the side-stream context) and start the copy of what it asks for next."""
        try:
            want = next(self.gen) if values is None else self.gen.send(values)
            self.pending = kp_index.PendingRead(want)
        except StopIteration as fin:
            self.result, self.done, self.pending = fin.value, True, None
            self.event = self.side.record_event()

    def poll(self):
        if self.done:
            return
        with torch.cuda.stream(self.side):
            while not self.done and self.pending.ready():
                self._step(self.pending.values())

    def finish(self):
        with torch.cuda.stream(self.side):
            while not self.done:
                self._step(self.pending.values())
        return self.result, self.event


KPConv = KPConvModel  # the reference's class name (models/instance/kpconv.py:38)
