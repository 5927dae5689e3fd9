"""GridSampling3D(size, quantize_coords=True, mode="last") on the GPU for a whole batch of clouds
(reference: torch_points3d/core/data_transform/grid_transform.py:83-140, applied per sample in DataLoader workers).

``voxelize_last(pos, lengths, size, perm=None)`` returns per-cloud voxel coordinates (int32, ascending voxel key:
z-major / x fastest), the index of the representative point of every voxel in the ORIGINAL stacked order (every
per-point attribute is then ``attr[keep]``), the new cloud lengths, and the integer bounding box that the sparse
coordinate manager wants.  ``perm`` is the within-cloud shuffle the reference draws with ``torch.randperm``
(grid_transform.py:24); pass it to be comparable, omit it to draw one per cloud the same way.
Only mode="last" is implemented (the NFI sparse pipelines use it: sparse-xy.yaml:95-99).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .kp_index import _elem_of_row, _lengths, _ptr_tensor

_P = _lib.ptr
_V, _I, _F = _lib.c_void_p, _lib.c_int, _lib.c_float
_lib.declare("agb_voxelize_last_workspace_bytes", [_I, _I, _I])
_lib.declare("agb_voxelize_last_ws", [_V, _V, _V, _V, _I, _I, _F, _I, _V, _V, _V, _V, _V, _V, _V, _V])
_lib.declare("agb_voxelize_last_seeded_ws", [_V, ctypes.c_ulonglong, _V, _V, _I, _I, _F, _I, _V, _V, _V, _V, _V, _V, _V, _V])


def draw_permutations(lengths):
    """One torch.randperm per cloud, in batch order (what shuffle_data does when the transform runs per sample)."""
    return torch.cat([torch.randperm(int(n)) for n in lengths]) if len(lengths) else torch.zeros(0, dtype=torch.int64)


def device_permutations(lengths, device, generator=None):
    """The same shuffle drawn ON THE DEVICE, for input pipelines that must not spend host time per point (32 plots of 13 000
    rows: 4.9 ms of ``torch.randperm`` on one core, then a 4 MB upload): every row gets 40 random bits from torch's device
    generator and the rows of a cloud are ordered by them (one int64 sort of the batch) — a uniformly random permutation per
    cloud up to ties of the 40-bit keys (probability ~1e-4 per cloud, resolved by the sort).  Not the reference's draw
    sequence (``torch.randperm`` on the host, ``draw_permutations``); ``GridSampling3D(mode="last")`` only needs SOME
    uniformly random order to pick a voxel's representative."""
    lens = torch.as_tensor(np.asarray(lengths, dtype=np.int64))
    n = int(lens.sum())
    dev = torch.device(device)
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    from .kp_index import h2d_small      # (pinned ring: a pageable upload makes the runtime wait for the whole device)
    lens_d = h2d_small(lens.numpy(), dev)
    elem = torch.repeat_interleave(torch.arange(len(lens), device=dev), lens_d, output_size=n)
    key = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device=dev, generator=generator)
    order = torch.argsort((elem << 40) | key)
    ptr = torch.cumsum(lens_d, 0) - lens_d
    return order - ptr[elem]


class AsyncRead:
    """A small device tensor on its way to the host WITHOUT a wait: asynchronous copy into a recycled pinned buffer on the
    current stream + an event.  ``value()`` (a list) waits for the event — by then, in a pipelined caller, long past."""
    _POOL = {}

    def __init__(self, t):
        key = (t.numel(), t.dtype)
        pool = AsyncRead._POOL.setdefault(key, [])
        self.host = pool.pop() if pool else torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self.host.copy_(t, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self._keep = t

    def value(self):
        self.event.synchronize()
        out = self.host.tolist()
        AsyncRead._POOL.setdefault((self.host.numel(), self.host.dtype), []).append(self.host)
        self.host = self._keep = None
        return out


def voxelize_last_begin(pos, lengths, size, perm=None, extent_hint=None, seed=None, cloud_boxes=False):
    """First half of voxelize_last: everything up to (not including) the host read of the new lengths / bounds.  Returns
    (state, to_read): `to_read` is the device tensor whose values ``voxelize_last_end`` needs (None: nothing to voxelize).
    cloud_boxes: the box of every cloud's voxel coordinates rides in the same read (``voxelize_last_end(...,
    cloud_boxes=True)`` returns them): what a per-cloud coordinate augmentation needs to state its exact box."""
    lens = _lengths(lengths)
    B, n = len(lens), int(pos.shape[0])
    if int(lens.sum()) != n:
        raise ValueError("lengths do not sum to the number of points")
    if not torch.cuda.is_available():
        raise _lib.AgbError("voxelize_last needs a HIP device (no CPU fallback in the product path)")
    dev = pos.device if pos.is_cuda else torch.device("cuda", torch.cuda.current_device())
    if n == 0:      # nothing to voxelize: every cloud keeps zero voxels
        return (B, dev, None, None), None
    p = pos.to(device=dev, dtype=torch.float32).contiguous()
    if seed is None:
        if perm is None:
            perm = draw_permutations(lens)
        perm = perm.to(device=dev, dtype=torch.int64).contiguous()
    ptr = _ptr_tensor(lens, dev)
    elem = _elem_of_row(ptr, B, n, dev)
    size32 = float(np.float32(size))
    if extent_hint is None:
        from .kp_index import elem_bbox
        bb = elem_bbox(p, ptr, B)
        ext = ((bb[:, 3:] - bb[:, :3]).max(0).values / size32).tolist()   # one host read
    else:
        ext = list(extent_hint)
    cap = 1
    for e in ext:
        cap *= int(np.floor(e)) + 3
    if B * cap >= (1 << 30):
        raise _lib.AgbError(f"voxel grid of {B * cap} cells is too large: voxel size too small for these clouds")
    i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)  # noqa: E731
    ws = torch.empty(_lib.size_call("agb_voxelize_last_workspace_bytes", n, B, cap), dtype=torch.uint8, device=dev)
    coords = torch.empty(max(n, 1), 3, dtype=torch.int32, device=dev)
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    out_ptr, n_out, bounds, status = i32(B + 1), i32(1), i32(6), i32(4)
    if seed is not None:      # the shuffle drawn on the device from one 64-bit seed: no permutation tensor, no sort
        _lib.call("agb_voxelize_last_seeded_ws", _P(p), ctypes.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF), _P(ptr), _P(elem), B,
                  n, size32, cap, _P(ws), _P(coords), _P(keep), _P(out_ptr), _P(n_out), _P(bounds), _P(status), _lib.stream())
    else:
        _lib.call("agb_voxelize_last_ws", _P(p), _P(perm), _P(ptr), _P(elem), B, n, size32, cap, _P(ws), _P(coords), _P(keep),
                  _P(out_ptr), _P(n_out), _P(bounds), _P(status), _lib.stream())
    parts = [out_ptr, bounds, status[:1]]                  # new lengths + coordinate bounds + status
    if cloud_boxes:
        from .kp_index import elem_bbox
        parts.append(elem_bbox(coords.float(), out_ptr, B).reshape(-1).to(torch.int32))     # (|coordinate| < 2^24: exact)
    return (B, dev, coords, keep), torch.cat(parts)


def voxelize_last_end(state, host, cloud_boxes=False):
    """Second half: `host` = the values of ``voxelize_last_begin``'s `to_read` as a list.  cloud_boxes: a fifth result, int64
    [B, 6] = (min xyz, max xyz) of every cloud's voxel coordinates (rows of empty clouds are meaningless)."""
    B, dev, coords, keep = state
    if coords is None:
        out = (torch.empty(0, 3, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int64, device=dev),
               np.zeros(B, dtype=np.int64), (0,) * 6)
        return out + (np.zeros((B, 6), np.int64),) if cloud_boxes else out
    if host[B + 7]:
        raise _lib.AgbError("voxelize_last: a cloud exceeds the reserved cell capacity (extent_hint too small)")
    optr = np.asarray(host[:B + 1], dtype=np.int64)
    m = int(optr[-1])
    out = (coords[:m], keep[:m], np.diff(optr).astype(np.int64), tuple(int(v) for v in host[B + 1:B + 7]))
    if cloud_boxes:
        out = out + (np.asarray(host[B + 8:B + 8 + 6 * B], dtype=np.int64).reshape(B, 6),)
    return out


def voxelize_last(pos, lengths, size, perm=None, extent_hint=None, seed=None):
    """pos: float [N,3] stacked clouds (tensor, any device); lengths: int [B].
    extent_hint: optional upper bound of (max - min) of pos/size per axis (saves the sizing read-back).
    seed (int): draw the shuffle on the device from it (agb_voxelize_last_seeded_ws) instead of taking / drawing `perm`."""
    state, to_read = voxelize_last_begin(pos, lengths, size, perm=perm, extent_hint=extent_hint, seed=seed)
    return voxelize_last_end(state, None if to_read is None else to_read.tolist())   # one host read


class GridSampling3D:
    """Batch-level drop-in for the reference transform: takes an object with ``pos`` [N,3], ``batch`` [N] (sorted)
    and any per-point tensors, returns a PlotBatch-like object with ``coords``, per-point fields restricted to one
    point per voxel, and ``coord_bounds``."""

    def __init__(self, size, quantize_coords=True, mode="last", verbose=False):
        if mode != "last":
            raise NotImplementedError("only mode='last' is implemented (the NFI sparse pipelines use it)")
        self._grid_size, self._quantize_coords, self._mode = size, quantize_coords, mode

    def __call__(self, data, perm=None):
        from .synthetic import PlotBatch
        batch = data.batch
        num = len(data)
        lens = torch.bincount(batch.cpu(), minlength=num).numpy()
        coords, keep, new_lens, bounds = voxelize_last(data.pos, lens, self._grid_size, perm=perm)
        dev = coords.device
        pick = lambda t: None if t is None else t.to(dev)[keep]  # noqa: E731
        out = PlotBatch(pick(batch), coords if self._quantize_coords else None, pick(data.x), pick(data.pos),
                        None, None, num, bounds)
        out.y_reg = None if data.y_reg is None else data.y_reg.to(dev)
        out.y_reg_mask = None if data.y_reg_mask is None else data.y_reg_mask.to(dev)
        out.y_reg_mask_all = getattr(data, "y_reg_mask_all", None)
        out.grid_size = torch.tensor([self._grid_size])
        return out

    def __repr__(self):
        return f"GridSampling3D(grid_size={self._grid_size}, quantize_coords={self._quantize_coords}, mode={self._mode})"
