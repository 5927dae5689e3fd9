"""Device-side coordinate manager for the sparse-voxel path.

Plays the role of MinkowskiEngine's CoordinateManager behind ``ME.SparseTensor(features, coordinates)``
(reference call site: torch_points3d/models/instance/minkowski.py:67-80) and of the coordinate/kernel maps
that ``ME.MinkowskiConvolution`` / ``ME.MinkowskiMaxPooling`` request (SENet.py:47-53, resnet_block.py:48-55).

MI355X-first layout: every level keeps
  * coords   int32 [N, 4]  (b, x, y, z), rows batch-contiguous, order = first occurrence in the parent level
  * a 64-bit-key open-addressing hash (capacity = pow2 >= 2N) in HBM
  * kernel maps as dense neighbour tables nbr[K^3][N_out] (int32, -1 = absent) so that the convolution
    can run output-stationary (no scatter atomics); tables are cached per (level, kernel, stride, dilation)
    exactly like ME caches kernel maps per coordinate-map key pair.
"""
from typing import Dict, Tuple

import torch

from . import _lib


class CoordinateMapKey:
    """Identifies one coordinate level: the isotropic tensor stride (0 = the pooled 'one row per batch' level)."""

    def __init__(self, tensor_stride: int):
        self.tensor_stride = int(tensor_stride)

    def get_tensor_stride(self):
        return [self.tensor_stride] * 3

    def get_key(self):
        return ([self.tensor_stride] * 3, "")

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and other.tensor_stride == self.tensor_stride

    def __hash__(self):
        return hash(self.tensor_stride)

    def __repr__(self):
        return f"CoordinateMapKey(tensor_stride={self.tensor_stride})"


class _Level:
    __slots__ = ("coords", "n", "ts", "keys", "vals", "cap", "_ptr", "slot")

    def __init__(self, coords, n, ts, keys, vals, cap):
        self.coords, self.n, self.ts = coords, n, ts
        self.keys, self.vals, self.cap = keys, vals, cap
        self._ptr = None
        self.slot = None


def _as_int(v):
    if isinstance(v, (list, tuple)):
        assert all(int(x) == int(v[0]) for x in v), "only isotropic kernel/stride/dilation are supported"
        return int(v[0])
    return int(v)


class CoordinateManager:
    def __init__(self, coordinates: torch.Tensor, device=None, tensor_stride: int = 1, batch_size: int = None):
        if coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise ValueError("coordinates must be [N, 1+3] = (batch, x, y, z)")
        if device is None:
            device = coordinates.device
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.AgbError("the sparse-voxel path runs on a HIP device only (no CPU fallback); got device "
                                f"'{device}'")
        coords = coordinates.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()
        self.device = device
        self.D = 3
        n = coords.shape[0]
        if batch_size is None:
            batch_size = int(coordinates[:, 0].max().item()) + 1 if n > 0 else 1
        self.batch_size = int(batch_size)
        self.levels: Dict[int, _Level] = {}
        self.kernel_maps: Dict[Tuple, torch.Tensor] = {}
        self.origin_ts = int(tensor_stride)

        cap = _lib.hash_capacity(n)
        keys = torch.empty(cap, dtype=torch.int64, device=device)
        vals = torch.empty(cap, dtype=torch.int32, device=device)
        slot = torch.empty(max(n, 1), dtype=torch.int32, device=device)
        status = torch.empty(4, dtype=torch.int32, device=device)
        _lib.call("agb_coords_insert", _lib.ptr(coords), n, None, _lib.ptr(keys), _lib.ptr(vals), cap,
                  _lib.ptr(slot), _lib.ptr(status), _lib.stream())
        lvl = _Level(coords, n, self.origin_ts, keys, vals, cap)
        self.levels[self.origin_ts] = lvl
        self._status = status  # checked lazily (one host read) in validate()
        self._validated = False

    # ------------------------------------------------------------------ helpers
    def validate(self):
        """Host-side check of the insert status (duplicates / range / batch order). One device read."""
        if self._validated:
            return
        dup, rng, order, _ = self._status.tolist()
        if rng:
            raise _lib.AgbError(f"{rng} coordinates fall outside the packed 16-bit range [-32768, 32767]")
        if dup:
            raise _lib.AgbError(f"{dup} duplicate coordinates: voxelise first (GridSampling3D keeps one point per "
                                "voxel), duplicates are not merged here")
        if order:
            raise _lib.AgbError("rows must be ordered by batch index (as torch_geometric's Batch collation gives)")
        self._validated = True

    def level(self, ts: int) -> _Level:
        return self.levels[int(ts)]

    def num_rows(self, key: CoordinateMapKey) -> int:
        if key.tensor_stride == 0:
            return self.batch_size
        return self.levels[key.tensor_stride].n

    def coords_of(self, key: CoordinateMapKey) -> torch.Tensor:
        if key.tensor_stride == 0:
            c = torch.zeros(self.batch_size, 4, dtype=torch.int32, device=self.device)
            c[:, 0] = torch.arange(self.batch_size, device=self.device, dtype=torch.int32)
            return c
        lvl = self.levels[key.tensor_stride]
        return lvl.coords[: lvl.n]

    def batch_ptr(self, ts: int) -> torch.Tensor:
        lvl = self.levels[int(ts)]
        if lvl._ptr is None:
            p = torch.empty(self.batch_size + 1, dtype=torch.int32, device=self.device)
            _lib.call("agb_batch_ptr", _lib.ptr(lvl.coords), lvl.n, None, self.batch_size, _lib.ptr(p),
                      _lib.stream())
            lvl._ptr = p
        return lvl._ptr

    # ------------------------------------------------------------------ levels
    def stride(self, ts_in: int, stride: int) -> int:
        """Create (or fetch) the level with tensor stride ts_in*stride: unique(floor(c/ts_out)*ts_out)."""
        stride = _as_int(stride)
        ts_out = int(ts_in) * stride
        if stride == 1 or ts_out in self.levels:
            return ts_out
        src = self.levels[int(ts_in)]
        n = src.n
        dev = self.device
        cap = _lib.hash_capacity(n)
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        vals = torch.empty(cap, dtype=torch.int32, device=dev)
        slot = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        flags = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        excl = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        scratch = torch.empty(_lib.scan_scratch_elems(n), dtype=torch.int32, device=dev)
        out_coords = torch.empty(max(n, 1), 4, dtype=torch.int32, device=dev)
        n_out_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("agb_coords_stride", _lib.ptr(src.coords), n, None, ts_out, _lib.ptr(keys), _lib.ptr(vals), cap,
                  _lib.ptr(slot), _lib.ptr(flags), _lib.ptr(excl), _lib.ptr(scratch), _lib.ptr(out_coords),
                  _lib.ptr(n_out_dev), None, _lib.stream())
        n_out = int(n_out_dev.item())  # the one host read per new level (sizes the level's tensors)
        self.levels[ts_out] = _Level(out_coords, n_out, ts_out, keys, vals, cap)
        return ts_out

    # -------------------------------------------------------------- kernel maps
    def kernel_map(self, ts_in: int, kernel_size: int, stride: int = 1, dilation: int = 1) -> torch.Tensor:
        """nbr[K^3, N_out]: row (at level ts_in) of out_coord + offset_k * ts_in * dilation, or -1."""
        K, s, d = _as_int(kernel_size), _as_int(stride), _as_int(dilation)
        key = ("fwd", int(ts_in), K, s, d)
        if key in self.kernel_maps:
            return self.kernel_maps[key]
        ts_out = self.stride(ts_in, s)
        src, dst = self.levels[int(ts_in)], self.levels[ts_out]
        nbr = torch.empty(K ** 3, max(dst.n, 1), dtype=torch.int32, device=self.device)
        pairs = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.call("agb_kernel_map", _lib.ptr(dst.coords), dst.n, None, K, int(ts_in) * d, 1, 0, _lib.ptr(src.keys),
                  _lib.ptr(src.vals), src.cap, _lib.ptr(nbr), nbr.stride(0), _lib.ptr(pairs), _lib.stream())
        nbr.agb_pairs = pairs  # device-side kernel-map size (ME's kernel-map size); read only by profiling code
        self.kernel_maps[key] = nbr
        return nbr

    def transposed_map(self, ts_in: int, kernel_size: int, stride: int = 1, dilation: int = 1) -> torch.Tensor:
        """nbrT[K^3, N_in]: row (at level ts_out) of in_coord - offset_k * ts_in * dilation, or -1."""
        K, s, d = _as_int(kernel_size), _as_int(stride), _as_int(dilation)
        key = ("bwd", int(ts_in), K, s, d)
        if key in self.kernel_maps:
            return self.kernel_maps[key]
        ts_out = self.stride(ts_in, s)
        src, dst = self.levels[int(ts_in)], self.levels[ts_out]
        nbrT = torch.empty(K ** 3, max(src.n, 1), dtype=torch.int32, device=self.device)
        _lib.call("agb_kernel_map", _lib.ptr(src.coords), src.n, None, K, int(ts_in) * d, -1, ts_out,
                  _lib.ptr(dst.keys), _lib.ptr(dst.vals), dst.cap, _lib.ptr(nbrT), nbrT.stride(0), None,
                  _lib.stream())
        self.kernel_maps[key] = nbrT
        return nbrT
