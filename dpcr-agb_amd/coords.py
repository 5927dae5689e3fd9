"""Device-side coordinate manager for the sparse-voxel path.

Plays the role of MinkowskiEngine's CoordinateManager behind ``ME.SparseTensor(features, coordinates)``
(reference call site: torch_points3d/models/instance/minkowski.py:67-80) and of the coordinate/kernel maps
that ``ME.MinkowskiConvolution`` / ``ME.MinkowskiMaxPooling`` request (SENet.py:47-53, resnet_block.py:48-55).

MI355X-first layout: every level keeps
  * coords   int32 [N, 4]  (b, x, y, z), rows batch-contiguous, order = first occurrence in the parent level
  * a lookup structure in HBM: a dense grid int32[B][Z][Y][X] when the batch's bounding volume is small
    (LiDAR plots: ~81^3 cells per plot -> 68 MB at B=32; one load per probe, coalesced), otherwise a
    64-bit-key open-addressing hash (capacity = pow2 >= 2N)
  * kernel maps as dense neighbour tables nbr[K^3][N_out] (int32, -1 = absent) so that the convolution
    can run output-stationary (no scatter atomics); tables are cached per (level, kernel, stride, dilation)
    exactly like ME caches kernel maps per coordinate-map key pair.

``prefetch_strides`` builds a whole pyramid of levels with device-side row counts and reads all counts back
in ONE host synchronisation; without it every new level costs one read-back (ME-like lazy behaviour).
"""
import ctypes
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib

GRID_MAX_CELLS = 1 << 28  # 1 GiB of int32 per level at most; beyond that the hash mode is used
GRID_HALO = 3             # empty cells around every level's grid: stride-1 kernels up to 7^3 probe it without bounds checks


_PINNED_COUNTS = {}      # (numel, dtype) -> idle pinned read-back buffers (prefetch_strides / finish_prefetch)

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
    """One coordinate level. ``n`` is the host row count (None until read back), ``bound`` an upper bound that
    sizes buffers, ``n_dev`` the device-side count (None when n is exact from the start)."""
    __slots__ = ("coords", "n", "bound", "n_dev", "ts", "keys", "vals", "cap", "grid", "desc_host", "_ptr",
                 "status")

    def __init__(self, coords, n, bound, n_dev, ts):
        self.coords, self.n, self.bound, self.n_dev, self.ts = coords, n, bound, n_dev, ts
        self.keys = self.vals = self.grid = self.desc_host = self._ptr = self.status = None
        self.cap = 0


def _as_int(v):
    if isinstance(v, (list, tuple)):
        assert all(int(x) == int(v[0]) for x in v), "only isotropic kernel/stride/dilation are supported"
        return int(v[0])
    return int(v)


def _floor_to(v, ts):
    return (v // ts) * ts  # python floor division: correct for negatives


_CLASS_TABLES = {}


def _class_table(K, s, d, device):
    """cls_tab[class] = (count, offsets k...) : kernel offsets through which an input row of that parity class can
    be reached from the stride-s output lattice: (c/ts - o*d) = 0 (mod s) per axis."""
    key = (K, s, d, str(device))
    if key not in _CLASS_TABLES:
        half = K // 2 if K % 2 == 1 else 0
        K3 = K ** 3
        tab = torch.zeros(s ** 3, 1 + K3, dtype=torch.int32)
        for cls in range(s ** 3):
            p = (cls % s, (cls // s) % s, cls // (s * s))
            offs = []
            for k in range(K3):
                i = (k % K, (k // K) % K, k // (K * K))
                if all(((p[a] - (i[a] - half) * d) % s) == 0 for a in range(3)):
                    offs.append(k)
            tab[cls, 0] = len(offs)
            tab[cls, 1:1 + len(offs)] = torch.tensor(offs, dtype=torch.int32)
        _CLASS_TABLES[key] = tab.to(device)
    return _CLASS_TABLES[key]


class CoordinateManager:
    def __init__(self, coordinates: torch.Tensor, device=None, tensor_stride: int = 1, batch_size: int = None,
                 bounds: Optional[Sequence[int]] = None, mode: str = "auto"):
        """bounds: optional (min_x, min_y, min_z, max_x, max_y, max_z) known by the caller (e.g. computed by the
        voxeliser); saves the bounding-box read-back. mode: 'auto' | 'grid' | 'hash'."""
        if coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise ValueError("coordinates must be [N, 1+3] = (batch, x, y, z)")
        if device is None:
            device = coordinates.device
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.AgbError("the sparse-voxel path runs on a HIP device only (no CPU fallback); got device "
                                f"'{device}'")
        n = coordinates.shape[0]
        if mode not in ("auto", "grid", "hash"):
            raise ValueError(f"mode '{mode}'")
        if bounds is None and mode != "hash" and n > 0 and not coordinates.is_cuda:
            c = coordinates[:, 1:]
            bounds = tuple(int(v) for v in c.min(0).values.tolist()) + tuple(int(v) for v in c.max(0).values.tolist())
        if batch_size is None:
            batch_size = int(coordinates[:, 0].max().item()) + 1 if n > 0 else 1
        coords = coordinates.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()
        self.device = device
        self.D = 3
        self.batch_size = int(batch_size)
        self.levels: Dict[int, _Level] = {}
        self.kernel_maps: Dict[Tuple, torch.Tensor] = {}
        self.origin_ts = ts0 = int(tensor_stride)
        if bounds is None and mode != "hash" and n > 0:
            bbox = torch.empty(8, dtype=torch.int32, device=device)
            _lib.call("agb_coords_bbox", _lib.ptr(coords), n, None, _lib.ptr(bbox), _lib.stream())
            bounds = tuple(bbox[:6].tolist())  # one host read (callers that know the bounds pass them in)
        self.bounds = None if bounds is None else tuple(int(v) for v in bounds)
        self.mode = "hash"
        if mode != "hash" and n > 0 and self._grid_cells(ts0) <= GRID_MAX_CELLS:
            self.mode = "grid"
        elif mode == "grid":
            raise _lib.AgbError("dense-grid mode requested but the bounding volume is empty or too large")

        lvl = _Level(coords, n, n, None, ts0)
        lvl.status = torch.empty(4, dtype=torch.int32, device=device)
        if self.mode == "grid":
            self._alloc_grid(lvl)
            cell = torch.empty(max(n, 1), dtype=torch.int64, device=device)
            _lib.call("agb_grid_insert", _lib.ptr(coords), n, None, lvl.desc_host, _lib.ptr(lvl.grid),
                      _lib.ptr(cell), _lib.ptr(lvl.status), _lib.stream())
        else:
            lvl.cap = _lib.hash_capacity(n)
            lvl.keys = torch.empty(lvl.cap, dtype=torch.int64, device=device)
            lvl.vals = torch.empty(lvl.cap, dtype=torch.int32, device=device)
            slot = torch.empty(max(n, 1), dtype=torch.int32, device=device)
            _lib.call("agb_coords_insert", _lib.ptr(coords), n, None, _lib.ptr(lvl.keys), _lib.ptr(lvl.vals),
                      lvl.cap, _lib.ptr(slot), _lib.ptr(lvl.status), _lib.stream())
        self.levels[ts0] = lvl
        self._validated = False

    # ------------------------------------------------------------------ dense-grid geometry
    def _grid_geometry(self, ts):
        lo = [_floor_to(v, ts) - GRID_HALO * ts for v in self.bounds[:3]]
        hi = [_floor_to(v, ts) + GRID_HALO * ts for v in self.bounds[3:]]
        dims = [(h - l) // ts + 1 for l, h in zip(lo, hi)]
        return lo, dims

    def _grid_cells(self, ts):
        _, dims = self._grid_geometry(ts)
        return self.batch_size * dims[0] * dims[1] * dims[2]

    def _alloc_grid(self, lvl: _Level):
        lo, dims = self._grid_geometry(lvl.ts)
        lvl.desc_host = (ctypes.c_int32 * 8)(lo[0], lo[1], lo[2], dims[0], dims[1], dims[2], lvl.ts,
                                             self.batch_size | (GRID_HALO << 16))
        lvl.grid = torch.empty(self.batch_size * dims[0] * dims[1] * dims[2], dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ helpers
    def validate(self):
        """Host-side check of the insert status (duplicates / range / batch order). One device read."""
        if self._validated:
            return
        self._check_status(self.levels[self.origin_ts].status.tolist())

    def _check_status(self, status):
        dup, rng, order, _ = status
        if rng:
            raise _lib.AgbError(f"{rng} coordinates fall outside the supported range (packed 16-bit keys / the "
                                "declared bounds)")
        if dup:
            raise _lib.AgbError(f"{dup} duplicate coordinates: voxelise first (GridSampling3D keeps one point per "
                                "voxel), duplicates are not merged here")
        if order:
            raise _lib.AgbError("rows must be ordered by batch index (as torch_geometric's Batch collation gives)")
        self._validated = True

    def _resolve(self, lvl: _Level):
        """Make the host row count of a level available (one read-back if it is still device-only)."""
        if lvl.n is None:
            # the one unavoidable read-back of a lazily created level also carries the insert status (duplicates /
            # range / batch order): strided levels add to the range counter, so it is re-checked with every level
            vals = torch.cat([lvl.n_dev, self.levels[self.origin_ts].status]).tolist()
            lvl.n = int(vals[0])
            lvl.n_dev = None
            lvl.bound = lvl.n
            self._check_status(vals[1:])
        return lvl

    def level(self, ts: int) -> _Level:
        return self._resolve(self.levels[int(ts)])

    def num_rows(self, key: CoordinateMapKey) -> int:
        if key.tensor_stride == 0:
            return self.batch_size
        return self.level(key.tensor_stride).n

    def coords_of(self, key: CoordinateMapKey) -> torch.Tensor:
        if key.tensor_stride == 0:
            c = torch.zeros(self.batch_size, 4, dtype=torch.int32, device=self.device)
            c[:, 0] = torch.arange(self.batch_size, device=self.device, dtype=torch.int32)
            return c
        lvl = self.level(key.tensor_stride)
        return lvl.coords[: lvl.n]

    def batch_ptr(self, ts: int) -> torch.Tensor:
        lvl = self.levels[int(ts)]
        if lvl._ptr is None:
            p = torch.empty(self.batch_size + 1, dtype=torch.int32, device=self.device)
            _lib.call("agb_batch_ptr", _lib.ptr(lvl.coords), lvl.bound, _lib.ptr(lvl.n_dev), self.batch_size,
                      _lib.ptr(p), _lib.stream())
            lvl._ptr = p
        return lvl._ptr

    # ------------------------------------------------------------------ levels
    def _make_level(self, ts_in: int, ts_out: int) -> _Level:
        """Enqueue the construction of level ts_out from ts_in; the row count stays on the device."""
        src = self.levels[int(ts_in)]
        dev = self.device
        bound = src.bound
        if self.mode == "grid":
            bound = min(bound, self._grid_cells(ts_out))
        nb = max(src.bound, 1)
        flags = torch.empty(nb, dtype=torch.int32, device=dev)
        excl = torch.empty(nb, dtype=torch.int32, device=dev)
        scratch = torch.empty(_lib.scan_scratch_elems(nb), dtype=torch.int32, device=dev)
        out_coords = torch.empty(max(bound, 1), 4, dtype=torch.int32, device=dev)
        n_out_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        lvl = _Level(out_coords, None, bound, n_out_dev, ts_out)
        if self.mode == "grid":
            self._alloc_grid(lvl)
            cell = torch.empty(nb, dtype=torch.int64, device=dev)
            _lib.call("agb_grid_stride", _lib.ptr(src.coords), src.bound, _lib.ptr(src.n_dev), lvl.desc_host,
                      _lib.ptr(lvl.grid), _lib.ptr(cell), _lib.ptr(flags), _lib.ptr(excl), _lib.ptr(scratch),
                      _lib.ptr(out_coords), _lib.ptr(n_out_dev), _lib.ptr(self.levels[self.origin_ts].status),
                      _lib.stream())
        else:
            lvl.cap = _lib.hash_capacity(src.bound)
            lvl.keys = torch.empty(lvl.cap, dtype=torch.int64, device=dev)
            lvl.vals = torch.empty(lvl.cap, dtype=torch.int32, device=dev)
            slot = torch.empty(nb, dtype=torch.int32, device=dev)
            _lib.call("agb_coords_stride", _lib.ptr(src.coords), src.bound, _lib.ptr(src.n_dev), ts_out,
                      _lib.ptr(lvl.keys), _lib.ptr(lvl.vals), lvl.cap, _lib.ptr(slot), _lib.ptr(flags),
                      _lib.ptr(excl), _lib.ptr(scratch), _lib.ptr(out_coords), _lib.ptr(n_out_dev), None,
                      _lib.stream())
        self.levels[ts_out] = lvl
        return lvl

    def stride(self, ts_in: int, stride: int) -> int:
        """Create (or fetch) the level with tensor stride ts_in*stride: unique(floor(c/ts_out)*ts_out)."""
        stride = _as_int(stride)
        ts_out = int(ts_in) * stride
        if stride == 1 or ts_out in self.levels:
            return ts_out
        self._make_level(ts_in, ts_out)
        return ts_out

    def prefetch_strides(self, tensor_strides: Sequence[int], defer: bool = False):
        """Build the levels ts0 -> tensor_strides[0] -> tensor_strides[1] ... without any host synchronisation in
        between, then read every row count back at once.  defer=True: the read-back is only STARTED (asynchronous copy
        into pinned memory + an event); ``finish_prefetch`` completes it later, by which time the copy has long landed
        — the host never waits for the device (two-deep input pipeline of MinkowskiBaselineModel.prefetch_input)."""
        ts = self.origin_ts
        todo = []
        for t in tensor_strides:
            t = int(t)
            if t not in self.levels:
                todo.append(self._make_level(ts, t))
            ts = t
        pending = [lv for lv in todo if lv.n is None]
        if pending or not self._validated:
            # the single read-back: every new level's row count + the insert status (dup / range / order)
            status = self.levels[self.origin_ts].status
            vals_dev = torch.cat([lv.n_dev for lv in pending] + [status])
            if defer:
                # pinned landing buffers are recycled here: handing them back to torch's pinned allocator costs an event
                # record + queries per buffer (and a completion callback in the runtime per query of an unfinished event)
                pool = _PINNED_COUNTS.setdefault((vals_dev.numel(), vals_dev.dtype), [])
                host = pool.pop() if pool else torch.empty(vals_dev.shape, dtype=vals_dev.dtype, pin_memory=True)
                host.copy_(vals_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._deferred = (pending, host, ev, vals_dev)
                return
            self._apply_counts(pending, vals_dev.tolist())

    def _apply_counts(self, pending, vals):
        for lv, c in zip(pending, vals):
            lv.n, lv.n_dev, lv.bound = int(c), None, int(c)
        self._check_status(vals[len(pending):])

    def finish_prefetch(self):
        """Complete a deferred ``prefetch_strides`` (no-op otherwise)."""
        d = getattr(self, "_deferred", None)
        if d is None:
            return
        pending, host, ev, _keep = d
        self._deferred = None
        ev.synchronize()
        vals = host.tolist()
        _PINNED_COUNTS.setdefault((host.numel(), host.dtype), []).append(host)
        self._apply_counts(pending, vals)

    def grid_probe(self, ts, K, stride=1, dilation=1):
        """(coords, grid, desc) of level ts if a K^3 stride-1 convolution can probe the level's dense grid directly
        (grid mode, odd K within the halo, no dilation), else None."""
        if self.mode != "grid" or stride != 1 or dilation != 1 or K % 2 == 0 or K // 2 > GRID_HALO:
            return None
        lvl = self.level(ts)
        if lvl.grid is None or self.batch_size > 0xffff:
            return None
        return lvl.coords, lvl.grid, lvl.desc_host

    def prebuild(self, specs, opts=None):
        """Build kernel maps ahead of the forward pass (opts: the model's KernelOptions, for the tile tables). specs: iterable of (ts_in, K, stride, dilation, need_T[, probe
        [, (cin, cout)]]): probe = True marks a layer that reads the dense grid itself when it can (it writes its own map
        then); (cin, cout) = padded widths of a stride-1 layer on the map, whose work-balanced tile tables (forward and
        data gradient) are built here too instead of inside the first convolution that uses them."""
        for spec in specs:
            ts_in, K, s, d, need_t = spec[:5]
            if len(spec) > 5 and spec[5] and not need_t and self.grid_probe(ts_in, K, s, d) is not None:
                continue
            nbr = self.kernel_map(ts_in, K, s, d)
            if len(spec) > 6 and spec[6] is not None and _as_int(s) == 1 and _as_int(K) % 2 == 1 and _as_int(K) > 1:
                from . import sparse_ops
                opts = opts or sparse_ops.current()
                if not opts.low_precision:      # (the bf16 kernels do not take the tables)
                    cin, cout = spec[6]
                    n = self.levels[int(ts_in)].n
                    sparse_ops.cmp_tile_table(nbr, n, _as_int(K) ** 3, cin, cout, cin, cout, opts)
                    sparse_ops.cmp_tile_table(nbr, n, _as_int(K) ** 3, cout, cin, cout, cin, opts)   # data gradient: same map, flipped
            if need_t:
                self.transposed_map(ts_in, K, s, d)
                if s > 1:
                    self.transposed_plan(ts_in, K, s, d)
        for ts in list(self.levels):
            self.batch_ptr(ts)

    def tensors(self):
        for lvl in self.levels.values():
            for t in (lvl.coords, lvl.grid, lvl.keys, lvl.vals, lvl._ptr, lvl.status, lvl.n_dev):
                if t is not None:
                    yield t
        for m in self.kernel_maps.values():
            if isinstance(m, tuple):      # transposed plan: (perm, tile_cls, cls_tab, max_tiles)
                yield m[0]
                yield m[1]
                continue
            yield m
            p = getattr(m, "agb_pairs", None)
            if p is not None:
                yield p
            for tab in getattr(m, "agb_tiles", {}).values():      # work-balanced tile tables (sparse_ops.cmp_tile_table)
                yield tab

    def record_stream(self, stream):
        """Tell the caching allocator that `stream` uses every tensor of this manager (needed when the manager was
        built on a side stream and is consumed on the compute stream)."""
        for t in self.tensors():
            t.record_stream(stream)

    # -------------------------------------------------------------- kernel maps
    def _lookup(self, q: _Level, table: _Level, K, step, sign, require_multiple_of, count_pairs):
        self._resolve(q)
        if not self._validated:
            # first kernel map of a manager that was not built through prefetch_strides (e.g. a plain
            # ME.SparseTensor(features, coordinates) in user code): duplicates / out-of-range rows / unordered batches
            # must not pass silently — one read-back, like ME's own host synchronisation at this point
            self.validate()
        nbr = torch.empty(K ** 3, max(q.n, 1), dtype=torch.int32, device=self.device)
        # sharded counter (64 slots on separate lines); the kernel-map size is the sum — read only by profiling code
        pairs = torch.zeros(64 * 16, dtype=torch.int64, device=self.device) if count_pairs else None
        if self.mode == "grid":
            _lib.call("agb_grid_kernel_map", _lib.ptr(q.coords), q.n, None, K, step, sign, table.desc_host,
                      _lib.ptr(table.grid), _lib.ptr(nbr), nbr.stride(0), _lib.ptr(pairs), _lib.stream())
        else:
            _lib.call("agb_kernel_map", _lib.ptr(q.coords), q.n, None, K, step, sign, require_multiple_of,
                      _lib.ptr(table.keys), _lib.ptr(table.vals), table.cap, _lib.ptr(nbr), nbr.stride(0),
                      _lib.ptr(pairs), _lib.stream())
        if count_pairs:
            nbr.agb_pairs = pairs  # device-side kernel-map size (ME's kernel-map size); read only by profiling code
        return nbr

    def kernel_map(self, ts_in: int, kernel_size: int, stride: int = 1, dilation: int = 1) -> torch.Tensor:
        """nbr[K^3, N_out]: row (at level ts_in) of out_coord + offset_k * ts_in * dilation, or -1."""
        K, s, d = _as_int(kernel_size), _as_int(stride), _as_int(dilation)
        key = ("fwd", int(ts_in), K, s, d)
        if key not in self.kernel_maps:
            ts_out = self.stride(ts_in, s)
            self.kernel_maps[key] = self._lookup(self.levels[ts_out], self.levels[int(ts_in)], K, int(ts_in) * d, 1,
                                                 0, True)
        return self.kernel_maps[key]

    def transposed_plan(self, ts_in: int, kernel_size: int, stride: int, dilation: int = 1):
        """Class-partitioned row order for the data gradient of a strided operator: rows of level ts_in grouped by
        lattice parity (c/ts_in mod stride per axis); a row of a given class can only be reached through the few
        kernel offsets o with o = c/ts_in (mod stride) — 1/2/4/8 of 27 for K=3, s=2.
        Returns (perm int32[n + s^3*64], tile_cls int32[max_tiles], cls_tab int32[s^3, 1+K^3], max_tiles)."""
        K, s, d = _as_int(kernel_size), _as_int(stride), _as_int(dilation)
        key = ("plan", int(ts_in), K, s, d)
        if key not in self.kernel_maps:
            src = self._resolve(self.levels[int(ts_in)])
            n, ncls = src.n, s ** 3
            max_tiles = n // 64 + ncls + 1
            perm = torch.empty(n + ncls * 64, dtype=torch.int32, device=self.device)
            tile_cls = torch.empty(max_tiles, dtype=torch.int32, device=self.device)
            scratch = torch.empty(256, dtype=torch.int32, device=self.device)
            _lib.call("agb_parity_partition", _lib.ptr(src.coords), n, int(ts_in), s, _lib.ptr(perm),
                      _lib.ptr(tile_cls), max_tiles, _lib.ptr(scratch), _lib.stream())
            self.kernel_maps[key] = (perm, tile_cls, _class_table(K, s, d, self.device), max_tiles)
        return self.kernel_maps[key]

    def transposed_map(self, ts_in: int, kernel_size: int, stride: int = 1, dilation: int = 1) -> torch.Tensor:
        """nbrT[K^3, N_in]: row (at level ts_out) of in_coord - offset_k * ts_in * dilation, or -1."""
        K, s, d = _as_int(kernel_size), _as_int(stride), _as_int(dilation)
        key = ("bwd", int(ts_in), K, s, d)
        if key not in self.kernel_maps:
            ts_out = self.stride(ts_in, s)
            self.kernel_maps[key] = self._lookup(self.levels[int(ts_in)], self.levels[ts_out], K, int(ts_in) * d, -1,
                                                 ts_out, False)
        return self.kernel_maps[key]
