"""KPConv index path on the GPU behind the reference's Python wrappers
(torch_points3d/modules/KPConv/common.py:8-157):

    batch_neighbors(queries, supports, q_batches, s_batches, radius)          -> int32 [Nq, max_count]
    batch_grid_subsampling(points, batches_len, features=None, labels=None, sampleDl=0.1, max_p=0, verbose=0,
                           random_grid_orient=True)                           -> (points, lengths[, features])

numpy in -> numpy out (drop-in for the reference's call sites, models/instance/kpconv.py:184,200,210);
device tensors in -> device tensors out (no host round trip: what the model wrapper uses).
Results are bit-identical to the reference's C++ (same float32 operation order, no FMA), except that
  * neighbours at EXACTLY equal distance are ordered by ascending index (unspecified in the reference), and
  * subsampled points are emitted in canonical order (cell key ascending per cloud) instead of the iteration order
    of the reference's unordered_map.
Extra keyword ``rotations=`` injects the per-cloud grid orientation matrices that the reference draws from
``np.random`` (common.py:59-66); when omitted they are drawn exactly like the reference does.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_P = _lib.ptr
_V, _I, _F = _lib.c_void_p, _lib.c_int, _lib.c_float
_lib.declare("agb_elem_bbox", [_V, _V, _I, _I, _V, _V, _V])
_lib.declare("agb_elem_of_row", [_V, _I, _I, _V, _V])
_lib.declare("agb_rotate_points", [_V, _V, _V, _I, _I, _V, _V])
_lib.declare("agb_ball_grid_build", [_V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V])
_lib.declare("agb_ball_query_count", [_V, _I, _V, _V, _V, _V, _V, _F, _V, _V, _V])
_lib.declare("agb_ball_query_fill", [_V, _I, _V, _V, _V, _V, _V, _F, _I, _I, _V, _V, _V])
_lib.declare("agb_ball_query_offsets", [_V, _I, _V, _V, _V])
_lib.declare("agb_ball_query_fill_csr", [_V, _I, _V, _V, _V, _V, _V, _F, _I, _V, _V, _I, _V, _V])
_lib.declare("agb_ball_query_fill_csr_m", [_V, _I, _V, _V, _V, _V, _V, _F, _I, _V, _V, _I, _I, _V, _V])
_lib.declare("agb_csr_to_padded", [_V, _V, _I, _I, _I, _V, _V])
_lib.declare("agb_grid_subsample_workspace_bytes", [_I, _I, _I])
_lib.declare("agb_grid_subsample_ws", [_V, _V, _I, _I, _V, _V, _I, _F, _I, _V, _V, _V, _V, _V, _V, _V])

MAX_CELLS = 1 << 27


def _dev():
    if not torch.cuda.is_available():
        raise _lib.AgbError("the KPConv index kernels need a HIP device (there is no CPU fallback in the product path)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(device=_dev() if not a.is_cuda else a.device, dtype=dtype).contiguous(), True
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=_dev(), dtype=dtype).contiguous(), False


def _lengths(v):
    if isinstance(v, torch.Tensor):
        v = v.cpu().numpy()
    return np.asarray(v, dtype=np.int64).reshape(-1)


class _PinnedRing:
    """Small host <-> device transfers through a ring of pinned buffers: a copy from / to PAGEABLE memory makes the
    runtime wait for the whole device (every stream) — with the training step of the previous batch queued on the compute
    stream that stalled the input pyramid's side stream for whole steps (50-150 ms hiccups in the KPConv loop).  A slot
    is reused only after the device has executed the copy that used it (one event per slot)."""
    SLOTS, BYTES = 128, 8192

    def __init__(self):
        self.buf = torch.empty(self.SLOTS, self.BYTES, dtype=torch.uint8).pin_memory()
        self.done = [None] * self.SLOTS
        self.next = 0

    def slot(self):
        i = self.next % self.SLOTS
        self.next += 1
        if self.done[i] is not None:
            self.done[i].synchronize()
        return i

    def h2d(self, arr, device):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.BYTES or arr.nbytes == 0:
            return torch.from_numpy(arr).to(device)
        i = self.slot()
        host = self.buf[i, :arr.nbytes].view(torch.from_numpy(arr).dtype).reshape(arr.shape)
        host.copy_(torch.from_numpy(arr))
        out = host.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[i] = ev
        return out

    def d2h(self, t):
        """Values of a small device tensor as a list; waits only for the stream it was produced on."""
        nbytes = t.numel() * t.element_size()
        if nbytes > self.BYTES or nbytes == 0:
            return t.tolist()
        i = self.slot()
        host = self.buf[i, :nbytes].view(t.dtype).reshape(t.shape)
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ev.synchronize()
        self.done[i] = None
        return host.tolist()


_RING = None


def _ring():
    global _RING
    if _RING is None:
        _RING = _PinnedRing()
    return _RING


def h2d_small(arr, device):
    return _ring().h2d(arr, device) if torch.device(device).type == "cuda" else torch.from_numpy(np.asarray(arr)).to(device)


_PTR_CACHE = {}      # (row offsets as bytes, device, stream) -> device tensor: read-only once uploaded


def _ptr_tensor(lens, device):
    """Row offsets int32 [B + 1] of a batch on the device.  One batch's offsets are asked for again and again (the input
    pyramid of a KPConv batch: 30 times for 5 distinct length vectors — query and support side of every search, the
    subsampling, the pooling searches; a sparse batch: by the chain, the voxeliser and the model's staging), each an upload of
    ~35 us of host time: the tensor of the same values requested on the same stream is handed out again (stream order
    guarantees the upload has happened before any later use on that stream)."""
    p = np.zeros(len(lens) + 1, dtype=np.int32)
    np.cumsum(lens, out=p[1:])
    dev = torch.device(device)
    if dev.type != "cuda":
        return h2d_small(p, device)
    key = (p.tobytes(), dev.index, _lib.stream())
    t = _PTR_CACHE.get(key)
    if t is None:
        if len(_PTR_CACHE) >= 64:
            _PTR_CACHE.clear()
        t = _PTR_CACHE[key] = h2d_small(p, device)
    return t


def elem_bbox(points, ptr, B):
    """Per-cloud bounding boxes (device): float [B, 6] = (min xyz, max xyz)."""
    n = points.shape[0]
    ordb = torch.empty(6 * B, dtype=torch.int32, device=points.device)
    out = torch.empty(B, 6, dtype=torch.float32, device=points.device)
    _lib.call("agb_elem_bbox", _P(points), _P(ptr), B, n, _P(ordb), _P(out), _lib.stream())
    return out


def _elem_of_row(ptr, B, n, device):
    e = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    _lib.call("agb_elem_of_row", _P(ptr), B, n, _P(e), _lib.stream())
    return e


def _check_shapes(queries, supports, q_batches, s_batches):
    # same checks (and messages) as cpp_neighbors/wrapper.cpp:127-171
    if queries.dim() != 2 or queries.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
    if supports.dim() != 2 or supports.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
    if len(q_batches) != len(s_batches):
        raise RuntimeError("Wrong number of batch elements")
    if int(q_batches.sum()) != queries.shape[0] or int(s_batches.sum()) != supports.shape[0]:
        raise RuntimeError("Wrong dimensions : batch lengths do not sum to the number of points")


def read_back(*tensors):
    """ONE host synchronisation (of the current stream only) for several small device tensors: their values as lists."""
    if not tensors:
        return []
    if all(not t.is_floating_point() for t in tensors):
        flat = torch.cat([t.reshape(-1).to(torch.int64) for t in tensors])
    elif all(t.is_floating_point() for t in tensors):
        flat = torch.cat([t.reshape(-1).double() for t in tensors])
    else:
        return [_ring().d2h(t.reshape(-1)) for t in tensors]
    vals, out, off = _ring().d2h(flat), [], 0
    for t in tensors:
        out.append(vals[off:off + t.numel()])
        off += t.numel()
    return out


class PendingRead:
    """``read_back`` without the wait: the small device tensors are copied (one asynchronous copy into recycled pinned
    memory, on the current stream) and an event is recorded; ``ready()`` polls it, ``values()`` returns what ``read_back``
    would have (waiting only if the copy has not landed yet) — the building block of the wait-free input pyramid."""
    _POOL = {}

    def __init__(self, tensors):
        self.shapes = [t.numel() for t in tensors]
        self.host = self.event = None
        if not tensors:
            return
        if all(not t.is_floating_point() for t in tensors):
            flat = torch.cat([t.reshape(-1).to(torch.int64) for t in tensors])
        else:
            flat = torch.cat([t.reshape(-1).double() for t in tensors])
        self.is_int = not flat.is_floating_point()
        pool = PendingRead._POOL.setdefault((flat.numel(), flat.dtype), [])
        self.host = pool.pop() if pool else torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
        self.host.copy_(flat, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self._keep = flat

    def ready(self):
        return self.event is None or self.event.query()

    def values(self):
        if self.host is None:
            return []
        self.event.synchronize()
        vals = self.host.tolist()
        PendingRead._POOL.setdefault((self.host.numel(), self.host.dtype), []).append(self.host)
        self.host = self._keep = None
        out, off = [], 0
        for n in self.shapes:
            out.append(vals[off:off + n])
            off += n
        return out


class Neighbors:
    """The radius neighbours of nq queries as RAGGED rows — what the kernels of this library walk (the
    ball query writes sum(counts) * 4 bytes instead of the nq x max_count matrix the reference pads to,
    cpp_neighbors/neighbors.cpp:319-325): ``row_ptr`` int32[nq + 1], ``indices`` int32[total], every row sorted by
    (distance, index).  ``limit``: the reference's neighborhood_limits crop (rows are cut at that many entries).
    ``padded()`` builds the reference's int32 [nq, width] matrix (shadow index ``ns`` behind every row) on demand;
    ``shape`` is that matrix's shape."""
    __slots__ = ("row_ptr", "indices", "nq", "ns", "max_count", "max_count_dev", "limit", "agb_symmetric", "_padded")

    def __init__(self, row_ptr, indices, nq, ns, max_count, max_count_dev):
        self.row_ptr, self.indices, self.nq, self.ns = row_ptr, indices, int(nq), int(ns)
        self.max_count, self.max_count_dev = int(max_count), max_count_dev
        self.limit, self.agb_symmetric, self._padded = 0x7fffffff, False, None

    @property
    def width(self):
        return min(self.max_count, self.limit)

    @property
    def shape(self):
        return (self.nq, self.width)

    @property
    def device(self):
        return self.row_ptr.device

    def cropped(self, limit):
        out = Neighbors(self.row_ptr, self.indices, self.nq, self.ns, self.max_count, self.max_count_dev)
        out.limit, out.agb_symmetric = min(self.limit, int(limit)), self.agb_symmetric and int(limit) >= self.max_count
        return out

    def padded(self):
        if self._padded is None:
            out = torch.empty(self.nq, self.width, dtype=torch.int32, device=self.row_ptr.device)
            _lib.call("agb_csr_to_padded", _P(self.row_ptr), _P(self.indices), self.nq, self.width, self.ns, _P(out),
                      _lib.stream())
            self._padded = out
        return self._padded

    def tensors(self):
        return [t for t in (self.row_ptr, self.indices, self.max_count_dev, self._padded) if t is not None]

    def cpu(self):
        """The padded matrix on the host (what code written against the reference's matrices asks for)."""
        return self.padded().cpu()

    def numel(self):
        return self.nq * self.width


class NeighborJob:
    """A radius search whose count pass is enqueued; ``max_count`` (device int32[1]) is the width of the padded matrix,
    ``row_ptr`` the exclusive scan of the counts (``row_ptr[nq]`` = total number of neighbours)."""
    __slots__ = ("q", "q_elem", "origin_cs", "dims_c", "cell_start", "sorted_pts", "radius", "ns", "nq", "max_count",
                 "counts", "row_ptr")


def neighbors_begin(q, s, ql, sl, radius, bounds):
    """Enqueue the cell grid of the supports and the count pass.  bounds: (min xyz, max xyz) covering every support."""
    dev = q.device
    B, nq, ns = len(ql), q.shape[0], s.shape[0]
    radius = float(np.float32(radius))
    if nq == 0 or ns == 0:
        raise RuntimeError("Error")  # the reference raises on an empty result (wrapper.cpp:201-205)
    q_ptr, s_ptr = _ptr_tensor(ql, dev), _ptr_tensor(sl, dev)
    lo, hi = list(bounds[:3]), list(bounds[3:])
    cs = radius * 1.001
    while True:
        dims = [int(np.floor((h - l) / cs)) + 1 for l, h in zip(lo, hi)]
        if B * dims[0] * dims[1] * dims[2] <= MAX_CELLS:
            break
        cs *= 2.0  # coarser cells stay correct for as long as cell >= radius
    job = NeighborJob()
    job.origin_cs = (ctypes.c_float * 4)(lo[0], lo[1], lo[2], cs)
    job.dims_c = (ctypes.c_int32 * 4)(dims[0], dims[1], dims[2], B)
    cells = B * dims[0] * dims[1] * dims[2]
    job.cell_start = torch.empty(cells + 1, dtype=torch.int32, device=dev)
    cell_fill = torch.empty(cells + 1, dtype=torch.int32, device=dev)
    cell_of = torch.empty(ns, dtype=torch.int32, device=dev)
    job.sorted_pts = torch.empty(ns, 4, dtype=torch.float32, device=dev)
    scratch = torch.empty(_lib.scan_scratch_elems(cells + 1), dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.call("agb_ball_grid_build", _P(s), ns, _P(s_ptr), job.origin_cs, job.dims_c, _P(job.cell_start),
              _P(job.sorted_pts), _P(cell_of), _P(cell_fill), _P(scratch), _P(total), _lib.stream())
    job.q, job.q_elem = q, _elem_of_row(q_ptr, B, nq, dev)
    job.counts = torch.empty(nq, dtype=torch.int32, device=dev)
    job.max_count = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.call("agb_ball_query_count", _P(q), nq, _P(job.q_elem), job.origin_cs, job.dims_c, _P(job.cell_start),
              _P(job.sorted_pts), radius, _P(job.counts), _P(job.max_count), _lib.stream())
    job.row_ptr = torch.empty(nq + 1, dtype=torch.int32, device=dev)
    scan = torch.empty(_lib.scan_scratch_elems(nq), dtype=torch.int32, device=dev)
    _lib.call("agb_ball_query_offsets", _P(job.counts), nq, _P(job.row_ptr), _P(scan), _lib.stream())
    job.radius, job.ns, job.nq = radius, ns, nq
    return job


def neighbors_finish_csr(job, width, total):
    """Fill pass into ragged rows for a job whose ``max_count`` and ``row_ptr[nq]`` have been read back."""
    width, total = int(width), int(total)
    if width == 0:
        raise RuntimeError("Error")
    if width > 1024:
        raise _lib.AgbError("a neighbourhood holds more than 1024 points: beyond the kernel's LDS capacity")
    dev = job.q.device
    indices = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    status = torch.empty(4, dtype=torch.int32, device=dev)
    # (the longest list is known here: the kernel sizes its LDS slab from it instead of the 1024-key worst case)
    _lib.call("agb_ball_query_fill_csr_m", _P(job.q), job.nq, _P(job.q_elem), job.origin_cs, job.dims_c, _P(job.cell_start),
              _P(job.sorted_pts), job.radius, job.ns, _P(job.row_ptr), _P(indices), total, width, _P(status), _lib.stream())
    return Neighbors(job.row_ptr, indices, job.nq, job.ns, width, job.max_count)


def neighbors_finish(job, width):
    """Fill pass for a job whose ``max_count`` has been read back (``width``)."""
    width = int(width)
    if width == 0:
        raise RuntimeError("Error")
    if width > 1024:
        raise _lib.AgbError("a neighbourhood holds more than 1024 points: beyond the kernel's LDS capacity")
    out = torch.empty(job.nq, width, dtype=torch.int32, device=job.q.device)
    status = torch.empty(4, dtype=torch.int32, device=job.q.device)
    _lib.call("agb_ball_query_fill", _P(job.q), job.nq, _P(job.q_elem), job.origin_cs, job.dims_c, _P(job.cell_start),
              _P(job.sorted_pts), job.radius, job.ns, width, _P(out), _P(status), _lib.stream())
    return out


def support_bounds(s, sl, cloud_diag=False):
    """(min xyz, max xyz) over all clouds: one host read.  cloud_diag: a seventh value, the largest bounding-box diagonal of
    a single cloud (what bounds the extent of a cloud rotated about any point, whatever the clouds' world positions)."""
    dev = s.device
    bb = elem_bbox(s, _ptr_tensor(sl, dev), len(sl))
    if not cloud_diag:
        lo, hi = read_back(bb[:, :3].min(0).values, bb[:, 3:].max(0).values)
        return tuple(lo) + tuple(hi)
    lo, hi, dg = read_back(bb[:, :3].min(0).values, bb[:, 3:].max(0).values,
                           (bb[:, 3:] - bb[:, :3]).norm(dim=1).max().reshape(1))
    return tuple(lo) + tuple(hi) + (dg[0],)


def batch_neighbors(queries, supports, q_batches, s_batches, radius, bounds=None):
    """bounds: optional float (min_x, min_y, min_z, max_x, max_y, max_z) covering every support (saves the
    bounding-box read-back)."""
    q, q_is_t = _to_dev(queries, torch.float32)
    s, _ = _to_dev(supports, torch.float32)
    ql, sl = _lengths(q_batches), _lengths(s_batches)
    _check_shapes(q, s, ql, sl)
    if q.shape[0] == 0 or s.shape[0] == 0:
        raise RuntimeError("Error")  # the reference raises on an empty result (wrapper.cpp:201-205)
    if bounds is None:
        bounds = support_bounds(s, sl)
    job = neighbors_begin(q, s, ql, sl, radius, bounds)
    # the padded matrix the reference returns is as wide as the fullest neighbourhood: one host read
    out = neighbors_finish(job, read_back(job.max_count)[0][0])
    return out if q_is_t else out.cpu().numpy()


def batch_neighbors_ragged(queries, supports, q_batches, s_batches, radius, bounds=None):
    """The same search as ragged rows (``Neighbors``: row_ptr / indices on the device, ``.padded()`` for the matrix)."""
    q, _ = _to_dev(queries, torch.float32)
    s, _ = _to_dev(supports, torch.float32)
    ql, sl = _lengths(q_batches), _lengths(s_batches)
    _check_shapes(q, s, ql, sl)
    if q.shape[0] == 0 or s.shape[0] == 0:
        raise RuntimeError("Error")
    if bounds is None:
        bounds = support_bounds(s, sl)
    job = neighbors_begin(q, s, ql, sl, radius, bounds)
    width, total = read_back(job.max_count, job.row_ptr[-1:])
    return neighbors_finish_csr(job, width[0], total[0])


def create_3D_rotations(axis, angle):
    """Rotation matrices from axes and angles (same formula and float64 evaluation order as
    torch_points3d/modules/KPConv/kernel_points.py:38-69)."""
    t1 = np.cos(angle)
    t2 = 1 - t1
    t3 = axis[:, 0] * axis[:, 0]
    t6 = t2 * axis[:, 0]
    t7 = t6 * axis[:, 1]
    t8 = np.sin(angle)
    t9 = t8 * axis[:, 2]
    t11 = t6 * axis[:, 2]
    t12 = t8 * axis[:, 1]
    t15 = axis[:, 1] * axis[:, 1]
    t19 = t2 * axis[:, 1] * axis[:, 2]
    t20 = t8 * axis[:, 0]
    t24 = axis[:, 2] * axis[:, 2]
    R = np.stack([t1 + t2 * t3, t7 - t9, t11 + t12, t7 + t9, t1 + t2 * t15, t19 - t20, t11 - t12, t19 + t20,
                  t1 + t2 * t24], axis=1)
    return np.reshape(R, (-1, 3, 3))


def random_grid_rotations(B):
    """Draws from np.random in the reference's order (common.py:59-72): theta, phi, alpha, each rand(B)."""
    theta = np.random.rand(B) * 2 * np.pi
    phi = (np.random.rand(B) - 0.5) * np.pi
    u = np.vstack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
    alpha = np.random.rand(B) * 2 * np.pi
    return create_3D_rotations(u.T, alpha).astype(np.float32)


def _rotate(points, elem, R_dev, transpose):
    out = torch.empty_like(points)
    _lib.call("agb_rotate_points", _P(points), _P(elem), _P(R_dev), points.shape[0], int(transpose), _P(out),
              _lib.stream())
    return out


class SubsampleJob:
    __slots__ = ("out_p", "out_f", "out_ptr", "status", "elem", "B")


def subsample_begin(p, f, lens, dl, ext):
    """Enqueue the grid subsampling of stacked clouds; ext: upper bound of every cloud's extent per axis (sizes the cell
    grid without a read-back).  ``subsample_finish`` needs ``out_ptr`` and ``status[:1]`` on the host."""
    dev = p.device
    B, n = len(lens), p.shape[0]
    ptr = _ptr_tensor(lens, dev)
    elem = _elem_of_row(ptr, B, n, dev)
    cap = 1
    for e in ext:
        cap *= int(np.floor(e / dl)) + 3
    if B * cap > MAX_CELLS:
        raise _lib.AgbError(f"grid subsampling would need {B * cap} cells: sampleDl too small for these clouds")
    i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)  # noqa: E731
    ws = torch.empty(_lib.size_call("agb_grid_subsample_workspace_bytes", n, B, cap), dtype=torch.uint8, device=dev)
    job = SubsampleJob()
    job.out_p = torch.empty(max(n, 1), 3, dtype=torch.float32, device=dev)
    fdim = 0 if f is None else f.shape[1]
    job.out_f = torch.empty(max(n, 1), fdim, dtype=torch.float32, device=dev) if f is not None else None
    job.out_ptr, n_out, job.status = i32(B + 1), i32(1), i32(4)
    job.elem, job.B = elem, B
    _lib.call("agb_grid_subsample_ws", _P(p), _P(f), fdim, n, _P(ptr), _P(elem), B, float(np.float32(dl)), cap, _P(ws),
              _P(job.out_p), _P(job.out_f), _P(job.out_ptr), _P(n_out), _P(job.status), _lib.stream())
    return job


def subsample_finish(job, out_ptr_host, status0):
    if int(status0):
        raise _lib.AgbError("grid subsampling: a cloud exceeds the reserved cell capacity")
    optr = np.asarray(out_ptr_host, dtype=np.int64)
    m = int(optr[-1])
    return job.out_p[:m], (None if job.out_f is None else job.out_f[:m]), np.diff(optr).astype(np.int32), job.elem


def _subsample_core(p, f, lens, dl, bounds_hint=None):
    dev = p.device
    B, n = len(lens), p.shape[0]
    if n == 0:      # no points at all: every cloud stays empty
        return p[:0], (None if f is None else f[:0]), np.zeros(B, dtype=np.int32), torch.empty(0, dtype=torch.int32, device=dev)
    if bounds_hint is None:
        bb = elem_bbox(p, _ptr_tensor(lens, dev), B)
        ext = (bb[:, 3:] - bb[:, :3]).max(0).values.tolist()   # one host read
    else:
        ext = list(bounds_hint)
    job = subsample_begin(p, f, lens, dl, ext)
    optr, st = read_back(job.out_ptr, job.status[:1])         # one host read: sizes of the subsampled clouds
    return subsample_finish(job, optr, st[0])


def rotate_points(points, lens, R, transpose):
    """points @ R[cloud] (or its transpose) for stacked clouds; R float32 [B, 3, 3] (numpy)."""
    dev = points.device
    R_dev = h2d_small(np.ascontiguousarray(R, dtype=np.float32), dev)
    ptr = _ptr_tensor(lens, dev)
    elem = _elem_of_row(ptr, len(lens), points.shape[0], dev)
    return _rotate(points.contiguous(), elem, R_dev, transpose)


def batch_grid_subsampling(points, batches_len, features=None, labels=None, sampleDl=0.1, max_p=0, verbose=0,
                           random_grid_orient=True, rotations=None):
    if labels is not None:
        raise NotImplementedError("label voting is not on the AGB regression path (and is broken for ldim > 1 in "
                                  "the reference, grid_subsampling.cpp:157-158)")
    p, is_t = _to_dev(points, torch.float32)
    f = None if features is None else _to_dev(features, torch.float32)[0]
    lens = _lengths(batches_len)
    if p.dim() != 2 or p.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    if int(lens.sum()) != p.shape[0]:
        raise RuntimeError("Wrong dimensions : batch lengths do not sum to the number of points")
    B = len(lens)
    R = None
    if random_grid_orient:
        R = random_grid_rotations(B) if rotations is None else np.asarray(rotations, dtype=np.float32)
        R_dev = h2d_small(np.ascontiguousarray(R), p.device)
        ptr = _ptr_tensor(lens, p.device)
        elem = _elem_of_row(ptr, B, p.shape[0], p.device)
        p = _rotate(p, elem, R_dev, False)
    sp, sf, slen, _ = _subsample_core(p, f, lens, sampleDl)
    if max_p and max_p > 0:
        keep, off = [], 0
        for n_b in slen:
            keep.append(torch.arange(off, off + min(int(n_b), int(max_p)), device=sp.device))
            off += int(n_b)
        keep = torch.cat(keep)
        sp, sf = sp[keep], (None if sf is None else sf[keep])
        slen = np.minimum(slen, int(max_p)).astype(np.int32)
    if random_grid_orient:
        optr = _ptr_tensor(slen, sp.device)
        oelem = _elem_of_row(optr, B, sp.shape[0], sp.device)
        sp = _rotate(sp.contiguous(), oelem, R_dev, True)
    if is_t:
        res = (sp, torch.from_numpy(slen)) + ((sf,) if sf is not None else ())
    else:
        res = (sp.cpu().numpy(), slen) + ((sf.cpu().numpy(),) if sf is not None else ())
    return res


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1, verbose=0):
    """Single cloud (common.py:8-36)."""
    n = points.shape[0]
    res = batch_grid_subsampling(points, [n], features=features, labels=labels, sampleDl=sampleDl,
                                 random_grid_orient=False)
    return res[0] if features is None else (res[0], res[2])
