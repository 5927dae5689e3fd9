"""ORACLE (test infrastructure only). ctypes front-ends of
  * oracle/_build/liboracle_kpconv.so — this repo's CPU restatement (kpconv_index_ref.cpp)
  * oracle/_ref/libref_kpconv.so      — the reference's own C++ (built by oracle/Makefile from /root/reference)
Same call shapes as the reference's Python wrappers (torch_points3d/modules/KPConv/common.py:39-157)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE = os.path.join(HERE, "_build", "liboracle_kpconv.so")
_REF = os.path.join(HERE, "_ref", "libref_kpconv.so")
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def _build():
    subprocess.run(["make", "-s"], cwd=HERE, check=True)


def _load(path):
    if not os.path.exists(path):
        _build()
    return ctypes.CDLL(path) if os.path.exists(path) else None


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        _oracle = _load(_ORACLE)
        if _oracle is None:
            raise RuntimeError("could not build oracle/_build/liboracle_kpconv.so (needs g++)")
    return _oracle


def ref_available():
    return os.path.exists(_REF) or os.path.isdir("/root/reference")


def ref_lib():
    global _ref
    if _ref is None:
        _ref = _load(_REF)
        if _ref is None:
            raise RuntimeError("oracle/_ref/libref_kpconv.so is not built and /root/reference is not mounted")
        _ref.ref_free.argtypes = [ctypes.c_void_p]
    return _ref


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def batch_neighbors(queries, supports, q_batches, s_batches, radius):
    """Restatement of batch_query: int32 [Nq, max_count], padded with Ns."""
    lib = oracle_lib()
    q, s = _c(queries, np.float32), _c(supports, np.float32)
    qb, sb = _c(q_batches, np.int32), _c(s_batches, np.int32)
    counts = np.zeros(len(q), dtype=np.int32)
    lib.oracle_ball_query_count.restype = ctypes.c_int
    args = [_ptr(q, _f32p), len(q), _ptr(s, _f32p), len(s), _ptr(qb, _i32p), _ptr(sb, _i32p), len(qb),
            ctypes.c_float(radius)]
    width = lib.oracle_ball_query_count(*args, _ptr(counts, _i32p))
    out = np.empty((len(q), width), dtype=np.int32)
    lib.oracle_ball_query_fill(*args, width, _ptr(out, _i32p))
    return out


def batch_grid_subsampling(points, batches, features=None, sampleDl=0.1, max_p=0, order="canonical",
                           return_keys=False):
    lib = oracle_lib()
    p, b = _c(points, np.float32), _c(batches, np.int32)
    n = len(p)
    f = None if features is None else _c(features, np.float32)
    fdim = 0 if f is None else f.shape[1]
    op = np.empty((max(n, 1), 3), dtype=np.float32)
    of = np.empty((max(n, 1), max(fdim, 1)), dtype=np.float32) if f is not None else None
    ob = np.zeros(len(b), dtype=np.int32)
    keys = np.zeros(max(n, 1), dtype=np.int64)
    lib.oracle_grid_subsample.restype = ctypes.c_int
    m = lib.oracle_grid_subsample(_ptr(p, _f32p), n, None if f is None else _ptr(f, _f32p), fdim, _ptr(b, _i32p),
                                  len(b), ctypes.c_float(sampleDl), int(max_p), 0 if order == "canonical" else 1,
                                  _ptr(op, _f32p), None if of is None else _ptr(of, _f32p), _ptr(ob, _i32p),
                                  keys.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
    res = [op[:m].copy(), ob]
    if f is not None:
        res.append(of[:m, :fdim].copy())
    if return_keys:
        res.append(keys[:m].copy())
    return tuple(res)


# ---------------------------------------------------------------- the reference itself (only where _ref is built)
def ref_batch_neighbors(queries, supports, q_batches, s_batches, radius):
    lib = ref_lib()
    q, s = _c(queries, np.float32), _c(supports, np.float32)
    qb, sb = _c(q_batches, np.int32), _c(s_batches, np.int32)
    out = _i32p()
    lib.ref_batch_neighbors.restype = ctypes.c_int
    width = lib.ref_batch_neighbors(_ptr(q, _f32p), len(q), _ptr(s, _f32p), len(s), _ptr(qb, _i32p), _ptr(sb, _i32p),
                                    len(qb), ctypes.c_float(radius), ctypes.byref(out))
    res = np.ctypeslib.as_array(out, shape=(len(q), width)).copy() if width > 0 else np.zeros((len(q), 0), np.int32)
    lib.ref_free(out)
    return res


def ref_batch_grid_subsampling(points, batches, features=None, sampleDl=0.1, max_p=0):
    lib = ref_lib()
    p, b = _c(points, np.float32), _c(batches, np.int32)
    f = None if features is None else _c(features, np.float32)
    fdim = 0 if f is None else f.shape[1]
    op, of = _f32p(), _f32p()
    ob = np.zeros(len(b), dtype=np.int32)
    lib.ref_batch_grid_subsampling.restype = ctypes.c_int
    m = lib.ref_batch_grid_subsampling(_ptr(p, _f32p), len(p), None if f is None else _ptr(f, _f32p), fdim,
                                       _ptr(b, _i32p), len(b), ctypes.c_float(sampleDl), int(max_p),
                                       ctypes.byref(op), ctypes.byref(of) if f is not None else None,
                                       _ptr(ob, _i32p))
    pts = np.ctypeslib.as_array(op, shape=(m, 3)).copy()
    lib.ref_free(op)
    res = [pts, ob]
    if f is not None:
        res.append(np.ctypeslib.as_array(of, shape=(m, fdim)).copy())
        lib.ref_free(of)
    return tuple(res)


def neighbor_d2(queries, supports, nbr):
    """float32 d2 of every listed neighbour with the reference's operation order ((dx*dx + dy*dy) + dz*dz);
    shadow entries (index == len(supports)) get +inf."""
    q = _c(queries, np.float32)
    s = np.concatenate([_c(supports, np.float32), np.full((1, 3), np.inf, dtype=np.float32)])
    d = q[:, None, :] - s[nbr]
    with np.errstate(invalid="ignore"):
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return np.where(nbr == len(s) - 1, np.float32(np.inf), d2.astype(np.float32))


def same_up_to_ties(got, ref, queries, supports):
    """True when two neighbour matrices are identical except for the relative order of EXACTLY equidistant
    neighbours (the reference sorts with std::sort on d2 alone, nanoflann.hpp:1286-1287: the order inside a tie
    is unspecified there — e.g. the two points of a 2-point cell are equidistant from their barycentre)."""
    if got.shape != ref.shape:
        return False
    if np.array_equal(got, ref):
        return True
    dg, dr = neighbor_d2(queries, supports, got), neighbor_d2(queries, supports, ref)
    if not np.array_equal(dg, dr):  # same multiset of distances in the same sorted positions
        return False
    rows = np.nonzero((got != ref).any(1))[0]
    for i in rows:
        if sorted(got[i].tolist()) != sorted(ref[i].tolist()):
            return False
        diff = np.nonzero(got[i] != ref[i])[0]
        for j in diff:  # every disagreeing slot must sit inside a run of equal d2
            same = dr[i] == dr[i, j]
            if same.sum() < 2:
                return False
    return True
