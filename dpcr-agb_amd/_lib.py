"""ctypes binding of libagbhip.so — the C-ABI drop-in library (include/agb_hip.h).

There is deliberately NO fallback: if the library is missing, or a tensor that is not resident on a
HIP device is handed to a kernel wrapper, the call raises.  The oracle under ``oracle/`` is test
infrastructure only and is never imported from here.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGB_LIBRARY: another build of the same library (tools/wave_timeline.py loads the instrumented `make timeline` build)
LIB_PATH = os.environ.get("AGB_LIBRARY") or os.path.join(_HERE, "libagbhip.so")

c_int = ctypes.c_int
c_ll = ctypes.c_longlong
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p

# name -> argtypes (all functions return int except where noted in _RESTYPES)
_SIGNATURES = {
    "agb_hash_capacity": [c_int],
    "agb_scan_scratch_elems": [c_int],
    "agb_hash_clear": [c_void_p, c_void_p, c_int, c_void_p],
    "agb_coords_insert": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "agb_coords_stride": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "agb_kernel_map": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                       c_ll, c_void_p, c_void_p],
    "agb_coords_bbox": [c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "agb_grid_insert": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "agb_grid_stride": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                        c_void_p, c_void_p, c_void_p, c_void_p],
    "agb_grid_kernel_map": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_ll,
                            c_void_p, c_void_p],
    "agb_batch_ptr": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "agb_spconv_fwd": [c_void_p, c_int, c_void_p, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                       c_int, c_int, c_void_p],
    "agb_spconv_fwd_ex": [c_void_p, c_int, c_void_p, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                          c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "agb_spconv_split_hint": [c_int, c_int, c_int, c_int],
    "agb_spconv_cmp_occupancy": [c_int],
    "agb_spconv_weight_transpose": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "agb_spconv_weight_transpose_z": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "agb_parity_partition": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "agb_spconv_bwd_weight": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_ll, c_void_p, c_int, c_int, c_int, c_int,
                              c_void_p],
    "agb_spconv_fwd3_grid": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                             c_int, c_int, c_void_p, c_ll, c_void_p],
    "agb_spconv_fwd3_grid_lp": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                c_int, c_int, c_void_p, c_ll, c_int, c_void_p],
    "agb_maxpool_fwd": [c_void_p, c_int, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "agb_maxpool_bwd": [c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "agb_maxpool_fwd_k": [c_void_p, c_int, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "agb_maxpool_bwd_k": [c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "agb_segment_reduce": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p],
    "agb_segment_broadcast": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                              c_void_p],
    "agb_segment_scale_add": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                              c_void_p],
    "agb_segment_max_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "agb_adabelief_chunk": [],
    "agb_adabelief_step": [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_float,
                           c_float, c_float, c_int, c_float, c_void_p],
}

# entry points over activation / gradient ROW MATRICES exist twice: fp32 rows (agb_xxx) and bf16 rows (agb_xxx_h), same
# argument lists (csrc/norm_rows.inc, csrc/pool_rows.inc)
for _n in ("agb_maxpool_fwd", "agb_maxpool_bwd", "agb_maxpool_fwd_k", "agb_maxpool_bwd_k", "agb_segment_reduce",
           "agb_segment_broadcast", "agb_segment_scale_add", "agb_segment_max_bwd"):
    _SIGNATURES[_n + "_h"] = _SIGNATURES[_n]

_lib = None


class AgbError(RuntimeError):
    pass


def load():
    """Load libagbhip.so (once). Raises if it has not been built: run ``python __graft_entry__.py``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AgbError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.agb_last_error.restype = ctypes.c_char_p
    lib.agb_last_error.argtypes = []
    lib.agb_last_kernel.restype = ctypes.c_char_p
    lib.agb_last_kernel.argtypes = []
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue  # optional symbols are checked by tests/test_abi.py against include/agb_hip.h
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


def declare(name, argtypes, rows=False):
    """Register one more entry point (used by the modules that bind the later-added kernels).  rows: the entry point has a
    bf16-row form ``name + "_h"`` with the same argument list."""
    for nm in ((name, name + "_h") if rows else (name,)):
        _SIGNATURES[nm] = argtypes
        if _lib is not None:
            fn = getattr(_lib, nm)
            fn.argtypes = argtypes
            fn.restype = c_int


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Refuses host tensors: the product path is HIP-only."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AgbError("libagbhip kernels need tensors resident on a HIP device (got a CPU tensor); "
                       "there is no CPU fallback in the product path")
    if t.dtype == torch.bfloat16:
        raise AgbError("this entry point takes fp32 rows (got a bf16 row matrix: the bf16-activation mode reaches only the "
                       "entry points with a _h form; see _lib.rows / _lib.sfx)")
    return t.data_ptr()


def ptr16(t):
    """Device pointer of a bf16 row matrix (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.bfloat16:
        raise AgbError("expected a device-resident bf16 row matrix")
    return t.data_ptr()


def rows(t):
    """Device pointer of an activation / gradient row matrix of either storage type (fp32 or bf16); the caller picks the
    entry point with ``sfx``."""
    if t is None:
        return None
    return ptr16(t) if t.dtype == torch.bfloat16 else ptr(t)


def sfx(*ts):
    """"" when the row matrices are fp32, "_h" when they are bf16 (csrc/*_rows.inc); mixed storage is an error."""
    kinds = {t.dtype for t in ts if t is not None}
    if kinds == {torch.bfloat16}:
        return "_h"
    if kinds <= {torch.float32}:
        return ""
    raise AgbError(f"row matrices of mixed storage types in one call: {sorted(str(k) for k in kinds)}")


# torch.cuda.current_stream() builds a Stream object through four Python layers (~9 us): at ~170 launches per training
# step and direction that was 0.8 ms of host time per step.  The raw handle of the current stream of the current device is
# one C call.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


_FN = {}     # entry point name -> ctypes function (one dictionary lookup per launch instead of load() + getattr)
# An input pipeline that reads counts back asynchronously (instance/kpconv.py: the wait-free input pyramid) registers a
# callable here: it is given a turn every POLL_EVERY library calls — i.e. every ~0.5 ms while a step is being enqueued —
# to pick up the copies that have landed and enqueue its next stage.  None: nothing in flight.
POLL_HOOK = None
POLL_EVERY = 16
_poll_count = [0]
CALL_NOTE = None     # (instrumented runs: what a wrapper knows about the call it is about to make — tools/bench_config.py)


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    if POLL_HOOK is not None:
        _poll_count[0] += 1
        if _poll_count[0] >= POLL_EVERY:
            _poll_count[0] = 0
            POLL_HOOK()
    rc = fn(*args)
    if rc != 0:
        raise AgbError(f"{name} failed ({rc}): {load().agb_last_error().decode()}")
    return rc


def last_kernel():
    """Name of the compute kernel the last convolution / dense-product / weight-gradient call of this thread launched."""
    return load().agb_last_kernel().decode()


def size_call(name, *args):
    """Host helper that returns a byte count (size_t), e.g. the *_workspace_bytes functions."""
    fn = getattr(load(), name)
    fn.restype = ctypes.c_size_t
    return int(fn(*args))


def hash_capacity(n):
    return load().agb_hash_capacity(int(n))


def scan_scratch_elems(n):
    return load().agb_scan_scratch_elems(int(n))
