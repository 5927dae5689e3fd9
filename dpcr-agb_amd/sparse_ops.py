"""autograd bindings of the sparse-voxel HIP kernels (csrc/spconv.hip, csrc/pool.hip).

Each Function only marshals pointers/sizes into the C ABI on torch's current stream; all arithmetic of the
sparse path happens in libagbhip.so.  Host tensors are refused (see _lib.ptr).
"""
import contextlib
import ctypes
import os
import threading
import weakref

import torch
import torch.nn.functional as F

from . import _lib

_P = _lib.ptr
_P16 = _lib.ptr16

class KernelOptions:
    """Operand precision and kernel-choice knobs of ONE model (or one call) — carried by the model / its modules, not by
    the process: two models of different precision coexist.  The library itself keeps no state: every value travels as a
    per-call argument of the C ABI.

      precision            operand precision of the sparse / 1x1 convolution MFMAs (forward, data and weight gradient;
                           accumulation, I/O, the 3-channel stem in bf16x3, BatchNorm, SE and the index kernels stay fp32):
                             "fp32"   exact fp32 MFMA (default; bench.py's headline)
                             "bf16"   bf16 operands, fp32 accumulate (BASELINE.json config 5)
                             "bf16x3" split-bf16 (hi+lo), three MFMAs per product: fp32-level accuracy
      cmp_mode             fp32 forward / data-gradient kernel: 1 = automatic (pair-compacted kernel for many-row layers),
                           0 = never, 64 / 128 = always with that tile height (tests, tuning), 129 = 128-row tiles on the
                           C++ twin of the hand-scheduled kernel (bit-identical sums: tests)
      cmp_interleave       log2 of the row-block size of its interleaved tiles (-1: by the level's size, 0: contiguous)
      balanced_tiles       work-balanced tile tables for that kernel (csrc/tiles.hip)
      bn_stats_in_epilogue forward dense products leave BatchNorm statistics partials (agb_dense_fwd_bn)
      fused_tail           SE / bottleneck block tails as one autograd node (se_ops, backbones)
      deterministic_wgrad  weight gradients summed over row chunks in a fixed order through a workspace — in EVERY operand
                           precision and for every shape that reaches sparse_ops.weight_grad_raw (fp32: csrc/dwreg.hip and the
                           stem's grouped sub-chunks, the HBM-bound dense shapes included; bf16 / bf16x3: partial tiles of
                           k_spconv_dw_cmp folded in ascending order): bitwise reproducible training, 3.7 % slower MSENet14
                           fp32 step than the default (LDS-staged kernels, fp32 atomic accumulation)
      dw_variant           0 = automatic, 1 = LDS-staged weight-gradient kernel, 2 = register-operand kernel, 3 = persistent
                           accumulators (csrc/dwa.hip; opt-in: equal to the staged kernel inside the step), 4 = LDS-staged with
                           the 2048-row chunks of rounds 2-4 (A/B measurements)
      bf16_storage         precision "bf16": convolutions read bf16 twins of their inputs (rows and weights converted once,
                           gathered as 2-byte channels) instead of converting fp32 rows while staging them
      bf16_activations     precision "bf16" on the sparse backbones: every activation / gradient ROW MATRIX of the network is
                           stored in bf16 (the stem convolution writes bf16 rows, every later kernel follows the storage type
                           of its input: csrc/*_rows.inc, agb_spconv_fwd_h); accumulators, BatchNorm / SE statistics,
                           per-plot matrices, parameters and their gradients stay fp32.  Halves the HBM traffic of the
                           element-wise passes that dominate MSENet50 (BASELINE config 5).  Default off.
      closed_form_bias_grad  the bias gradient of a convolution that feeds a BatchNorm is taken from the BatchNorm's backward in
                           closed form (exactly 0 under batch statistics: INTEGRATION.md section A.1) instead of summing the
                           incoming gradient's columns — the one deliberate difference of the training trajectory to the
                           reference's (which sees fp32 rounding noise around that zero).  Default on; off = the column sums
                           (a paired 13-seed run with it off).
      fused_blocks         the fp32 SENet / stem blocks run as ONE library call per block and direction (fused_blocks.py,
                           csrc/net.hip: same kernels, same order, bit-identical; a fifth of the host work).  Default on;
                           off = every operator driven from Python (what the bf16 modes and non-standard blocks take anyway).
      fused_head           the regression head + loss of MinkowskiBaselineModel as one launch per direction (head_ops.py,
                           csrc/head.hip) instead of ~27 small library kernels; same arithmetic, its own summation order
                           (1e-6 of the library's).  Default on.
      fused_kpconv         rigid KPConv layers on one point set (16 / 32 channels, ragged symmetric neighbour rows) as ONE
                           kernel per direction (kpconv_ops.KPConvFusedFunction, csrc/kpfused.hip): the weighted neighbourhood
                           features never go through HBM; fixed summation order.  Default on (fp32 operands only).
      join_dgrad           a KPConv bottleneck block's input feeds its first Linear AND its shortcut: the shortcut's gradient
                           rides in the final store of that Linear's data gradient (DenseLinearFunction join form,
                           agb_spconv_bwd_data's addend) instead of a separate addition pass per block.  Default on (fp32).

    Use: ``model.kernel_options = KernelOptions(precision="bf16")`` (the backbones run their forward pass inside it), or
    ``with KernelOptions(cmp_mode=128): ...`` around direct calls.  Autograd nodes keep the options they were created
    under for their backward pass.  ``DEFAULTS`` (environment-initialised) applies where nothing else is set."""
    __slots__ = ("precision", "cmp_mode", "cmp_interleave", "balanced_tiles", "bn_stats_in_epilogue", "fused_tail",
                 "deterministic_wgrad", "dw_variant", "bf16_storage", "bf16_activations", "closed_form_bias_grad",
                 "fused_blocks", "fused_head", "fused_kpconv", "join_dgrad")
    PRECISIONS = ("fp32", "bf16", "bf16x3")

    def __init__(self, precision=None, cmp_mode=None, cmp_interleave=None, balanced_tiles=None,
                 bn_stats_in_epilogue=None, fused_tail=None, deterministic_wgrad=None, dw_variant=None,
                 bf16_storage=None, bf16_activations=None, closed_form_bias_grad=None, fused_blocks=None, fused_head=None,
                 fused_kpconv=None, join_dgrad=None, base=None):
        base = base if base is not None else (current() if "DEFAULTS" in globals() else None)
        pick = lambda v, name, dflt: v if v is not None else (getattr(base, name) if base is not None else dflt)  # noqa: E731
        self.precision = pick(precision, "precision", "fp32")
        if self.precision not in self.PRECISIONS:
            raise ValueError(f"conv precision '{self.precision}': choose fp32, bf16 or bf16x3")
        self.cmp_mode = int(pick(cmp_mode, "cmp_mode", 1))
        self.cmp_interleave = int(pick(cmp_interleave, "cmp_interleave", -1))
        self.balanced_tiles = bool(pick(balanced_tiles, "balanced_tiles", True))
        self.bn_stats_in_epilogue = bool(pick(bn_stats_in_epilogue, "bn_stats_in_epilogue", True))
        self.fused_tail = bool(pick(fused_tail, "fused_tail", True))
        self.deterministic_wgrad = bool(pick(deterministic_wgrad, "deterministic_wgrad", False))
        self.dw_variant = int(pick(dw_variant, "dw_variant", 0))
        self.bf16_storage = bool(pick(bf16_storage, "bf16_storage", True))
        self.bf16_activations = bool(pick(bf16_activations, "bf16_activations", False))
        self.closed_form_bias_grad = bool(pick(closed_form_bias_grad, "closed_form_bias_grad", True))
        self.fused_blocks = bool(pick(fused_blocks, "fused_blocks", True))
        self.fused_head = bool(pick(fused_head, "fused_head", True))
        self.fused_kpconv = bool(pick(fused_kpconv, "fused_kpconv", True))
        self.join_dgrad = bool(pick(join_dgrad, "join_dgrad", True))

    def replace(self, **kw):
        return KernelOptions(base=self, **kw)

    @property
    def prec_id(self):
        return _PREC_ID.get(self.precision, 0)

    @property
    def low_precision(self):
        return self.precision in _PREC_ID

    @property
    def rows_bf16(self):
        """True when the backbone's row matrices are to be stored in bf16 (bf16 operands AND bf16_activations)."""
        return self.bf16_activations and self.precision == "bf16"

    def __enter__(self):
        _scope().append(self)
        return self

    def __exit__(self, *exc):
        _scope().pop()
        return False

    def __repr__(self):
        return "KernelOptions(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in self.__slots__) + ")"


_PREC_ID = {"bf16": 1, "bf16x3": 2}
_TLS = threading.local()


def _scope():
    st = getattr(_TLS, "stack", None)
    if st is None:
        st = _TLS.stack = []
    return st


def current():
    """The options in force here: the innermost ``with KernelOptions(...)`` / model scope of this thread, else DEFAULTS."""
    st = getattr(_TLS, "stack", None)
    return st[-1] if st else DEFAULTS


DEFAULTS = KernelOptions(precision=os.environ.get("AGB_CONV_PRECISION", "fp32"), cmp_mode=1, cmp_interleave=-1,
                         balanced_tiles=os.environ.get("AGB_BALANCED_TILES", "1") != "0",
                         bn_stats_in_epilogue=os.environ.get("AGB_BN_EPILOGUE", "1") != "0", fused_tail=True,
                         deterministic_wgrad=os.environ.get("AGB_DETERMINISTIC_WGRAD", "0") != "0",
                         dw_variant=int(os.environ.get("AGB_DW_VARIANT", "0")),
                         bf16_storage=os.environ.get("AGB_BF16_STORAGE", "1") != "0",
                         fused_blocks=os.environ.get("AGB_FUSED_BLOCKS", "1") != "0",
                         fused_head=os.environ.get("AGB_FUSED_HEAD", "1") != "0",
                         fused_kpconv=os.environ.get("AGB_FUSED_KPCONV", "1") != "0",
                         join_dgrad=os.environ.get("AGB_JOIN_DGRAD", "1") != "0")


def set_conv_precision(name):
    """Process DEFAULT operand precision (command-line tools; a model's own ``kernel_options`` wins).  Returns the previous
    default."""
    if name not in KernelOptions.PRECISIONS:
        raise ValueError(f"conv precision '{name}': choose fp32, bf16 or bf16x3")
    old, DEFAULTS.precision = DEFAULTS.precision, name
    return old


def model_scope(module):
    """Context of a model's forward pass: its own ``kernel_options`` when it carries some, else whatever is in force."""
    opts = getattr(module, "kernel_options", None)
    return opts if opts is not None else contextlib.nullcontext()


_lib.declare("agb_spconv_fwd_opt", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_ll, _lib.c_int,
                                    _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                    _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int,
                                    _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_void_p])
_lib.declare("agb_spconv_bwd_data", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_ll, _lib.c_int,
                                     _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p,
                                     _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                     _lib.c_int, _lib.c_void_p])
_lib.declare("agb_spconv_bwd_data_h", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_ll, _lib.c_int,
                                       _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p,
                                       _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                       _lib.c_int, _lib.c_void_p])
_lib.declare("agb_spconv_split_hint_opt", [_lib.c_int] * 5)
_lib.declare("agb_spconv_bwd_weight_persistent", [_lib.c_int] * 6)
_lib.declare("agb_spconv_cmp_geometry", [_lib.c_int] * 8 + [_lib.c_void_p])
_lib.declare("agb_spconv_balance_tiles_workspace_bytes", [_lib.c_int] * 3)
_lib.declare("agb_spconv_balance_tiles", [_lib.c_void_p, _lib.c_ll, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                          _lib.c_void_p, _lib.c_void_p, _lib.c_void_p])
_lib.declare("agb_spconv_fwd_tiles", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_ll, _lib.c_int,
                                      _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                      _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_void_p,
                                      _lib.c_int, _lib.c_int, _lib.c_void_p])
# Work-balanced tiles for the pair-compacted kernel (csrc/tiles.hip): one table per (kernel map, tile geometry), kept on
# the map tensor (`nbr.agb_tiles`), built on first use — on the side stream when the input pipeline prebuilds the maps.
_GEO = (ctypes.c_int32 * 4)()


def cmp_tile_table(nbr, n_out, K3, cin, cout, ldx, ldy, opts=None):
    """Tile table of the fp32 product on kernel map `nbr` for this shape (None: the call takes another kernel or
    contiguous tiles).  Same sums with or without it; it only evens out the work per tile."""
    opts = opts or current()
    if not opts.balanced_tiles or n_out <= 0:
        return None
    split = _lib.load().agb_spconv_split_hint_opt(n_out, K3, cin, cout, opts.cmp_mode)
    _lib.call("agb_spconv_cmp_geometry", n_out, cin, cout, ldx, ldy, split, opts.cmp_mode, opts.cmp_interleave, _GEO)
    rpt, ntiles, il = _GEO[1], _GEO[2], _GEO[3]
    if _GEO[0] == 0 or il == 0 or (K3 << il) + 1 > 1024:
        return None     # another kernel, contiguous tiles, or more pairs per row block than the counting sort has bins
    cache = getattr(nbr, "agb_tiles", None)
    if cache is None:
        cache = nbr.agb_tiles = {}
    tab = cache.get((rpt, ntiles, il))
    if tab is None:
        tab = torch.empty(ntiles, rpt >> il, dtype=torch.int32, device=nbr.device)
        ws = torch.empty(_lib.size_call("agb_spconv_balance_tiles_workspace_bytes", n_out, K3, il), dtype=torch.uint8,
                         device=nbr.device)
        _lib.call("agb_spconv_balance_tiles", _P(nbr), nbr.stride(0), n_out, K3, il, ntiles, rpt >> il, _P(tab), _P(ws),
                  _lib.stream())
        cache[(rpt, ntiles, il)] = tab
    return tab


_lib.declare("agb_dense_split_hint", [_lib.c_int] * 3)
_lib.declare("agb_dense_bn_chunks", [_lib.c_int] * 3)
_lib.declare("agb_dense_fwd_bn", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int,
                                  _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p])
_lib.declare("agb_spconv_bwd_weight_lp", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_ll,
                                          _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                          _lib.c_void_p])
_lib.declare("agb_spconv_bwd_weight_workspace_bytes", [_lib.c_int] * 6)
_lib.declare("agb_spconv_bwd_weight_ws", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_ll,
                                          _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                          _lib.c_int, _lib.c_void_p, ctypes.c_size_t, _lib.c_void_p])
_lib.declare("agb_spconv_fwd_lp", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_ll, _lib.c_int,
                                   _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                   _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int,
                                   _lib.c_void_p, _lib.c_int, _lib.c_void_p])

# When set to a list, every sparse-conv launch appends
#   dict(kind, K3, cin, cout, rows, pairs (device int64 tensor or None), start, end (torch.cuda.Event))
# bench.py uses this to time the dominant kernel with HIP events on the launch stream.  PROFILE_FILTER (a predicate over
# the same fields, events excluded) restricts the instrumentation: an event pair costs ~6 us of queue bubbles around
# the launch it brackets, so the timed region of the benchmark only brackets the kernel the roofline is quoted on.
PROFILE = None
PROFILE_FILTER = None


def _prof_begin(kind=None, K3=0, cin=0, cout=0, rows=0, perm=False, split=1, rows_in=None):
    if PROFILE is None:
        return None
    if PROFILE_FILTER is not None and not PROFILE_FILTER(dict(kind=kind, K3=K3, cin=cin, cout=cout, rows=rows, perm=perm,
                                                              split=split, rows_in=rows_in)):
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _prof_end(ev0, kind, K3, cin, cout, rows, pairs, perm=False, split=1, rows_in=None):
    if ev0 is None:
        return
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record()
    PROFILE.append(dict(kind=kind, K3=K3, cin=cin, cout=cout, rows=rows, pairs=pairs, start=ev0, end=ev1, perm=perm,
                        split=split, rows_in=rows_in, kernel=_lib.last_kernel()))


class _ZeroArena:
    """Parameter-gradient buffers that must START AT ZERO (weight gradients are accumulated into; the bias gradient of a
    convolution in front of a training-mode BatchNorm IS zero) as slices of ONE tensor cleared by one fill per backward pass
    instead of one fill per buffer (MSENet50: 115 fills of 3.7 us per step).  ``new_step`` (InstanceBase.optimize_parameters,
    before ``loss.backward()``) sizes the tensor from what the previous pass asked for; without it — or beyond it — a request
    is a plain ``torch.zeros``.  A slice is handed out once: nothing is ever cleared twice or shared."""

    def __init__(self):
        self.buf, self.off, self.need = None, 0, 0

    def new_step(self, device):
        cap, self.need, self.off = self.need, 0, 0
        self.buf = torch.zeros(cap, dtype=torch.float32, device=device) if cap > 0 else None

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64
        self.need += n_al
        buf = self.buf
        if buf is None or buf.device != device or self.off + n_al > buf.numel():
            return torch.zeros(shape, dtype=torch.float32, device=device)
        v = buf[self.off:self.off + n].view(shape)
        self.off += n_al
        return v


ZERO_ARENA = _ZeroArena()


def zeros_f32(shape, device):
    """A zero-filled fp32 gradient buffer (see _ZeroArena)."""
    return ZERO_ARENA.take(tuple(shape) if not isinstance(shape, int) else (shape,), device)


def _colsum_hint(dy):
    """Column sums of dy left by the BatchNorm backward that produced it (norm_ops.py), or None.  Valid only for the
    very tensor the BatchNorm node returned: an in-place accumulation of another consumer's gradient bumps the
    version counter (and an out-of-place one creates a new tensor without the attribute)."""
    hint = getattr(dy, "agb_colsum", None)
    if hint is None:
        return None
    colsum, version = hint
    if dy._version != version:
        return None
    if isinstance(colsum, int):       # training-mode BatchNorm: the sums are zero; the buffer is only made when someone asks
        return zeros_f32((colsum,), dy.device)
    return colsum


def _pad_cols(x, mult):
    c = x.shape[-1]
    pad = (-c) % mult
    if pad == 0:
        return x.contiguous()
    return F.pad(x, (0, pad)).contiguous()


def _small_cin_pad(cin):
    """Channel count the kernels want: 4 or 8 for the stem-like small-Cin path, else a multiple of 4."""
    if cin <= 4:
        return 4
    if cin <= 8:
        return 8
    return (cin + 3) // 4 * 4


# Partial BatchNorm statistics from the epilogue of the forward dense products (agb_dense_fwd_bn); the product's caller
# moves them onto the tensor it returns (take_bn_hint), norm_ops.batch_norm_act picks them up from there.  The hand-off
# slot is per thread, holds the output only weakly, and is emptied at the start of every product.
def _set_last_bn_part(value):
    _TLS.last_bn_part = value


def take_bn_hint(out):
    """Attach the statistics partials of the dense product that just produced `out` (if it left any) to `out`."""
    last = getattr(_TLS, "last_bn_part", None)
    _TLS.last_bn_part = None
    if last is not None and isinstance(out, torch.Tensor):
        part, chunks, yref = last
        y = yref()
        if y is not None and out.data_ptr() == y.data_ptr() and out.shape == y.shape and out.is_contiguous():
            out.agb_bn_part = (part, chunks, out._version)
    return out


def bn_hint(x, c):
    """(part, chunks) of a still-valid statistics hint on x, or None."""
    h = getattr(x, "agb_bn_part", None)
    if h is None or h[2] != x._version or h[0].numel() != h[1] * 3 * c:
        return None
    return h[0], h[1]


_lib.declare("agb_to_bf16", [_lib.c_void_p, _lib.c_ll, _lib.c_ll, _lib.c_int, _lib.c_void_p, _lib.c_ll, _lib.c_void_p])
_lib.declare("agb_spconv_bwd_weight_b16", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_ll,
                                           _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p])
_lib.declare("agb_spconv_bwd_weight_b16_ws", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_ll,
                                              _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p,
                                              ctypes.c_size_t, _lib.c_void_p])
_lib.declare("agb_spconv_fwd_b16", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_ll, _lib.c_int,
                                    _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                    _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int,
                                    _lib.c_void_p, _lib.c_void_p])
_lib.declare("agb_spconv_fwd_h", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_ll, _lib.c_int,
                                    _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int,
                                    _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int,
                                    _lib.c_void_p, _lib.c_void_p])
_lib.declare("agb_spconv_fwd3_grid_h", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p,
                                        _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int,
                                        _lib.c_void_p, _lib.c_ll, _lib.c_void_p])


_lib.declare("agb_stem_fwd_pairs", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p,
                                    _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int,
                                    _lib.c_void_p, _lib.c_ll, _lib.c_void_p])
_V_, _I_ = _lib.c_void_p, _lib.c_int
_lib.declare("agb_spconv_fwd3_grid_dense", [_V_, _I_, _V_, _V_, _V_, _V_, _I_, _V_, _V_, _I_, _I_, _I_, _V_, _lib.c_ll, _V_])
_lib.declare("agb_stem_bwd_weight_grid_workspace_bytes", [_I_, _I_])
_lib.declare("agb_stem_bwd_weight_grid", [_V_, _I_, _V_, _I_, _V_, _V_, _V_, _I_, _V_, _I_, _I_, _V_, ctypes.c_size_t, _V_])
_lib.declare("agb_weight_twins_bf16", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                       _lib.c_void_p])


_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """Parameters were rewritten through raw device pointers (the fused optimiser step: torch's version counters do not
    see it): cached bf16 forms of the weights are stale."""
    _WEIGHT_EPOCH[0] += 1


_lib.declare("agb_weight_twins_batched", [_lib.c_void_p, _lib.c_int, _lib.c_ll, _lib.c_void_p])


class _TwinRegistry:
    """The bf16 operand forms (W16 [K3, cin, cout], Wt16 [K3, cout, cin]) of every convolution kernel that ever asked for them,
    in persistent buffers refreshed by ONE launch (agb_weight_twins_batched) the first time a step asks after the weights
    changed — 52 launches of 6 us per MSENet50 step before."""

    def __init__(self):
        self.entries = []          # [weakref(kernel), w16, wt16, (K3, cin, cout)]
        self.table = None
        self.table_ptrs = None
        self.tiles = 0

    def register(self, kernel):
        w = kernel.detach()
        K3, cin, cout = (1,) + tuple(w.shape) if w.dim() == 2 else tuple(w.shape)
        w16 = torch.empty(K3, cin, cout, dtype=torch.bfloat16, device=w.device)
        wt16 = torch.empty(K3, cout, cin, dtype=torch.bfloat16, device=w.device)
        self.entries.append([weakref.ref(kernel), w16, wt16, (K3, cin, cout)])
        return w16, wt16

    def refresh(self, device):
        """Both forms of every live registered kernel on `device`, current weights, one launch; marks them fresh."""
        live, rows, ptrs, tiles = [], [], [], 0
        for e in self.entries:
            k = e[0]()
            if k is None:
                continue
            live.append(e)
            if k.device != device or not k.is_contiguous():
                continue
            K3, cin, cout = e[3]
            rows.append([k.data_ptr(), e[1].data_ptr(), e[2].data_ptr(), K3, cin, cout, tiles])
            # (source AND destinations: a freed kernel's address is reused by the next model's kernel of the same size)
            ptrs.append((k.data_ptr(), e[1].data_ptr(), e[2].data_ptr(), K3, cin, cout))
            tiles += K3 * ((cin + 63) // 64) * ((cout + 63) // 64)
        self.entries = live
        if not rows:
            return
        ptrs = tuple(ptrs)
        if ptrs != self.table_ptrs:
            self.table = torch.tensor(rows, dtype=torch.int64).to(device)
            self.table_ptrs, self.tiles = ptrs, tiles
        _lib.call("agb_weight_twins_batched", self.table.data_ptr(), len(rows), self.tiles, _lib.stream())
        epoch = _WEIGHT_EPOCH[0]
        for e in live:
            k = e[0]()
            if k is not None and k.device == device and k.is_contiguous():
                k.agb_twins = ((k._version, epoch), e[1], e[2])


_TWINS = _TwinRegistry()


def weight_twins(kernel):
    """(W16 [K3, cin, cout], Wt16 [K3, cout, cin]) bf16 operand forms of a convolution kernel [K3, cin, cout] (or [cin, cout]),
    cached on the parameter until its next update (version counter for torch's in-place writes, the weight epoch for the fused
    optimiser): the forward pass takes Wt16, the data gradient W16.  A stale kernel refreshes EVERY registered kernel of its
    device in one launch (the optimiser rewrote them all)."""
    key = (kernel._version, _WEIGHT_EPOCH[0])
    h = getattr(kernel, "agb_twins", None)
    if h is not None and h[0] == key:
        return h[1], h[2]
    if not kernel.is_contiguous():      # (not a case of the models here: the one-layer launch on a contiguous copy)
        w = kernel.detach().contiguous()
        K3, cin, cout = (1,) + tuple(w.shape) if w.dim() == 2 else tuple(w.shape)
        w16 = torch.empty(K3, cin, cout, dtype=torch.bfloat16, device=w.device)
        wt16 = torch.empty(K3, cout, cin, dtype=torch.bfloat16, device=w.device)
        _lib.call("agb_weight_twins_bf16", _P(w), K3, cin, cout, _P16(w16), _P16(wt16), _lib.stream())
        kernel.agb_twins = (key, w16, wt16)
        return w16, wt16
    if h is None:
        _TWINS.register(kernel)
    _TWINS.refresh(kernel.device)
    h = kernel.agb_twins
    return h[1], h[2]


def has_twin(t):
    if t.dtype == torch.bfloat16:
        return True
    h = getattr(t, "agb_bf16", None)
    return h is not None and h[1] == t._version and h[0].shape == t.shape


def bf16_twin(t, cache=True):
    """bf16 copy (round to nearest even, csrc k_to_bf16) of a 2-D fp32 row matrix.  cache: keep it on the tensor (valid while
    the tensor's version stands) — an activation that feeds several convolutions, forward and weight gradient, is
    converted once."""
    if t.dtype == torch.bfloat16:      # bf16 storage: the row matrix is its own twin
        return t
    if cache:
        h = getattr(t, "agb_bf16", None)
        if h is not None and h[1] == t._version and h[0].shape == t.shape:
            return h[0]
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.AgbError("bf16_twin takes a 2-D row matrix with contiguous rows")
    n, c = t.shape
    y = torch.empty(n, c, dtype=torch.bfloat16, device=t.device)
    _lib.call("agb_to_bf16", _P(t), t.stride(0), n, c, _P16(y), y.stride(0), _lib.stream())
    if cache:
        t.agb_bf16 = (y, t._version)
    return y


def spconv_forward_raw(x, w2d, nbr, kflip, bias, n_out, K3, cin, cout, kind="fwd", pairs=None, plan=None,
                       w_kmajor=None, bn_stats=False, opts=None, out_bf16=None, w16=None):
    """Y = bias + sum_k X[nbr[k]] @ W[k].  x: [N_in, cin] (cin % 4 == 0), w2d: [K3*cin, cout].
    plan: optional (perm, tile_cls, cls_tab, max_tiles) class partition of the output rows (strided data grad).
    w_kmajor: the same weights as [K3, cout, cin] (k contiguous), needed by the bf16 / bf16x3 operand modes.
    opts: KernelOptions of the call (default: the ones in force).
    out_bf16: storage type of the output rows (default: that of x).  bf16 rows (in or out) take the bf16 operand mode:
    w_kmajor is required — or w16, its bf16 form [K3, cout, cin] (weight_twins), which then is what the kernel reads."""
    opts = opts or current()
    _set_last_bn_part(None)
    x_bf16 = x.dtype == torch.bfloat16
    if out_bf16 is None:
        out_bf16 = x_bf16
    if x_bf16 or out_bf16:
        if (w_kmajor is None and w16 is None) or opts.prec_id != 1:
            raise _lib.AgbError("bf16 row matrices need the bf16 operand mode (KernelOptions precision='bf16') and K-major "
                                "weights")
        if cin % 8 != 0 or x.stride(0) % 8 != 0 or cout % 4 != 0:
            # widths the bf16-storage kernel does not take (none in MSENet14/50): fp32 rows through the staging kernel
            if w_kmajor is None:
                w_kmajor = w16.float()
            y = spconv_forward_raw(x.float() if x_bf16 else x, w2d, nbr, kflip, bias, n_out, K3, cin, cout, kind, pairs, plan,
                                   w_kmajor, bn_stats, opts, out_bf16=False)
            return y.to(torch.bfloat16) if out_bf16 else y
    y = torch.empty(n_out, cout, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    if n_out == 0:
        return y
    if plan is not None:
        # class-partitioned strided data gradient: few 64-row tiles with a long reduction (the 512 -> 256 layer's 14 k rows are
        # 900 workgroups of up to 8 offsets x 512 channels: 273 us) split the reduction four ways (186 us)
        split = 4 if ((n_out // 64 + 1) * ((cout + 63) // 64) < 1100 and cin >= 256 and cin % 256 == 0) else 1
    elif nbr is not None:
        split = _lib.load().agb_spconv_split_hint_opt(n_out, K3, cin, cout, opts.cmp_mode)
    else:
        split = _lib.load().agb_dense_split_hint(n_out, cin, cout)
    partial = torch.empty(split, n_out, cout, dtype=torch.float32, device=x.device) if split > 1 else None
    perm = tile_cls = cls_tab = None
    n_tiles = 0
    if plan is not None:
        perm, tile_cls, cls_tab, n_tiles = plan
    if nbr is None and pairs is None:
        pairs = int(n_out)        # dense product: one (in, out) pair per row
    ev = _prof_begin(kind, K3, cin, cout, n_out, plan is not None, split, x.shape[0])
    prec = opts.prec_id if ((w_kmajor is not None or w16 is not None) and cin >= 12) else 0
    bn_chunks = 0
    if opts.bn_stats_in_epilogue and bn_stats and nbr is None and not prec and split == 1:
        bn_chunks = _lib.load().agb_dense_bn_chunks(n_out, cin, cout)
    if bn_chunks > 0:
        # forward dense product in training: the BatchNorm that follows takes its statistics from this epilogue
        part = torch.empty(bn_chunks * 3 * cout, dtype=torch.float32, device=x.device)
        _lib.call("agb_dense_fwd_bn", _P(x), x.stride(0), _P(w2d), _P(bias), _P(y), y.stride(0), n_out, cin, cout, _P(part),
                  _lib.stream())
        _set_last_bn_part((part, bn_chunks, weakref.ref(y)))
    elif prec == 1 and (x_bf16 or out_bf16 or (opts.bf16_storage and cin % 8 == 0 and x.stride(0) % 8 == 0
                                               and (nbr is not None or has_twin(x)))):
        # (a dense product reads every row once per column tile: converting first only pays when the twin exists already;
        # a 3^3 gather re-reads every row ~15 times)
        # bf16 mode on bf16 storage: twins of the rows (cached on the tensor: a block input feeds two convolutions) and of
        # the K-major weights; the kernel gathers 2-byte channels straight into LDS
        x16 = bf16_twin(x)
        if w16 is None:
            w16 = bf16_twin(w_kmajor.view(-1, cin), cache=False)
        _lib.call("agb_spconv_fwd_h" if out_bf16 else "agb_spconv_fwd_b16", _P16(x16), x16.stride(0), _P16(w16), _P(nbr),
                  0 if nbr is None else nbr.stride(0), int(kflip), _P(bias), _lib.rows(y), y.stride(0), n_out, K3, cin, cout,
                  _P(perm), _P(tile_cls), _P(cls_tab), n_tiles, split, _P(partial), _lib.stream())
    elif prec:
        _lib.call("agb_spconv_fwd_lp", _P(x), x.stride(0), _P(w_kmajor), _P(nbr), 0 if nbr is None else nbr.stride(0), int(kflip),
                  _P(bias), _P(y), y.stride(0), n_out, K3, cin, cout, _P(perm), _P(tile_cls), _P(cls_tab), n_tiles,
                  split, _P(partial), prec, _lib.stream())
    else:
        # a table pays where it serves several launches: the stride-1 maps (every convolution of a level, forward and
        # data gradient); a strided layer's forward map serves one
        tiles = (cmp_tile_table(nbr, n_out, K3, cin, cout, x.stride(0), y.stride(0), opts)
                 if (nbr is not None and plan is None and x.shape[0] == n_out and K3 > 1) else None)
        if tiles is not None:
            _lib.call("agb_spconv_fwd_tiles", _P(x), x.stride(0), _P(w2d), _P(nbr), nbr.stride(0), int(kflip), _P(bias), _P(y),
                      y.stride(0), n_out, K3, cin, cout, split, _P(partial), opts.cmp_mode, opts.cmp_interleave, _P(tiles),
                      tiles.shape[0], tiles.shape[1], _lib.stream())
        else:
            _lib.call("agb_spconv_fwd_opt", _P(x), x.stride(0), _P(w2d), _P(nbr), 0 if nbr is None else nbr.stride(0),
                      int(kflip), _P(bias), _P(y), y.stride(0), n_out, K3, cin, cout, _P(perm), _P(tile_cls), _P(cls_tab),
                      n_tiles, split, _P(partial), opts.cmp_mode, opts.cmp_interleave, _lib.stream())
    _prof_end(ev, kind, K3, cin, cout, n_out, pairs, plan is not None, split, x.shape[0])
    return y


def weight_grad_raw(x, dy, nbr, dw, n_out, K3, cin, cout, opts):
    """dw += gathered(x)^T dy through the C ABI.  KernelOptions.deterministic_wgrad: with the workspace of the fixed-order
    two-level sum — honoured in every operand precision (fp32, bf16 on twins or bf16 rows, bf16x3) and for every shape,
    the HBM-bound dense ones included (they leave the streaming kernel and its cross-workgroup atomics)."""
    prec = opts.prec_id if cin >= 12 else 0
    rows16 = x.dtype == torch.bfloat16 or dy.dtype == torch.bfloat16
    if rows16 and prec != 1:
        # (the stem: fp32 3-channel features against a bf16 gradient; a handful of rows x 64 columns converted back)
        x, dy, rows16 = x.float(), dy.float(), False
    det = opts.deterministic_wgrad and opts.dw_variant != 1
    if prec == 1 and x.stride(0) % 4 == 0 and dy.stride(0) % 4 == 0 and (rows16 or (opts.bf16_storage and nbr is not None)):
        x16, dy16 = bf16_twin(x), bf16_twin(dy)     # (dy's twin is shared with the data gradient of the same layer)
        nbytes = _lib.size_call("agb_spconv_bwd_weight_workspace_bytes", n_out, K3, cin, cout, int(nbr is None), 1) if det else 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        _lib.call("agb_spconv_bwd_weight_b16_ws", _P16(x16), x16.stride(0), _P16(dy16), dy16.stride(0), _P(nbr),
                  0 if nbr is None else nbr.stride(0), _P(dw), n_out, K3, cin, cout, _P(ws), nbytes, _lib.stream())
        return
    if rows16:
        x, dy = x.float(), dy.float()
    # the 3-channel stem (rows 4 floats wide, 64 output channels) always brings a workspace: with one the library takes its
    # pair-sparse kernel (csrc/stem.hip: one 4x4x1 MFMA per pair, partial tiles folded in a fixed order) — faster than the
    # dense-over-offsets kernel AND reproducible
    stem = (cin == 4 and cout == 64 and nbr is not None and x.stride(0) == 4 and opts.dw_variant == 0
            and x.dtype == torch.float32 and dy.dtype == torch.float32)
    # fp32 maps with Cin, Cout multiples of 64 (every 3^3 / 2^3 layer of the SENets): with a workspace the library runs the
    # persistent-accumulator kernel (csrc/dwa.hip) — the product path, AND reproducible; KernelOptions.dw_variant 1 / 2
    # still select the older kernels for A/B measurements
    # (k_spconv_dwa addresses the rows of X with 32-bit byte offsets and the ABI carries no n_in: a level whose rows reach
    # past 4 GiB — none of the NFI shapes — keeps the staged kernel)
    fits32 = x.shape[0] * x.stride(0) * 4 < (1 << 32) and dy.shape[0] * dy.stride(0) * 4 < (1 << 32)
    persistent = (prec == 0 and nbr is not None and x.dtype == torch.float32 and dy.dtype == torch.float32 and fits32 and
                  (opts.dw_variant == 3 or (opts.dw_variant == 0 and PERSISTENT_WGRAD and _lib.load().
                   agb_spconv_bwd_weight_persistent(n_out, K3, cin, cout, x.stride(0), dy.stride(0)) == 1)))
    nbytes = _lib.size_call("agb_spconv_bwd_weight_workspace_bytes", n_out, K3, cin, cout, int(nbr is None), prec) \
        if (det or stem or persistent) else 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    variant = opts.dw_variant
    if not fits32 and nbr is not None and prec == 0 and (det or variant == 3):
        variant = 2        # (reproducible sums without the persistent kernel's 32-bit row offsets: the register-operand kernel)
    _lib.call("agb_spconv_bwd_weight_ws", _P(x), x.stride(0), _P(dy), dy.stride(0), _P(nbr), 0 if nbr is None else nbr.stride(0),
              _P(dw), n_out, K3, cin, cout, prec, variant, _P(ws), nbytes, _lib.stream())


# The persistent-accumulator weight gradient (csrc/dwa.hip) is OPT-IN: measured equal to the LDS-staged kernel inside the
# training step (EXPERIMENTS.md round 5: its kernels win 6-24 % on the shapes it takes, its 115 MB fold gives that back)
PERSISTENT_WGRAD = os.environ.get("AGB_PERSISTENT_WGRAD", "0") != "0"
_PROBE_PAIRS = {}


def _probe_pair_count(coords, grid, desc, K, n_out):
    """Profiling only: number of (row, offset) pairs of a K^3 stride-1 map on a grid-mode level (device int64 scalar), from
    a scratch kernel map (the product path no longer writes one for the 64-channel stem).  Counted once per level (forward
    and weight gradient of a step ask for the same number; the scratch map is 578 MB at B = 32)."""
    key = (coords.data_ptr(), grid.data_ptr(), K, n_out)
    hit = _PROBE_PAIRS.get(key)
    if hit is not None and hit[0]() is coords:
        return hit[1]
    nbr = torch.empty(K ** 3, max(n_out, 1), dtype=torch.int32, device=coords.device)
    y = torch.empty(n_out, 64, dtype=torch.float32, device=coords.device)
    x = torch.zeros(n_out, 4, dtype=torch.float32, device=coords.device)
    w = torch.zeros(K ** 3, 3, 64, dtype=torch.float32, device=coords.device)
    _lib.call("agb_stem_fwd_pairs", _P(x), 4, _P(w), _P(coords), _P(grid), desc, K, None, _P(y), 64, n_out, 64, _P(nbr),
              nbr.stride(0), _lib.stream())
    pairs = (nbr >= 0).sum()
    if len(_PROBE_PAIRS) > 8:
        _PROBE_PAIRS.clear()
    _PROBE_PAIRS[key] = (weakref.ref(coords), pairs)
    return pairs


def _twins_for(kernel, rows, opts, cin, cout, cin_p, cout_p):
    """weight_twins(kernel) when the call runs on bf16 rows (`rows`: its input row matrix) with unpadded widths the bf16-storage
    kernels take; else None (the per-call conversions)."""
    if (rows.dtype != torch.bfloat16 or opts.prec_id != 1 or cin_p != cin or cout_p != cout or cin % 8 != 0 or cout % 8 != 0
            or cin < 12 or cout < 12 or not kernel.is_contiguous()):
        return None
    return weight_twins(kernel)


class SparseConvFunction(torch.autograd.Function):
    """Generalized sparse convolution. kernel: [K3, Cin, Cout]; nbr: forward map [K3, N_out];
    nbrT: transposed map [K3, N_in] or None when the k-flipped forward map serves (stride 1, odd kernel)."""

    @staticmethod
    def forward(ctx, feats, kernel, bias, nbr, nbrT, n_in, n_out, plan=None, probe=None):
        K3, cin, cout = kernel.shape
        if probe is not None:
            return SparseConvFunction._forward_probe(ctx, feats, kernel, bias, n_out, probe)
        cin_p = _small_cin_pad(cin)
        cout_p = (cout + 3) // 4 * 4
        x = (feats if cin_p == cin else F.pad(feats, (0, cin_p - cin))).contiguous()
        w = kernel
        if cin_p != cin or cout_p != cout:
            w = F.pad(kernel, (0, cout_p - cout, 0, cin_p - cin))
        w2d = w.contiguous().view(K3 * cin_p, cout_p)
        b = None
        if bias is not None:
            b = bias.reshape(-1)
            if cout_p != cout:
                b = F.pad(b, (0, cout_p - cout))
            b = b.contiguous()
        pairs = getattr(nbr, "agb_pairs", None)
        opts = ctx.opts = current()
        lp = opts.low_precision and cin_p >= 12
        if cin == 3 and cout_p == cout:
            # three input channels (the stem): rows stay 4 floats wide, the weights go in unpadded — the kernel packs
            # 10 offsets x 3 channels per K-chunk instead of 8 x 4
            y = spconv_forward_raw(x, kernel.contiguous().view(K3 * 3, cout), nbr, 0, b, n_out, K3, 3, cout_p, "fwd", pairs,
                                   opts=opts)
        else:
            if x.dtype == torch.bfloat16 and not lp:     # (bf16 rows into a layer the bf16 kernels do not take)
                x = x.float()
            # bf16 rows: both bf16 operand forms of the weights come from one launch per layer and step (weight_twins);
            # else the K-major fp32 kernel [K3, cout, cin] for the staging-conversion kernels
            tw = _twins_for(kernel, x, opts, cin, cout, cin_p, cout_p) if lp else None
            wkm = w.transpose(1, 2).contiguous() if (lp and not tw) else None
            y = spconv_forward_raw(x, w2d, nbr, 0, b, n_out, K3, cin_p, cout_p, "fwd", pairs, None, wkm, opts=opts,
                                   w16=tw[1] if tw else None, out_bf16=True if (opts.rows_bf16 and lp) else None)
        if opts.rows_bf16 and y.dtype != torch.bfloat16:
            y = y.to(torch.bfloat16)          # (a narrow first layer on the fp32 kernels: the rows leave in the model's storage)
        ctx.pairs = pairs
        ctx.plan = plan
        ctx.save_for_backward(x, w, nbr, nbrT if nbrT is not None else torch.empty(0))
        ctx.dims = (K3, cin, cout, cin_p, cout_p, n_in, n_out, nbrT is not None, bias is not None,
                    None if bias is None else bias.shape)
        return y if cout_p == cout else y[:, :cout].contiguous()

    @staticmethod
    def _forward_probe(ctx, feats, kernel, bias, n_out, probe):
        """3-channel stride-1 layer: the forward kernel probes the level's dense grid for its neighbours and writes the
        kernel map out for the weight gradient (csrc/spconv.hip GridProbe)."""
        (coords, grid, desc), K = probe
        K3, cin, cout = kernel.shape
        if cin != 3 or cout % 4 != 0:
            raise _lib.AgbError("the grid-probing path takes 3 input channels and a multiple of 4 output channels")
        x = F.pad(feats, (0, 1)).contiguous()
        b = None if bias is None else bias.reshape(-1).contiguous()
        ctx.opts = current()
        rows16 = ctx.opts.rows_bf16
        y = torch.empty(n_out, cout, dtype=torch.bfloat16 if rows16 else torch.float32, device=x.device)
        need_w = ctx.needs_input_grad[1]   # (grad mode is off inside Function.forward)
        # 64 output channels: the weight gradient probes the grid itself (csrc/stem.hip, agb_stem_bwd_weight_grid) — the
        # K^3 x N kernel map (578 MB for the 7^3 stem at B = 32) is neither written here nor read there
        # (the conditions of csrc/stem.hip agb_stem_dw_ok, so that a level it would refuse still gets its map written)
        grid_wgrad = (cout == 64 and x.dtype == torch.float32 and ctx.opts.dw_variant == 0 and 0 < n_out < (1 << 24)
                      and x.stride(0) == 4 and K3 <= 729)
        nbr = (torch.empty(K3, max(n_out, 1), dtype=torch.int32, device=x.device) if (need_w and not grid_wgrad) else None)
        ev = _prof_begin("fwd", K3, 3, cout, n_out)
        if rows16:
            _lib.call("agb_spconv_fwd3_grid_h", _P(x), x.stride(0), _P(kernel.contiguous()), _P(coords), _P(grid), desc, K,
                      _P(b), _P16(y), y.stride(0), n_out, cout, _P(nbr), 0 if nbr is None else nbr.stride(0), _lib.stream())
        else:
            _lib.call("agb_spconv_fwd3_grid_lp", _P(x), x.stride(0), _P(kernel.contiguous()), _P(coords), _P(grid), desc, K,
                      _P(b), _P(y), y.stride(0), n_out, cout, _P(nbr), 0 if nbr is None else nbr.stride(0),
                      ctx.opts.prec_id, _lib.stream())
        _prof_end(ev, "fwd", K3, 3, cout, n_out, None)
        if ev is not None:   # profiling only: the kernel-map size, after the closing event
            PROFILE[-1]["pairs"] = (nbr >= 0).sum() if nbr is not None else _probe_pair_count(coords, grid, desc, K, n_out)
        ctx.pairs = None
        ctx.probe = True
        ctx.probe_args = (coords, grid, desc, K) if (need_w and grid_wgrad) else None
        ctx.save_for_backward(x, nbr if nbr is not None else torch.empty(0))
        ctx.dims = (K3, cout, n_out, bias is not None, None if bias is None else bias.shape)
        return y

    @staticmethod
    def _backward_probe(ctx, dy):
        x, nbr = ctx.saved_tensors
        K3, cout, n_out, has_bias, bias_shape = ctx.dims
        colsum = _colsum_hint(dy) if (ctx.opts.closed_form_bias_grad and has_bias) else None
        dy = dy.contiguous()
        dk = db = None
        if ctx.needs_input_grad[1]:
            probe_args = getattr(ctx, "probe_args", None)
            if nbr.numel() == 0 and probe_args is None:
                raise _lib.AgbError("the forward pass ran without gradients enabled: no kernel map was written")
            dwp = zeros_f32((K3, 4, cout), dy.device)
            ev = _prof_begin("wgrad", K3, 4, cout, n_out)
            if probe_args is not None:
                coords, grid, desc, K = probe_args
                dyf = dy if dy.dtype == torch.float32 else dy.float()     # (bf16 rows: 64 columns converted back)
                nbytes = _lib.size_call("agb_stem_bwd_weight_grid_workspace_bytes", n_out, K)
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dy.device)
                _lib.call("agb_stem_bwd_weight_grid", _P(x), x.stride(0), _P(dyf), dyf.stride(0), _P(coords), _P(grid), desc, K,
                          _P(dwp), n_out, cout, _P(ws), nbytes, _lib.stream())
            else:
                weight_grad_raw(x, dy, nbr, dwp, n_out, K3, 4, cout, ctx.opts)
            _prof_end(ev, "wgrad", K3, 4, cout, n_out, None)
            if ev is not None:
                PROFILE[-1]["pairs"] = ((nbr >= 0).sum() if probe_args is None
                                        else _probe_pair_count(probe_args[0], probe_args[1], probe_args[2], probe_args[3], n_out))
            dk = dwp[:, :3, :].contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            db = (colsum if (colsum is not None and colsum.numel() == cout)
                  else dy.sum(0, dtype=torch.float32)).reshape(bias_shape)
        return None, dk, db, None, None, None, None, None, None

    @staticmethod
    def backward(ctx, dy):
        if getattr(ctx, "probe", False):
            return SparseConvFunction._backward_probe(ctx, dy)
        x, w, nbr, nbrT = ctx.saved_tensors
        K3, cin, cout, cin_p, cout_p, n_in, n_out, has_T, has_bias, bias_shape = ctx.dims
        opts = ctx.opts
        colsum = _colsum_hint(dy) if (opts.closed_form_bias_grad and has_bias) else None
        dy = dy.contiguous()
        if cout_p != cout:
            dy = F.pad(dy, (0, cout_p - cout)).contiguous()
        x_bf16 = x.dtype == torch.bfloat16
        if dy.dtype == torch.bfloat16 and not (opts.prec_id == 1 and cout_p >= 12 and cout_p % 8 == 0):
            dy = dy.float()                     # (bf16 rows on a layer the bf16 kernels do not take)
        dx = dk = db = dwp = None
        if ctx.needs_input_grad[0]:
            # dX[q] = sum_k dY[nbrT[k][q]] @ W[k]^T : same implicit GEMM with the transposed weights
            lp = opts.low_precision and cout_p >= 12
            # the data gradient multiplies by W[k]^T: its K-major form is the kernel itself ([K3, cin, cout])
            tw = _twins_for(w, dy, opts, cin, cout, cin_p, cout_p) if lp else None
            wkm = w.contiguous() if (lp and not tw) else None
            wt2d = None
            if not lp:
                w = w.contiguous()
                wt2d = torch.empty(K3 * cout_p, cin_p, dtype=torch.float32, device=w.device)
                if ctx.needs_input_grad[1]:   # the weight-gradient buffer is cleared by the same launch
                    dwp = torch.empty(K3, cin_p, cout_p, dtype=torch.float32, device=dy.device)
                _lib.call("agb_spconv_weight_transpose_z", _P(w), _P(wt2d), _P(dwp), K3, cin_p, cout_p, _lib.stream())
            if has_T:
                plan = ctx.plan if cout_p >= 12 else None
                dxp = spconv_forward_raw(dy, wt2d, nbrT, 0, None, n_in, K3, cout_p, cin_p, "dgrad", ctx.pairs, plan,
                                         wkm, opts=opts, out_bf16=x_bf16 if dy.dtype == torch.bfloat16 else None,
                                         w16=tw[0] if tw else None)
            else:
                dxp = spconv_forward_raw(dy, wt2d, nbr, 1, None, n_in, K3, cout_p, cin_p, "dgrad", ctx.pairs, None,
                                         wkm, opts=opts, out_bf16=x_bf16 if dy.dtype == torch.bfloat16 else None,
                                         w16=tw[0] if tw else None)
            dx = dxp if cin_p == cin else dxp[:, :cin].contiguous()
            if dx.dtype != x.dtype:
                dx = dx.to(x.dtype)
        if ctx.needs_input_grad[1]:
            if dwp is None:
                dwp = zeros_f32((K3, cin_p, cout_p), dy.device)
            ev = _prof_begin("wgrad", K3, cin_p, cout_p, n_out)
            weight_grad_raw(x, dy, nbr, dwp, n_out, K3, cin_p, cout_p, opts)
            _prof_end(ev, "wgrad", K3, cin_p, cout_p, n_out, ctx.pairs)
            dk = dwp if (cin_p == cin and cout_p == cout) else dwp[:, :cin, :cout].contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            if colsum is not None and colsum.numel() == cout and cout_p == cout:
                db = colsum.reshape(bias_shape)
            else:
                db = dy[:, :cout].sum(0, dtype=torch.float32).reshape(bias_shape)
        return dx, dk, db, None, None, None, None, None, None


class DenseConvFunction(torch.autograd.Function):
    """1x1 stride-1 convolution (ME's ``use_mm`` case: kernel [Cin, Cout], the coordinate map is the identity):
    Y = X @ W + b on this library's own MFMA kernels — the register-accumulator implicit-GEMM kernels and the weight-gradient
    kernel run with the identity map (nbr == NULL), in the operand precision of the KernelOptions in force.
    Needs Cin >= 12 and Cin, Cout multiples of 4 (the caller falls back to a library matmul otherwise)."""

    @staticmethod
    def supported(cin, cout):
        # (both widths are reduction dimensions of one of the three products)
        return cin >= 12 and cout >= 12 and cin % 4 == 0 and cout % 4 == 0

    @staticmethod
    def forward(ctx, feats, kernel, bias, join=False):
        """join: also return feats itself as a second output (a residual block's shortcut branch); the gradient that
        comes back for it is the ADDEND of this layer's data gradient (KernelOptions.join_dgrad; dense_conv_join)."""
        cin, cout = kernel.shape
        ctx.join = bool(join)
        x = feats.contiguous()
        n = x.shape[0]
        w = kernel.contiguous()
        b = None if bias is None else bias.reshape(-1).contiguous()
        opts = ctx.opts = current()
        lp = opts.low_precision
        if x.dtype == torch.bfloat16 and opts.prec_id != 1:
            x = x.float()
        tw = _twins_for(kernel, x, opts, cin, cout, cin, cout) if lp else None
        wkm = w.t().contiguous() if (lp and not tw) else None          # K-major [Cout][Cin]
        # (a forward pass that will be differentiated = training: the BatchNorm behind it wants batch statistics)
        y = spconv_forward_raw(x, w, None, 0, b, n, 1, cin, cout, "fwd1x1", None, None, wkm,
                               bn_stats=any(ctx.needs_input_grad), opts=opts, w16=tw[1] if tw else None,
                               out_bf16=True if (opts.rows_bf16 and lp) else None)
        if opts.rows_bf16 and y.dtype != torch.bfloat16:
            y = y.to(torch.bfloat16)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias_shape = None if bias is None else bias.shape
        return (y, feats) if join else y

    @staticmethod
    def backward(ctx, dy, dbranch=None):
        x, w = ctx.saved_tensors
        cin, cout = w.shape
        n = x.shape[0]
        colsum = _colsum_hint(dy) if (ctx.opts.closed_form_bias_grad and ctx.has_bias) else None
        dy = dy.contiguous()
        dx = dk = db = None
        opts = ctx.opts
        lp = opts.low_precision
        if dy.dtype == torch.bfloat16 and (opts.prec_id != 1 or cout % 8 != 0):
            dy = dy.float()
        if ctx.needs_input_grad[0]:
            if lp:     # the data gradient multiplies by W^T: its K-major form [Cin][Cout] is the kernel itself
                tw = _twins_for(w, dy, opts, cin, cout, cin, cout)
                if (dbranch is not None and tw and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
                        and dbranch.dtype == torch.bfloat16 and opts.prec_id == 1 and cout % 8 == 0 and dy.stride(0) % 8 == 0
                        and cin % 4 == 0 and _lib.load().agb_dense_split_hint(n, cout, cin) == 1):
                    # bf16 rows: the branch gradient joins in fp32 before the data gradient's one rounding
                    dbranch = dbranch.contiguous()
                    dx = torch.empty(n, cin, dtype=torch.bfloat16, device=dy.device)
                    ev = _prof_begin("dgrad1x1", 1, cout, cin, n, False, 1, n)
                    _lib.call("agb_spconv_bwd_data_h", _P16(dy), dy.stride(0), _P16(tw[0]), None, 0, 0, _P16(dx), dx.stride(0),
                              n, 1, cin, cout, None, None, None, 0, 1, None, _P16(dbranch), dbranch.stride(0), _lib.stream())
                    _prof_end(ev, "dgrad1x1", 1, cout, cin, n, int(n), False, 1, n)
                    dbranch = None
                else:
                    dx = spconv_forward_raw(dy, None, None, 0, None, n, 1, cout, cin, "dgrad1x1", None, None,
                                            None if tw else w, opts=opts, w16=tw[0] if tw else None,
                                            out_bf16=(x.dtype == torch.bfloat16) if dy.dtype == torch.bfloat16 else None)
            elif dbranch is not None and dy.dtype == torch.float32 and dbranch.dtype == torch.float32:
                wt = torch.empty(cout, cin, dtype=torch.float32, device=w.device)
                if ctx.needs_input_grad[1]:
                    dk = torch.empty(cin, cout, dtype=torch.float32, device=w.device)
                _lib.call("agb_spconv_weight_transpose_z", _P(w), _P(wt), _P(dk), 1, cin, cout, _lib.stream())
                dx = _dense_dgrad_add(dy, wt, dbranch, n, cout, cin)
                dbranch = None
            else:
                wt = torch.empty(cout, cin, dtype=torch.float32, device=w.device)
                if ctx.needs_input_grad[1]:   # the weight-gradient buffer is cleared by the same launch
                    dk = torch.empty(cin, cout, dtype=torch.float32, device=w.device)
                _lib.call("agb_spconv_weight_transpose_z", _P(w), _P(wt), _P(dk), 1, cin, cout, _lib.stream())
                dx = spconv_forward_raw(dy, wt, None, 0, None, n, 1, cout, cin, "dgrad1x1", opts=opts)
        if ctx.needs_input_grad[1]:
            if dk is None:
                dk = zeros_f32((cin, cout), w.device)
            ev = _prof_begin("wgrad1x1", 1, cin, cout, n)
            weight_grad_raw(x, dy, None, dk, n, 1, cin, cout, opts)
            _prof_end(ev, "wgrad1x1", 1, cin, cout, n, int(n))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = (colsum if (colsum is not None and colsum.numel() == cout)
                  else dy.sum(0, dtype=torch.float32)).reshape(ctx.bias_shape)
        if dx is not None and dx.dtype != x.dtype:
            dx = dx.to(x.dtype)
        if dbranch is not None:      # (join form where the fused store does not apply: the addition autograd would have made)
            dx = dbranch if dx is None else dx + dbranch.to(dx.dtype)
        return dx, dk, db, None


def dense_conv_join(feats, kernel, bias):
    """(y, feats_branch) = (1x1 convolution of feats, feats) for a residual block whose input also feeds the shortcut: the
    shortcut's gradient is added in this layer's data-gradient kernel (KernelOptions.join_dgrad; fp32 rows: ConvArgs.addend of
    the register-accumulator kernel, bf16 rows: agb_spconv_bwd_data_h — the sum in fp32 before the one rounding).
    Bottleneck / SEBottleneck: resnet_block.py:93-133, senet_block.py:99-147."""
    if (current().join_dgrad and feats.is_cuda and feats.requires_grad and torch.is_grad_enabled()):
        y, branch = DenseConvFunction.apply(feats, kernel, bias, True)
        return take_bn_hint(y), branch
    return take_bn_hint(DenseConvFunction.apply(feats, kernel, bias, False)), feats


def dense_product(x, w, kind="fwd1x1", bn_stats=False, opts=None):
    """x [n, cin] @ w [cin, cout] on the identity-map convolution kernels in the operand precision of the options in force
    (no autograd: building block of the fused Functions; a backward pass hands in the options its node was created under).
    Widths as in ``DenseConvFunction.supported``."""
    opts = opts or current()
    cin, cout = w.shape
    w = w.contiguous()
    if opts.low_precision:
        return spconv_forward_raw(x, None, None, 0, None, x.shape[0], 1, cin, cout, kind, None, None, w.t().contiguous(),
                                  opts=opts)
    return spconv_forward_raw(x, w, None, 0, None, x.shape[0], 1, cin, cout, kind, bn_stats=bn_stats, opts=opts)


def dense_weight_grad(x, dy, opts=None):
    """x^T [cin, n] @ dy [n, cout] (no autograd)."""
    opts = opts or current()
    n, cin = x.shape
    cout = dy.shape[1]
    dk = zeros_f32((cin, cout), x.device)
    ev = _prof_begin("wgrad1x1", 1, cin, cout, n)
    weight_grad_raw(x, dy, None, dk, n, 1, cin, cout, opts)
    _prof_end(ev, "wgrad1x1", 1, cin, cout, n, int(n))
    return dk


class DenseLinearFunction(torch.autograd.Function):
    """y = x @ weight.T + bias with nn.Linear's parameter layout (weight [out, in]) on this library's own MFMA kernels
    (the identity-map convolution kernels of csrc/spconv.hip), in the operand precision of the KernelOptions in force: the shared
    per-point MLP of MinkowskiPointNet (PointNet.py:16-28), KPConv's unary blocks (blocks.py:499-535) and its
    feature x kernel-weight contraction (blocks.py:396-400).  Feature widths are zero-padded to a multiple of 4 (>= 12)."""

    @staticmethod
    def forward(ctx, x, weight, bias, join=False):
        """join: also return x itself as a second output (the branch of a residual block that bypasses this layer); the
        gradient that comes back for it is then the ADDEND of this layer's data gradient — dx = dx_branch + dy @ weight leaves
        in one pass (fp32 operands, unpadded widths: dense_linear_join checks)."""
        n, cin = x.shape
        cout = weight.shape[0]
        ctx.join = bool(join)
        # both widths are the reduction dimension of one of the three products: at least 12, multiples of 4
        cin_p, cout_p = max(12, (cin + 3) // 4 * 4), max(12, (cout + 3) // 4 * 4)
        if join and (cin_p != cin or cout_p != cout):
            raise _lib.AgbError("the join form of the dense layer takes unpadded widths (multiples of 4, >= 12): "
                                "dense_linear_join falls back to the plain form otherwise")
        xp = (x if cin_p == cin else F.pad(x, (0, cin_p - cin))).contiguous()
        wp = weight if (cin_p == cin and cout_p == cout) else F.pad(weight, (0, cin_p - cin, 0, cout_p - cout))
        wp = wp.contiguous()                                   # [out, in]: K-major for the forward product
        b = None
        if bias is not None:
            b = (bias if cout_p == cout else F.pad(bias, (0, cout_p - cout))).contiguous()
        opts = ctx.opts = current()
        if opts.low_precision:
            y = spconv_forward_raw(xp, None, None, 0, b, n, 1, cin_p, cout_p, "fwd1x1", None, None, wp, opts=opts)
        else:
            # W^T [in, out]: from the model's per-step batch of transposes where it keeps one (fused_blocks.LinearTransposes:
            # one launch per optimiser step for all Linear layers of a backbone), else made here
            pre = getattr(weight, "agb_wt", None)
            if (pre is not None and wp is weight and pre[1] == _WEIGHT_EPOCH[0] and pre[2] == weight._version
                    and pre[0].device == x.device):
                wt = pre[0]
            else:
                wt = torch.empty(cin_p, cout_p, dtype=torch.float32, device=x.device)
                _lib.call("agb_spconv_weight_transpose", _P(wp), _P(wt), 1, cout_p, cin_p, _lib.stream())
            y = spconv_forward_raw(xp, wt, None, 0, b, n, 1, cin_p, cout_p, "fwd1x1",
                                   bn_stats=any(ctx.needs_input_grad) and cout_p == cout, opts=opts)
        ctx.save_for_backward(xp, wp)
        ctx.dims = (cin, cout, cin_p, cout_p, bias is not None)
        if join:
            return y, x
        return y if cout_p == cout else y[:, :cout].contiguous()

    @staticmethod
    def backward(ctx, dy, dbranch=None):
        xp, wp = ctx.saved_tensors
        cin, cout, cin_p, cout_p, has_bias = ctx.dims
        n = xp.shape[0]
        colsum = _colsum_hint(dy) if (ctx.opts.closed_form_bias_grad and has_bias) else None
        dy = dy.contiguous()
        dyp = dy if cout_p == cout else F.pad(dy, (0, cout_p - cout)).contiguous()
        dx = dw = db = None
        opts = ctx.opts
        lp = opts.low_precision
        if ctx.needs_input_grad[0]:
            # dX = dY @ weight: weight [out, in] is the [K, N] operand as stored; its K-major form is the transpose
            if ctx.join and dbranch is not None:
                dxp = _dense_dgrad_add(dyp, wp, dbranch, n, cout_p, cin_p)
                dbranch = None
            elif lp:
                wkm = wp.t().contiguous()
                dxp = spconv_forward_raw(dyp, None, None, 0, None, n, 1, cout_p, cin_p, "dgrad1x1", None, None, wkm,
                                         opts=opts)
            else:
                dxp = spconv_forward_raw(dyp, wp, None, 0, None, n, 1, cout_p, cin_p, "dgrad1x1", opts=opts)
            dx = dxp if cin_p == cin else dxp[:, :cin].contiguous()
        if ctx.needs_input_grad[1]:
            dwp = zeros_f32((cout_p, cin_p), dy.device)
            ev = _prof_begin("wgrad1x1", 1, cout_p, cin_p, n)
            # dWeight [out, in] = dY^T X: the weight-gradient kernel with the roles of the operands swapped
            weight_grad_raw(dyp, xp, None, dwp, n, 1, cout_p, cin_p, opts)
            _prof_end(ev, "wgrad1x1", 1, cout_p, cin_p, n, int(n))
            dw = dwp if (cin_p == cin and cout_p == cout) else dwp[:cout, :cin].contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum if (colsum is not None and colsum.numel() == cout) else dy.sum(0)
        if dbranch is not None:      # (join form whose own data gradient was not asked for, or is None)
            dx = dbranch if dx is None else dx + dbranch
        return dx, dw, db, None


def _dense_dgrad_add(dy, w, addend, n, cout, cin):
    """dx [n, cin] = addend + dy [n, cout] @ w [cout, cin] (agb_spconv_bwd_data on the identity map: the addend joins in the
    kernel's final store — after its own sum, the value a separate addition gives)."""
    addend = addend.contiguous()
    split = _lib.load().agb_dense_split_hint(n, cout, cin)
    partial = torch.empty(split, n, cin, dtype=torch.float32, device=dy.device) if split > 1 else None
    dx = torch.empty(n, cin, dtype=torch.float32, device=dy.device)
    ev = _prof_begin("dgrad1x1", 1, cout, cin, n, False, split, n)
    _lib.call("agb_spconv_bwd_data", _P(dy), dy.stride(0), _P(w), None, 0, 0, _P(dx), dx.stride(0), n, 1, cin, cout, None,
              None, None, 0, split, _P(partial), _P(addend), addend.stride(0), _lib.stream())
    _prof_end(ev, "dgrad1x1", 1, cout, cin, n, int(n), False, split, n)
    return dx


def dense_linear_join(x, weight, bias=None):
    """(y, x_branch) = (nn.Linear(x), x) for a residual block whose input also feeds a branch that bypasses this layer: the
    branch's gradient is added in this layer's data-gradient kernel (KernelOptions.join_dgrad; KPConv blocks.py:640-668:
    unary1 and the shortcut both read the block input).  Falls back to (dense_linear(x), x) where the join form does not
    apply (bf16 operand modes, padded widths, no gradient wanted)."""
    cin, cout = x.shape[1], weight.shape[0]
    opts = current()
    if (opts.join_dgrad and not opts.low_precision and x.is_cuda and x.dim() == 2 and x.shape[0] > 0
            and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled()
            and cin >= 12 and cin % 4 == 0 and cout >= 12 and cout % 4 == 0):
        y, branch = DenseLinearFunction.apply(x, weight, bias, True)
        return take_bn_hint(y), branch
    return dense_linear(x, weight, bias), x


def dense_linear(x, weight, bias=None):
    """nn.Linear semantics on the library's own kernels for device tensors with >= 1 row; plain F.linear otherwise."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and x.dtype == torch.float32:
        return take_bn_hint(DenseLinearFunction.apply(x, weight, bias, False))
    return F.linear(x, weight, bias)


class MaxPoolFunction(torch.autograd.Function):
    """Strided max pooling over a kernel map.  The winner of every output element is kept as its kernel-offset index
    (one byte; K3 <= 255) — the transposed map nbrT uses the same offset numbering, so the gradient pass only has to
    compare bytes — or as the input row (int32) for larger kernels."""

    @staticmethod
    def forward(ctx, feats, nbr, nbrT, n_in, n_out):
        c = feats.shape[1]
        x = _pad_cols(feats, 4)
        cp = x.shape[1]
        K3 = nbr.shape[0]
        y = torch.empty(n_out, cp, dtype=x.dtype, device=x.device)
        small = K3 <= 255
        arg = torch.empty(n_out, cp, dtype=torch.uint8 if small else torch.int32, device=x.device)
        _lib.call(("agb_maxpool_fwd_k" if small else "agb_maxpool_fwd") + _lib.sfx(x), _lib.rows(x), x.stride(0), _P(nbr),
                  nbr.stride(0), _lib.rows(y), y.stride(0), _P(arg), n_out, K3, cp, _lib.stream())
        ctx.save_for_backward(arg, nbrT)
        ctx.dims = (c, cp, n_in, K3, small)
        return y if cp == c else y[:, :c].contiguous()

    @staticmethod
    def backward(ctx, dy):
        arg, nbrT = ctx.saved_tensors
        c, cp, n_in, K3, small = ctx.dims
        dy = _pad_cols(dy, 4)
        dx = torch.empty(n_in, cp, dtype=dy.dtype, device=dy.device)
        _lib.call(("agb_maxpool_bwd_k" if small else "agb_maxpool_bwd") + _lib.sfx(dy), _lib.rows(dy), dy.stride(0), _P(arg),
                  _P(nbrT), nbrT.stride(0), _lib.rows(dx), dx.stride(0), n_in, K3, cp, _lib.stream())
        return (dx if cp == c else dx[:, :c].contiguous()), None, None, None, None


_MODES = {"sum": 0, "avg": 1, "max": 2}


def segment_reduce(a, bm, ptr, B, mode_id):
    """[N, C] -> [B, C] per-batch reduction (optionally of a*bm); long segments are cut into row chunks so the
    launch fills the chip, partials folded in a fixed order."""
    n, c = a.shape
    y = torch.empty(B, c, dtype=torch.float32, device=a.device)
    arg = torch.empty(B, c, dtype=torch.int32, device=a.device) if mode_id == 2 else None
    splits = max(1, min(64, (n // max(B, 1)) // 256))
    part = part_arg = None
    if splits > 1:
        part = torch.empty(B * splits, c, dtype=torch.float32, device=a.device)
        part_arg = torch.empty(B * splits, c, dtype=torch.int32, device=a.device) if mode_id == 2 else None
    _lib.call("agb_segment_reduce" + _lib.sfx(a, bm), _lib.rows(a), a.stride(0), _lib.rows(bm), 0 if bm is None else bm.stride(0),
              _P(ptr), B, c, mode_id, splits, _P(part), _P(part_arg), _P(y), _P(arg), _lib.stream())
    return y, arg


class GlobalPoolFunction(torch.autograd.Function):
    """Per-batch segment reduction of [N, C] rows -> [B, C]."""

    @staticmethod
    def forward(ctx, feats, coords, ptr, B, mode):
        x = feats.contiguous()
        n, c = x.shape
        m = _MODES[mode]
        y, arg = segment_reduce(x, None, ptr, B, m)
        ctx.save_for_backward(coords, ptr, arg if arg is not None else torch.empty(0))
        ctx.dims = (n, c, B, m)
        ctx.rows_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        coords, ptr, arg = ctx.saved_tensors
        n, c, B, m = ctx.dims
        dy = dy.contiguous()
        if m == 2:
            dx = torch.zeros(n, c, dtype=ctx.rows_dtype, device=dy.device)
            _lib.call("agb_segment_max_bwd" + _lib.sfx(dx), _P(dy), _P(arg), _lib.rows(dx), dx.stride(0), B, c, _lib.stream())
            return dx, None, None, None, None
        if c % 4 != 0:
            raise _lib.AgbError("global sum/avg pooling gradient needs a channel count that is a multiple of 4")
        dx = torch.empty(n, c, dtype=ctx.rows_dtype, device=dy.device)
        _lib.call("agb_segment_broadcast" + _lib.sfx(dx), _P(dy), _P(coords), _P(ptr), None, 0, _lib.rows(dx), dx.stride(0), n, c,
                  1 if m == 1 else 0, _lib.stream())
        return dx, None, None, None, None


class BroadcastMulFunction(torch.autograd.Function):
    """out[r, :] = x[r, :] * s[batch(r), :]   (ME.MinkowskiBroadcastMultiplication)."""

    @staticmethod
    def forward(ctx, x, s, coords, ptr):
        x = x.contiguous()
        s = s.contiguous()
        n, c = x.shape
        if c % 4 != 0:
            raise _lib.AgbError("broadcast multiplication needs a channel count that is a multiple of 4")
        out = torch.empty_like(x)
        _lib.call("agb_segment_broadcast" + _lib.sfx(x), _P(s), _P(coords), _P(ptr), _lib.rows(x), x.stride(0), _lib.rows(out),
                  out.stride(0), n, c, 0, _lib.stream())
        ctx.save_for_backward(x, s, coords, ptr)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, s, coords, ptr = ctx.saved_tensors
        dout = dout.contiguous()
        n, c = x.shape
        B = s.shape[0]
        dx = ds = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("agb_segment_broadcast" + _lib.sfx(dout), _P(s), _P(coords), _P(ptr), _lib.rows(dout), dout.stride(0),
                      _lib.rows(dx), dx.stride(0), n, c, 0, _lib.stream())
        if ctx.needs_input_grad[1]:
            ds, _ = segment_reduce(dout, x, ptr, B, 0)
        return dx, ds, None, None
