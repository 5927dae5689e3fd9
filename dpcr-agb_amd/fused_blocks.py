"""One library call per network block: the stem (conv 7^3 -> BatchNorm -> act -> max pool; reference SENet.py:47-53) and
the squeeze-excite basic block (senet_block.py:80-96 on resnet_block.py:62-73) as ONE autograd node each, forward and
backward one C entry point each (csrc/net.hip: agb_net_stem_fwd/bwd, agb_net_block_fwd/bwd).

The operator-by-operator path (me_compat / sparse_ops / norm_ops / se_ops: ~195 launches per MSENet14 step, each driven
from Python) stays the general implementation — every module combination, operand precision and storage type; this module
takes the fp32 BatchNorm / ReLU-GELU / SEBasicBlock networks (MSENet14/18/34) when nothing non-standard is configured and
enqueues the SAME kernels in the same order through the library: bit-identical results (tests/test_fused_blocks_gpu.py),
a fifth of the host work.  ``KernelOptions.fused_blocks = False`` (or AGB_FUSED_BLOCKS=0) turns it off.
"""
import ctypes
import struct

import torch
import torch.nn as nn

from . import _lib
from . import me_compat as ME
from . import sparse_ops
from .norm_ops import ACT_IDS
from .se_ops import ACT_IDS as SE_ACT_IDS, MAX_HIDDEN as MAX_SE_HIDDEN

_V, _I = _lib.c_void_p, _lib.c_int
_lib.declare("agb_net_field_count", [])
_lib.declare("agb_net_fields", [])           # (returns const char*: restype set in _fields)
_lib.declare("agb_net_stem_bytes", [_V, _I])  # (size_t)
_lib.declare("agb_net_block_bytes", [_V, _I])
for _n in ("agb_net_stem_fwd", "agb_net_stem_bwd", "agb_net_block_fwd", "agb_net_block_bwd"):
    _lib.declare(_n, [_V, _V, ctypes.c_size_t, _V, ctypes.c_size_t, _V])

_FIELDS = None      # name -> index of the library's field table (agb_net_fields)
_NF = 0


def _fields():
    global _FIELDS, _NF
    if _FIELDS is None:
        lib = _lib.load()
        lib.agb_net_fields.restype = ctypes.c_char_p
        lib.agb_net_fields.argtypes = []
        names = [n for n in lib.agb_net_fields().decode().split(",") if n]
        _NF = lib.agb_net_field_count()
        if len(names) != _NF:
            raise _lib.AgbError("agb_net_fields / agb_net_field_count disagree")
        _FIELDS = {n: i for i, n in enumerate(names)}
        for fn in ("agb_net_stem_bytes", "agb_net_block_bytes"):
            f = getattr(lib, fn)
            f.restype = ctypes.c_size_t
            f.argtypes = [_V, _I]
    return _FIELDS


def _dbits(v):
    return struct.unpack("q", struct.pack("d", float(v)))[0]


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_SCRATCH = {}      # raw stream handle -> uint8 scratch arena of this stream (grown on demand; calls on a stream are ordered)


def _scratch(nbytes, device):
    key = (_lib.stream(), device.index)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _SCRATCH[key] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
    return buf


def release_scratch():
    """Drop the per-stream scratch arenas (tests; a model that is done)."""
    _SCRATCH.clear()


class _Table:
    """The 64-bit field table of one call (csrc/net.hip AGB_NET_FIELDS)."""
    __slots__ = ("buf", "idx")

    def __init__(self):
        self.idx = _fields()
        self.buf = (ctypes.c_int64 * _NF)()

    def set(self, **kw):
        buf, idx = self.buf, self.idx
        for k, v in kw.items():
            buf[idx[k]] = int(v)

    def setp(self, prefix, **kw):
        buf, idx = self.buf, self.idx
        for k, v in kw.items():
            buf[idx[prefix + k]] = int(v)


def _bn_fields(tab, prefix, bn, cout):
    """BatchNorm fields of one convolution, as norm_ops.batch_norm_act hands them to the statistics kernel."""
    rm, rv = bn.running_mean, bn.running_var
    mom, nbt = 0.0, None
    if bn.training and rm is not None:
        mom, nbt = bn.momentum, bn.num_batches_tracked
    tab.setp(prefix, g=_ptr(bn.weight), be=_ptr(bn.bias), rm=_ptr(rm), rv=_ptr(rv), nbt=_ptr(nbt), eps=_dbits(bn.eps),
             mom=_dbits(mom))


def _bn_ok(norm):
    if not isinstance(norm, ME.MinkowskiBatchNorm):
        return False
    bn = norm.bn
    return bn.momentum is not None or bn.running_mean is None


def _uses_batch_stats(bn):
    return bn.training or bn.running_mean is None


def _conv_ok(conv, k, stride=None):
    return (isinstance(conv, ME.MinkowskiConvolution) and conv.kernel_size == k and conv.dilation == 1
            and (stride is None or conv.stride == stride) and not conv.use_mm
            and conv.in_channels % 4 == 0 and conv.in_channels >= 12 and conv.out_channels % 4 == 0
            and conv.out_channels >= 12 and conv.kernel.is_contiguous())


def block_supported(blk):
    """Static part of the eligibility of one residual block (its module structure)."""
    from .backbones.sparse import MinkowskiDropPath, SEBasicBlock
    if type(blk) is not SEBasicBlock:
        return False
    s = blk.conv1.stride
    if not (_conv_ok(blk.conv1, 3) and _conv_ok(blk.conv2, 3, 1) and _bn_ok(blk.norm1) and _bn_ok(blk.norm2)):
        return False
    C = blk.conv1.out_channels
    if blk.conv2.in_channels != C or blk.conv2.out_channels != C or s not in (1, 2):
        return False
    if getattr(blk.relu, "act_name", None) not in ("relu", "gelu"):
        return False
    fc = blk.se.fc
    if not (len(fc) == 4 and isinstance(fc[0], ME.MinkowskiLinear) and isinstance(fc[2], ME.MinkowskiLinear)
            and isinstance(fc[3], ME.MinkowskiSigmoid) and getattr(fc[1], "act_name", None) in ("relu", "gelu")
            and fc[0].linear.out_features <= MAX_SE_HIDDEN and fc[0].linear.in_features == C
            and fc[2].linear.out_features == C):
        return False
    if not isinstance(blk.drop_path, (MinkowskiDropPath, nn.Identity)):
        return False
    ds = blk.downsample
    if isinstance(ds, nn.Identity):
        return s == 1 and blk.conv1.in_channels == C
    return (isinstance(ds, nn.Sequential) and len(ds) == 2 and _conv_ok(ds[0], 1, s) and s == 2 and _bn_ok(ds[1])
            and ds[0].in_channels == blk.conv1.in_channels and ds[0].out_channels == C)


def stem_supported(stem):
    from .backbones.sparse import ConvNormActivation
    if not (isinstance(stem, nn.Sequential) and len(stem) == 2 and isinstance(stem[0], ConvNormActivation)
            and isinstance(stem[1], ME.MinkowskiMaxPooling)):
        return False
    cna, pool = stem[0], stem[1]
    conv = cna.conv
    act = getattr(cna.act, "act_name", None) if not isinstance(cna.act, nn.Identity) else "none"
    return (isinstance(conv, ME.MinkowskiConvolution) and conv.in_channels == 3 and conv.out_channels == 64
            and conv.stride == 1 and conv.dilation == 1 and conv.kernel_size % 2 == 1 and conv.kernel_size <= 9
            and conv.kernel.is_contiguous() and _bn_ok(cna.norm) and act in ("none", "relu", "gelu")
            and pool.kernel_size ** 3 <= 255 and pool.dilation == 1)


def options_allow(opts):
    return (getattr(opts, "fused_blocks", True) and opts.prec_id == 0 and not opts.rows_bf16 and opts.fused_tail
            and opts.closed_form_bias_grad and sparse_ops.PROFILE is None)


def _opt_fields(tab, opts, training):
    tab.set(cmp_mode=opts.cmp_mode, cmp_il=opts.cmp_interleave, dw_variant=opts.dw_variant,
            det=int(opts.deterministic_wgrad), persistent=int(sparse_ops.PERSISTENT_WGRAD), training=int(training))


# ------------------------------------------------------------------------------------------------------------ gradients
class _GradLayout:
    """Offsets (in floats, 64-float aligned) of the parameter gradients of one block inside one flat buffer.  zero_first: the
    indices whose buffers must start at zero — laid out first and contiguously (`zero_floats`), cleared by ONE fill."""

    def __init__(self, shapes, zero_first=()):
        order = list(zero_first) + [i for i in range(len(shapes)) if i not in zero_first]
        self.offsets, off = [None] * len(shapes), 0
        self.zero_floats = 0
        for i in order:
            shp = shapes[i]
            n = 1
            for d in shp:
                n *= d
            self.offsets[i] = (off, n, shp)
            off += (n + 63) // 64 * 64
            if i in zero_first:
                self.zero_floats = off
        self.total = max(off, 64)

    def views(self, flat):
        return [flat[o:o + n].view(shp) for o, n, shp in self.offsets]

    def ptrs(self, flat):
        base = flat.data_ptr()
        return [base + 4 * o for o, _n, _s in self.offsets]


# ------------------------------------------------------------------------------------- transposed weights, kept per step
class WeightTransposes:
    """W^T [K3][cout][cin] of every convolution the fused blocks of one backbone differentiate through, made by ONE launch
    (agb_spconv_weight_transpose_batched) the first time a step asks for them and kept until the weights change (the fused
    optimiser's weight epoch / torch's version counters) — instead of one transpose launch per layer and backward pass."""

    def __init__(self, convs, device):
        self.convs = list(convs)
        self.key = None
        total, tiles, rows = 0, 0, []
        self.offset = {}
        for c in self.convs:
            K3, cin, cout = c.kernel.shape
            self.offset[id(c)] = total
            rows.append([0, 0, K3, cin, cout, tiles])
            total += K3 * cin * cout
            tiles += K3 * ((cin + 63) // 64) * ((cout + 63) // 64)
        self.flat = torch.empty(max(total, 1), dtype=torch.float32, device=device)
        self.tiles = tiles
        self.rows = rows
        self.table = None
        self.ptrs = None

    def _state(self):
        return (sparse_ops._WEIGHT_EPOCH[0],) + tuple(c.kernel._version for c in self.convs) + \
            tuple(c.kernel.data_ptr() for c in self.convs)

    def ensure(self):
        """Transposes of the CURRENT weights on the current stream (a no-op while nothing changed)."""
        key = self._state()
        if key == self.key:
            return
        ptrs = key[1 + len(self.convs):]
        if ptrs != self.ptrs:       # (first use, or the parameters moved: model.to(), load_state_dict(assign=True))
            base = self.flat.data_ptr()
            tab = [[p, base + 4 * self.offset[id(c)]] + r[2:] for c, p, r in zip(self.convs, ptrs, self.rows)]
            self.table = torch.tensor(tab, dtype=torch.int64).to(self.flat.device)
            self.ptrs = ptrs
        _lib.call("agb_spconv_weight_transpose_batched", self.table.data_ptr(), len(self.convs), self.tiles, _lib.stream())
        self.key = key

    def ptr(self, conv):
        return self.flat.data_ptr() + 4 * self.offset[id(conv)]


_lib.declare("agb_spconv_weight_transpose_batched", [_V, _I, _lib.c_ll, _V])


def weight_transposes(model, blocks):
    """The model's WeightTransposes over the convolutions of `blocks` (made on first use)."""
    wt = model.__dict__.get("_agb_wt")
    if wt is not None and wt.convs and wt.convs[0].kernel.device != wt.flat.device:
        wt = None                                   # (the model moved to another device since)
    if wt is None:
        convs = []
        for blk in blocks:
            convs += [blk.conv1, blk.conv2] + ([] if isinstance(blk.downsample, nn.Identity) else [blk.downsample[0]])
        wt = model.__dict__["_agb_wt"] = WeightTransposes(convs, convs[0].kernel.device)
    return wt


class _LinearAsConv:
    """What WeightTransposes asks of a layer, for an nn.Linear: its weight [out, in] as a one-offset kernel [1, out, in] (the
    transposed form [1, in, out] is the [K, N] operand of the forward product)."""
    __slots__ = ("lin",)

    def __init__(self, lin):
        self.lin = lin

    @property
    def kernel(self):
        w = self.lin.weight.detach()          # (shares storage and version counter with the parameter)
        return w.view(1, w.shape[0], w.shape[1])


class LinearTransposes(WeightTransposes):
    """W^T of every nn.Linear a backbone runs through sparse_ops.dense_linear (KPConv's unary blocks: 37 layers), by ONE batched
    launch per optimiser step instead of one transpose launch per layer and forward pass; every weight carries its view
    (``weight.agb_wt``: view, weight epoch, version) for DenseLinearFunction to pick up."""

    def __init__(self, linears, device):
        self.linears = [l for l in linears if l.weight.shape[0] % 4 == 0 and l.weight.shape[1] % 4 == 0
                        and min(l.weight.shape) >= 12 and l.weight.dtype == torch.float32]
        super().__init__([_LinearAsConv(l) for l in self.linears], device)

    def _state(self):
        ws = [l.weight for l in self.linears]
        return (sparse_ops._WEIGHT_EPOCH[0],) + tuple(w._version for w in ws) + tuple(w.data_ptr() for w in ws)

    def ensure(self):
        before = self.key
        super().ensure()
        if self.key != before or not self.linears or getattr(self.linears[0].weight, "agb_wt", None) is None:
            epoch = sparse_ops._WEIGHT_EPOCH[0]
            for lin, conv in zip(self.linears, self.convs):
                w = lin.weight
                o = self.offset[id(conv)]
                w.agb_wt = (self.flat[o:o + w.numel()].view(w.shape[1], w.shape[0]), epoch, w._version)


def linear_transposes(model, linears):
    """The model's LinearTransposes (made on first use; `linears`: callable returning the nn.Linear modules)."""
    lt = model.__dict__.get("_agb_lt")
    if lt is not None and lt.linears and lt.linears[0].weight.device != lt.flat.device:
        lt = None                                   # (the model moved to another device since)
    if lt is None:
        mods = list(linears())
        if not mods:
            return None
        lt = model.__dict__["_agb_lt"] = LinearTransposes(mods, mods[0].weight.device)
    return lt


# ----------------------------------------------------------------------------------------------------------------- stem
class _StemCall:
    __slots__ = ("tab", "saved_bytes", "fwd_bytes", "bwd_bytes", "layout", "has", "n", "n_pool", "C", "keepalive")


class StemFunction(torch.autograd.Function):
    """conv K^3 (3 -> 64, grid-probing) -> BatchNorm -> act -> max pool 3^3 stride 2 as one node."""

    @staticmethod
    def forward(ctx, feats, call, kernel, bias, gamma, beta):
        tab = call.tab
        dev = feats.device
        y = torch.empty(call.n_pool, call.C, dtype=torch.float32, device=dev)
        saved = torch.empty(call.saved_bytes, dtype=torch.uint8, device=dev)
        scratch = _scratch(max(call.fwd_bytes, call.bwd_bytes), dev)
        tab.set(y=y.data_ptr(), ldy=call.C)
        _lib.call("agb_net_stem_fwd", tab.buf, saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(),
                  _lib.stream())
        ctx.call, ctx.arena = call, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        call, saved = ctx.call, ctx.arena
        tab = call.tab
        dy = dy.contiguous()
        dev = dy.device
        flat = torch.empty(call.layout.total, dtype=torch.float32, device=dev)
        gp = call.layout.ptrs(flat)
        tab.setp("c1_", dw=gp[0], db=gp[1], dg=gp[2], dbe=gp[3])
        tab.set(dy=dy.data_ptr(), lddy=dy.stride(0))
        scratch = _scratch(call.bwd_bytes, dev)
        _lib.call("agb_net_stem_bwd", tab.buf, saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(),
                  _lib.stream())
        g = call.layout.views(flat)
        has = call.has
        return (None, None, g[0], g[1] if has[1] else None, g[2] if has[2] else None, g[3] if has[3] else None)


def run_stem(stem, x, opts):
    """The fused stem on SparseTensor x, or None when this batch / mode is not one it takes (the caller then runs the
    modules)."""
    cna, pool = stem[0], stem[1]
    conv, bn = cna.conv, cna.norm.bn
    cm, ts = x.coordinate_manager, x._ts
    F = x.F
    if not (F.is_cuda and F.dtype == torch.float32 and F.dim() == 2 and F.stride(1) == 1 and F.shape[1] == 3):
        return None
    grad = torch.is_grad_enabled()
    if F.requires_grad and grad:
        return None                                  # (a stem whose input needs a gradient takes the kernel-map path)
    if ("fwd", ts, conv.kernel_size, 1, 1) in cm.kernel_maps:
        return None
    probe = cm.grid_probe(ts, conv.kernel_size, 1, 1)
    if probe is None:
        return None
    training = _uses_batch_stats(bn)
    needs_grad = grad and any(p.requires_grad for p in (conv.kernel, bn.weight) if p is not None)
    if needs_grad and not training:
        return None
    coords, grid, desc = probe
    n = cm.level(ts).n
    K = conv.kernel_size
    if not (0 < n < (1 << 24)) or opts.dw_variant != 0:
        return None
    ts_out = ts * pool.stride
    pnbr = cm.kernel_map(ts, pool.kernel_size, pool.stride, pool.dilation)
    pnbrT = cm.transposed_map(ts, pool.kernel_size, pool.stride, pool.dilation) if needs_grad else None
    n_pool = cm.level(ts_out).n
    if n_pool < 1:
        return None
    C = conv.out_channels
    tab = _Table()
    _opt_fields(tab, opts, training)
    act = "none" if isinstance(cna.act, nn.Identity) else cna.act.act_name
    tab.set(n_in=n, n_out=n, n_pool=n_pool, B=cm.batch_size, coords=coords.data_ptr(), act=ACT_IDS[act],
            feat=F.data_ptr(), ldf=F.stride(0), fdim=F.shape[1], grid=grid.data_ptr(), desc=ctypes.addressof(desc), K=K,
            pool_nbr=pnbr.data_ptr(), pool_nbr_ld=pnbr.stride(0), pool_nbrT=_ptr(pnbrT),
            pool_nbrT_ld=0 if pnbrT is None else pnbrT.stride(0), pool_K3=pool.kernel_size ** 3)
    tab.setp("c1_", w=conv.kernel.data_ptr(), b=_ptr(conv.bias), K3=K ** 3, cin=3, cout=C)
    _bn_fields(tab, "c1_", bn, C)
    lib = _lib.load()
    call = _StemCall()
    call.tab, call.n, call.n_pool, call.C = tab, n, n_pool, C
    call.saved_bytes = lib.agb_net_stem_bytes(tab.buf, 0)
    call.fwd_bytes = lib.agb_net_stem_bytes(tab.buf, 1)
    call.bwd_bytes = lib.agb_net_stem_bytes(tab.buf, 2) if needs_grad else 0
    call.layout = _GradLayout([tuple(conv.kernel.shape), (1, C), (C,), (C,)])
    call.has = (True, conv.bias is not None, bn.weight is not None, bn.bias is not None)
    call.keepalive = (F, coords, grid, desc, pnbr, pnbrT)
    out = StemFunction.apply(F, call, conv.kernel, conv.bias, bn.weight, bn.bias)
    return ME.SparseTensor(out, coordinate_map_key=ME.CoordinateMapKey(ts_out), coordinate_manager=cm)


# ---------------------------------------------------------------------------------------------------------------- block
_BLOCK_PARAMS = ("c1w", "c1b", "g1", "be1", "c2w", "c2b", "g2", "be2", "cdw", "cdb", "gd", "bed", "sw1", "sb1", "sw2", "sb2")


class _BlockCall:
    __slots__ = ("tab", "saved_bytes", "fwd_bytes", "bwd_bytes", "layout", "has", "n_out", "n_in", "C", "Cin", "keepalive",
                 "down")


class SEBlockFunction(torch.autograd.Function):
    """A whole SEBasicBlock (two 3^3 convolutions, their BatchNorms, the optional 1x1 stride-2 downsample branch, squeeze-excite,
    drop path, residual join, activation) as one node: agb_net_block_fwd / agb_net_block_bwd."""

    @staticmethod
    def forward(ctx, x, call, *params):
        tab = call.tab
        dev = x.device
        y = torch.empty(call.n_out, call.C, dtype=torch.float32, device=dev)
        saved = torch.empty(call.saved_bytes, dtype=torch.uint8, device=dev)
        scratch = _scratch(max(call.fwd_bytes, call.bwd_bytes), dev)
        tab.set(x=x.data_ptr(), ldx=x.stride(0), y=y.data_ptr(), ldy=call.C)
        _lib.call("agb_net_block_fwd", tab.buf, saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(),
                  _lib.stream())
        ctx.call, ctx.arena = call, saved
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        call, saved = ctx.call, ctx.arena
        (x,) = ctx.saved_tensors
        tab = call.tab
        dy = dy.contiguous()
        dev = dy.device
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty(call.n_in, call.Cin, dtype=torch.float32, device=dev) if need_dx else None
        flat = torch.empty(call.layout.total, dtype=torch.float32, device=dev)
        gp = call.layout.ptrs(flat)
        tab.setp("c1_", dw=gp[0], db=gp[1], dg=gp[2], dbe=gp[3])
        tab.setp("c2_", dw=gp[4], db=gp[5], dg=gp[6], dbe=gp[7])
        tab.setp("cd_", dw=gp[8], db=gp[9], dg=gp[10], dbe=gp[11])
        tab.set(d_se_w1=gp[12], d_se_b1=gp[13], d_se_w2=gp[14], d_se_b2=gp[15], x=x.data_ptr(), dy=dy.data_ptr(),
                lddy=dy.stride(0), dx=_ptr(dx), lddx=call.Cin, need_dx=int(need_dx), gzero=flat.data_ptr(),
                gzero_bytes=4 * call.layout.zero_floats)
        scratch = _scratch(call.bwd_bytes, dev)
        _lib.call("agb_net_block_bwd", tab.buf, saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(),
                  _lib.stream())
        g = call.layout.views(flat)
        return (dx, None) + tuple(gi if h else None for gi, h in zip(g, call.has))


def _conv_fields(tab, prefix, conv, bn, cm, ts_in, opts, need_t, n_in, n_out):
    """Fields of one convolution: parameters, BatchNorm, kernel maps, tile tables / class partition; returns the tensors the
    table points at (kept alive by the caller)."""
    K, s = conv.kernel_size, conv.stride
    K3, cin, cout = K ** 3, conv.in_channels, conv.out_channels
    nbr = cm.kernel_map(ts_in, K, s, 1)
    keep = [nbr]
    tab.setp(prefix, w=conv.kernel.data_ptr(), b=_ptr(conv.bias), K3=K3, cin=cin, cout=cout, nbr=nbr.data_ptr(),
             nbr_ld=nbr.stride(0), nbrT=0, nbrT_ld=0, perm=0, tile_cls=0, cls_tab=0, n_tiles=0, tf=0, tf_t=0, tf_b=0, tb=0,
             tb_t=0, tb_b=0, wt=0)
    _bn_fields(tab, prefix, bn, cout)
    if s == 1 and K3 > 1:
        tf = sparse_ops.cmp_tile_table(nbr, n_out, K3, cin, cout, cin, cout, opts)
        if tf is not None:
            tab.setp(prefix, tf=tf.data_ptr(), tf_t=tf.shape[0], tf_b=tf.shape[1])
            keep.append(tf)
        if need_t:
            tb = sparse_ops.cmp_tile_table(nbr, n_in, K3, cout, cin, cout, cin, opts)
            if tb is not None:
                tab.setp(prefix, tb=tb.data_ptr(), tb_t=tb.shape[0], tb_b=tb.shape[1])
                keep.append(tb)
    elif s > 1 and need_t:
        nbrT = cm.transposed_map(ts_in, K, s, 1)
        perm, tile_cls, cls_tab, max_tiles = cm.transposed_plan(ts_in, K, s, 1)
        tab.setp(prefix, nbrT=nbrT.data_ptr(), nbrT_ld=nbrT.stride(0), perm=perm.data_ptr(), tile_cls=tile_cls.data_ptr(),
                 cls_tab=cls_tab.data_ptr(), n_tiles=max_tiles)
        keep += [nbrT, perm, tile_cls, cls_tab]
    return keep


def run_block(blk, x, opts, wt=None):
    """The fused SEBasicBlock on SparseTensor x, or None when this batch / mode is not one it takes.  wt: the model's
    WeightTransposes (the data gradients then read the step's cached W^T instead of transposing per layer)."""
    from .backbones.sparse import MinkowskiDropPath
    F = x.F
    if not (F.is_cuda and F.dtype == torch.float32 and F.dim() == 2 and F.is_contiguous()):
        return None
    cm, ts = x.coordinate_manager, x._ts
    c1, c2 = blk.conv1, blk.conv2
    if F.shape[1] != c1.in_channels:
        return None
    down = not isinstance(blk.downsample, nn.Identity)
    bns = [blk.norm1.bn, blk.norm2.bn] + ([blk.downsample[1].bn] if down else [])
    training = _uses_batch_stats(bns[0])
    if any(_uses_batch_stats(b) != training for b in bns):
        return None
    grad = torch.is_grad_enabled()
    need_t = grad and F.requires_grad
    needs_grad = grad and (F.requires_grad or c1.kernel.requires_grad)
    if needs_grad and not training:
        return None                       # (gradients through running-statistics BatchNorm: the per-operator path)
    s = c1.stride
    ts_out = ts * s
    cm.stride(ts, s)
    n_in, n_out = cm.level(ts).n, cm.level(ts_out).n
    if n_in < 1 or n_out < 1:
        return None
    C, Cin = c1.out_channels, c1.in_channels
    tab = _Table()
    _opt_fields(tab, opts, training)
    keepalive = [F]
    keepalive += _conv_fields(tab, "c1_", c1, bns[0], cm, ts, opts, need_t, n_in, n_out)
    keepalive += _conv_fields(tab, "c2_", c2, bns[1], cm, ts_out, opts, needs_grad, n_out, n_out)
    if down:
        keepalive += _conv_fields(tab, "cd_", blk.downsample[0], bns[2], cm, ts, opts, need_t, n_in, n_out)
    lvl = cm.level(ts_out)
    ptr = cm.batch_ptr(ts_out)
    dp = blk.drop_path
    keep = dp.scale_vector(x) if isinstance(dp, MinkowskiDropPath) else None      # (or the step's pre-drawn vector)
    fc = blk.se.fc
    lin1, lin2 = fc[0].linear, fc[2].linear
    tab.set(n_in=n_in, n_out=n_out, B=cm.batch_size, coords=lvl.coords.data_ptr(), ptr=ptr.data_ptr(),
            act=ACT_IDS[blk.relu.act_name], stride=s, has_down=int(down), se_act=SE_ACT_IDS[fc[1].act_name],
            se_H=lin1.out_features, se_w1=lin1.weight.data_ptr(), se_b1=_ptr(lin1.bias), se_w2=lin2.weight.data_ptr(),
            se_b2=_ptr(lin2.bias), keep=_ptr(keep))
    keepalive += [lvl.coords, ptr, keep]
    lib = _lib.load()
    call = _BlockCall()
    call.tab, call.n_in, call.n_out, call.C, call.Cin, call.down = tab, n_in, n_out, C, Cin, down
    call.keepalive = keepalive
    call.saved_bytes = lib.agb_net_block_bytes(tab.buf, 0)
    call.fwd_bytes = lib.agb_net_block_bytes(tab.buf, 1)
    call.bwd_bytes = lib.agb_net_block_bytes(tab.buf, 2) if needs_grad else 0
    dsc = blk.downsample[0] if down else None
    H = lin1.out_features
    shapes = [tuple(c1.kernel.shape), (1, C), (C,), (C,), tuple(c2.kernel.shape), (1, C), (C,), (C,),
              tuple(dsc.kernel.shape) if down else (1,), (1, C), (C,), (C,), (H, C), (H,), (C, H), (C,)]
    call.layout = _block_layout(tuple(shapes))
    if wt is not None and needs_grad:      # (ensured for this step's weights by the caller: ResNetBase.forward_features)
        tab.setp("c1_", wt=wt.ptr(c1))
        tab.setp("c2_", wt=wt.ptr(c2))
        if down:
            tab.setp("cd_", wt=wt.ptr(dsc))
        call.keepalive.append(wt.flat)
    params = [c1.kernel, c1.bias, bns[0].weight, bns[0].bias, c2.kernel, c2.bias, bns[1].weight, bns[1].bias,
              dsc.kernel if down else None, dsc.bias if down else None, bns[2].weight if down else None,
              bns[2].bias if down else None, lin1.weight, lin1.bias, lin2.weight, lin2.bias]
    call.has = tuple(p is not None for p in params)
    out = SEBlockFunction.apply(F, call, *params)
    return ME.SparseTensor(out, coordinate_map_key=ME.CoordinateMapKey(ts_out), coordinate_manager=cm)


_LAYOUTS = {}


def _block_layout(shapes):
    lay = _LAYOUTS.get(shapes)
    if lay is None:
        # the weight gradients (accumulated into) and conv2's bias gradient (exactly zero in front of a training-mode
        # BatchNorm) start at zero: first in the buffer, one fill
        lay = _LAYOUTS[shapes] = _GradLayout(shapes, zero_first=(0, 4, 8, 5))
    return lay
