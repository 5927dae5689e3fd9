"""autograd bindings of csrc/norm.hip: training-mode BatchNorm over [N, C] rows fused with the following
activation, and the fused residual tail act(a * drop_path_scale[batch] + r)."""
import torch

from . import _lib
from . import se_ops as _se_ops  # noqa: F401  (declares the agb_se_tail_* entry points used below)

_P = _lib.ptr
_R, _sfx = _lib.rows, _lib.sfx
ACT_IDS = {None: 0, "none": 0, "relu": 1, "gelu": 2}

_lib.declare("agb_bn_chunks", [_lib.c_int])
_lib.declare("agb_bn_stats", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_float, _lib.c_float,
                              _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p,
                              _lib.c_void_p], rows=True)
_lib.declare("agb_bn_stats_tracked", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_float, _lib.c_float,
                                      _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p,
                                      _lib.c_void_p, _lib.c_void_p, _lib.c_void_p], rows=True)
_lib.declare("agb_bn_act_fwd", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p], rows=True)
_lib.declare("agb_bn_act_bwd", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int,
                                _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int,
                                _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                _lib.c_void_p], rows=True)
_lib.declare("agb_bn_act_bwd_colsum", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int,
                                       _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int,
                                       _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_void_p,
                                       _lib.c_void_p, _lib.c_void_p, _lib.c_void_p], rows=True)
_lib.declare("agb_add_act_fwd", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                 _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p], rows=True)
_lib.declare("agb_add_act_bwd", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                 _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p,
                                 _lib.c_void_p, _lib.c_void_p], rows=True)


_lib.declare("agb_bn_stats_fold", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_float, _lib.c_float, _lib.c_void_p,
                                   _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p])


def _statistics(x, n, c, eps, momentum, training, running_mean, running_var, counter, hint):
    """mean / rstd [2, C]: batch statistics of x (from the partials the producing kernel left, when it did) or the
    running ones."""
    stats = torch.empty(2, c, dtype=torch.float32, device=x.device)
    if training and hint is not None:
        part, chunks = hint
        _lib.call("agb_bn_stats_fold", _P(part), chunks, c, float(eps), float(momentum), _P(stats[0]), _P(stats[1]),
                  _P(running_mean), _P(running_var), _P(counter), _lib.stream())
        return stats
    part = torch.empty(bn_chunks(n) * 3 * c, dtype=torch.float32, device=x.device) if training else None
    _lib.call("agb_bn_stats_tracked" + _sfx(x), _R(x), x.stride(0), n, c, float(eps), float(momentum), int(bool(training)),
              _P(part), _P(stats[0]), _P(stats[1]), _P(running_mean), _P(running_var), _P(counter), _lib.stream())
    return stats


_lib.declare("agb_bn_bwd_fold", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p])


def bn_chunks(n):
    return _lib.load().agb_bn_chunks(int(n))


class BatchNormActFunction(torch.autograd.Function):
    """y = act(gamma * (x - mean) * rstd + beta) over the rows of x [N, C] (C % 4 == 0)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, act_id, training, counter=None,
                hint=None):
        x = x.contiguous()
        n, c = x.shape
        if c % 4 != 0:
            raise _lib.AgbError("fused batch norm needs a channel count that is a multiple of 4")
        stats = _statistics(x, n, c, eps, momentum, training, running_mean, running_var, counter, hint)
        y = torch.empty_like(x)
        _lib.call("agb_bn_act_fwd" + _sfx(x), _R(x), x.stride(0), n, c, _P(stats[0]), _P(stats[1]), _P(gamma), _P(beta),
                  act_id, _R(y), y.stride(0), _lib.stream())
        ctx.save_for_backward(x, stats, gamma if gamma is not None else torch.empty(0),
                              beta if beta is not None else torch.empty(0))
        ctx.cfg = (act_id, bool(training), gamma is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma, beta = ctx.saved_tensors
        act_id, training, has_g, has_b = ctx.cfg
        dy = dy.contiguous()
        n, c = x.shape
        dev = x.device
        part = torch.empty(bn_chunks(n) * 2 * c, dtype=torch.float32, device=dev)
        dgb = torch.empty(3, c, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.call("agb_bn_act_bwd_colsum" + _sfx(x, dy), _R(x), x.stride(0), _R(dy), dy.stride(0), n, c, _P(stats[0]),
                  _P(stats[1]),
                  _P(gamma) if has_g else None, _P(beta) if has_b else None, act_id, int(training), _P(part), _R(dx),
                  0 if dx is None else dx.stride(0), _P(dgb[0]), _P(dgb[1]), _P(dgb[2]) if dx is not None else None,
                  _lib.stream())
        if dx is not None:
            # column sums of dx in closed form (0 with batch statistics, gamma * rstd * dbeta with running ones): the
            # convolution that produced x (its backward node receives this very tensor) takes them as its bias gradient
            # instead of reducing dx again
            # The hint is tied to the tensor's version: if autograd accumulates another consumer's gradient into this
            # buffer in place, the version moves and the convolution's backward recomputes the sum itself.
            dx.agb_colsum = (dgb[2], dx._version)
        return dx, (dgb[0] if has_g else None), (dgb[1] if has_b else None), None, None, None, None, None, None, None, \
            None


def batch_norm_act(x, bn: torch.nn.BatchNorm1d, act=None):
    """nn.BatchNorm1d semantics (batch statistics + running-stat update in training, running stats in eval)."""
    rm, rv = bn.running_mean, bn.running_var  # None when track_running_stats is off
    use_batch_stats = bn.training or rm is None
    momentum = 0.0
    counter = None
    if bn.training and rm is not None:
        if bn.momentum is not None:
            # the int64 num_batches_tracked counter is bumped by the statistics fold kernel (one launch less per layer)
            momentum, counter = bn.momentum, bn.num_batches_tracked
        else:   # cumulative moving average: the factor needs the count on the host
            bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
    # training + tracked: batch stats, running stats updated in the fold kernel; eval + tracked: running stats;
    # untracked: batch stats, nothing to update
    from .sparse_ops import bn_hint
    hint = bn_hint(x, x.shape[1]) if (use_batch_stats and x.dim() == 2) else None
    return BatchNormActFunction.apply(x, bn.weight, bn.bias, rm, rv, momentum, bn.eps, ACT_IDS[act], use_batch_stats,
                                      counter, hint)


class AddActFunction(torch.autograd.Function):
    """y = act(a * scale[batch] + r); scale: float[B] or None (drop-path keep/(1-p) per batch element)."""

    @staticmethod
    def forward(ctx, a, r, scale, coords, act_id):
        a, r = a.contiguous(), r.contiguous()
        n, c = a.shape
        if c % 4 != 0:
            raise _lib.AgbError("fused residual tail needs a channel count that is a multiple of 4")
        y = torch.empty_like(a)
        _lib.call("agb_add_act_fwd" + _sfx(a, r), _R(a), a.stride(0), _R(r), r.stride(0), _P(scale),
                  _P(coords) if scale is not None else None, n, c, act_id, _R(y), y.stride(0), _lib.stream())
        ctx.save_for_backward(a, r, scale if scale is not None else torch.empty(0), coords)
        ctx.cfg = (act_id, scale is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, r, scale, coords = ctx.saved_tensors
        act_id, has_s = ctx.cfg
        dy = dy.contiguous()
        n, c = a.shape
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        # without a drop-path scale both gradients are the same values: one buffer, handed to both inputs (nothing in
        # this package writes gradients in place, and autograd only accumulates in place into buffers it owns alone)
        shared = not has_s and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
        dr = torch.empty_like(r) if (ctx.needs_input_grad[1] and not shared) else None
        _lib.call("agb_add_act_bwd" + _sfx(a, r, dy), _R(a), a.stride(0), _R(r), r.stride(0), _P(scale) if has_s else None,
                  _P(coords) if has_s else None, _R(dy), dy.stride(0), n, c, act_id, _R(da), _R(dr), _lib.stream())
        return da, (da if shared else dr), None, None, None


class BatchNormAddActFunction(torch.autograd.Function):
    """y = act(BatchNorm(z) + r): the tail of a residual block whose last layer is Linear / conv -> BatchNorm (KPConv
    blocks.py:640-668) without the BatchNorm output in memory: 2 passes + statistics forward (the statistics come from the
    producing product's epilogue where it left them), 8 passes backward instead of 10."""

    @staticmethod
    def forward(ctx, z, r, gamma, beta, running_mean, running_var, momentum, eps, act_id, training, counter, hint):
        z, r = z.contiguous(), r.contiguous()
        n, c = z.shape
        if c % 4 != 0:
            raise _lib.AgbError("fused batch norm + residual needs a channel count that is a multiple of 4")
        stats = _statistics(z, n, c, eps, momentum, training, running_mean, running_var, counter, hint)
        y = torch.empty_like(z)
        _lib.call("agb_se_tail_fwd" + _sfx(z, r), _R(z), z.stride(0), _R(r), r.stride(0), None, _P(stats[0]), _P(stats[1]),
                  _P(gamma), _P(beta), None, None, act_id, n, c, _R(y), y.stride(0), _lib.stream())
        none = torch.empty(0)
        ctx.save_for_backward(z, r, stats, gamma if gamma is not None else none, beta if beta is not None else none)
        ctx.cfg = (act_id, bool(training), gamma is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, r, stats, gamma, beta = ctx.saved_tensors
        act_id, training, has_g, has_b = ctx.cfg
        gamma, beta = (gamma if has_g else None), (beta if has_b else None)
        dy = dy.contiguous()
        n, c = z.shape
        dev = z.device
        chunks = _lib.load().agb_se_tail_chunks(n, c, 0)
        spart = torch.empty(chunks * 2 * c, dtype=torch.float32, device=dev)
        bn = (_P(stats[0]), _P(stats[1]), _P(gamma), _P(beta), None, None)
        sf = _sfx(z, r, dy)
        _lib.call("agb_se_tail_bwd_sums" + sf, _R(z), z.stride(0), _R(r), r.stride(0), _R(dy), dy.stride(0), None, 0, *bn,
                  act_id, n, c, _P(spart), _lib.stream())
        dgb = torch.empty(2, c, dtype=torch.float32, device=dev)
        _lib.call("agb_bn_bwd_fold", _P(spart), chunks, c, _P(dgb[0]), _P(dgb[1]), _lib.stream())
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(r) if ctx.needs_input_grad[1] else None
        _lib.call("agb_se_tail_bwd_apply" + sf, _R(z), z.stride(0), _R(r), r.stride(0), _R(dy), dy.stride(0), None, *bn, None,
                  _P(dgb[0]), _P(dgb[1]), act_id, int(training), n, c, _R(dz), 0 if dz is None else dz.stride(0), _R(dr),
                  0 if dr is None else dr.stride(0), _lib.stream())
        if dz is not None:
            colsum = int(c) if training else (stats[1] * dgb[0] * (gamma if gamma is not None else 1.0))
            dz.agb_colsum = (colsum, dz._version)
        return dz, dr, (dgb[1] if has_g else None), (dgb[0] if has_b else None), None, None, None, None, None, None, None, \
            None


def batch_norm_add_act(z, r, bn: torch.nn.BatchNorm1d, act):
    """act(bn(z) + r) with nn.BatchNorm1d semantics for `bn` (as batch_norm_act)."""
    rm, rv = bn.running_mean, bn.running_var
    use_batch_stats = bn.training or rm is None
    momentum, counter = 0.0, None
    if bn.training and rm is not None:
        if bn.momentum is not None:
            momentum, counter = bn.momentum, bn.num_batches_tracked
        else:
            bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
    from .sparse_ops import bn_hint
    hint = bn_hint(z, z.shape[1]) if (use_batch_stats and z.dim() == 2) else None
    return BatchNormAddActFunction.apply(z, r, bn.weight, bn.bias, rm, rv, momentum, bn.eps, ACT_IDS[act], use_batch_stats,
                                         counter, hint)


_V, _I = _lib.c_void_p, _lib.c_int
_lib.declare("agb_pointnet_pool_splits", [_I, _I])
_lib.declare("agb_pointnet_pool_fwd", [_V, _I, _I, _I, _V, _V, _V, _V, _I, _V, _I, _I, _I, _V, _V, _V, _V, _V])
_lib.declare("agb_pointnet_pool_bwd", [_V, _I, _I, _I, _V, _V, _I, _V, _V, _I, _V, _V, _V, _V, _I, _I, _V, _V, _I, _V, _V,
                                       _V])
_lib.declare("agb_pointnet_pool_fwd_aux", [_V, _I, _I, _I, _V, _V, _V, _V, _I, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _V])
_lib.declare("agb_pointnet_pool_bwd_aux", [_V, _I, _I, _I, _V, _V, _I, _V, _V, _I, _V, _V, _V, _V, _I, _I, _V, _V, _V, _I, _V,
                                           _V, _V])
POOL_MODES = {"sum": 0, "avg": 1, "mean": 1, "max": 2}


class BatchNormActPoolFunction(torch.autograd.Function):
    """pooled[b] = reduce_{rows of plot b} act(batchnorm(z))  without materialising the [N, C] activation, and its
    backward without materialising the broadcast pooled gradient (csrc/pointnet.hip): the tail of MinkowskiPointNet's
    shared MLP (PointNet.py:24-29)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, momentum, eps, act_id, training, counter, coords, ptr,
                B, mode, hint=None):
        z = z.contiguous()
        n, c = z.shape
        if c % 4 != 0:
            raise _lib.AgbError("fused batch norm + pooling needs a channel count that is a multiple of 4")
        dev = z.device
        stats = _statistics(z, n, c, eps, momentum, training, running_mean, running_var, counter, hint)
        sp = _lib.load().agb_pointnet_pool_splits(n, B)
        pooled = torch.empty(B, c, dtype=torch.float32, device=dev)
        arg = torch.empty(B, c, dtype=torch.int32, device=dev) if mode == 2 else None
        ppart = torch.empty(B * sp, c, dtype=torch.float32, device=dev) if sp > 1 else None
        parg = torch.empty(B * sp, c, dtype=torch.int32, device=dev) if (sp > 1 and mode == 2) else None
        # sum / avg pooling: the per-plot sums of act'(.) and act'(.) * zhat ride along (the backward's parameter gradients
        # then need no pass over z: 4.2 GB for the 1024-wide layer)
        aux = torch.empty(2, B, c, dtype=torch.float32, device=dev) if mode != 2 else None
        aux_part = torch.empty(2, B * sp, c, dtype=torch.float32, device=dev) if (mode != 2 and sp > 1) else None
        _lib.call("agb_pointnet_pool_fwd_aux", _P(z), z.stride(0), n, c, _P(stats[0]), _P(stats[1]), _P(gamma), _P(beta),
                  act_id, _P(ptr), B, mode, sp, _P(ppart), _P(parg), _P(pooled), _P(arg), _P(aux_part), _P(aux),
                  _lib.stream())
        ctx.save_for_backward(z, stats, gamma if gamma is not None else torch.empty(0),
                              beta if beta is not None else torch.empty(0), coords, ptr,
                              arg if arg is not None else torch.empty(0), aux if aux is not None else torch.empty(0))
        ctx.cfg = (act_id, bool(training), gamma is not None, beta is not None, B, mode)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        z, stats, gamma, beta, coords, ptr, arg, aux = ctx.saved_tensors
        act_id, training, has_g, has_b, B, mode = ctx.cfg
        dpooled = dpooled.contiguous()
        n, c = z.shape
        dev = z.device
        dgb = torch.empty(2, c, dtype=torch.float32, device=dev)
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        _lib.call("agb_pointnet_pool_bwd_aux", _P(z), z.stride(0), n, c, _P(coords), _P(ptr), B, _P(dpooled),
                  _P(arg) if mode == 2 else None, mode, _P(stats[0]), _P(stats[1]), _P(gamma) if has_g else None,
                  _P(beta) if has_b else None, act_id, int(training), None, _P(aux) if mode != 2 else None, _P(dz),
                  0 if dz is None else dz.stride(0), _P(dgb[0]), _P(dgb[1]), _lib.stream())
        return (dz, dgb[0] if has_g else None, dgb[1] if has_b else None) + (None,) * 12


def batch_norm_act_pool(z, bn: torch.nn.BatchNorm1d, act, coords, ptr, B, mode):
    """nn.BatchNorm1d semantics (as batch_norm_act) + activation + per-plot pooling ("sum" / "avg" / "max")."""
    rm, rv = bn.running_mean, bn.running_var
    use_batch_stats = bn.training or rm is None
    momentum, counter = 0.0, None
    if bn.training and rm is not None:
        if bn.momentum is not None:
            momentum, counter = bn.momentum, bn.num_batches_tracked
        else:
            bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
    from .sparse_ops import bn_hint
    hint = bn_hint(z, z.shape[1]) if (use_batch_stats and z.dim() == 2) else None
    return BatchNormActPoolFunction.apply(z, bn.weight, bn.bias, rm, rv, momentum, bn.eps, ACT_IDS[act], use_batch_stats,
                                          counter, coords, ptr, B, POOL_MODES[mode], hint)


_lib.declare("agb_pointnet_mlp_workspace_bytes", [_I] * 5)
_lib.declare("agb_pointnet_mlp_fwd", [_V, _I, _I, _I, _V, _V, _I, _V, _V, _I, _V, _V, _I, _I, _lib.c_float, _lib.c_float, _I, _V,
                                      _I, _I, _V, _V, _V, _V])


def pointnet_mlp_forward(x, layers, act, ptr, B, mode, training=False, momentum=0.0):
    """The whole shared MLP in ONE library call (agb_pointnet_mlp_fwd; inference path of MinkowskiPointNet):
    x [n, cin]; layers: three (nn.Linear without bias, nn.BatchNorm1d); returns pooled [B, c3] (and argmax for "max")."""
    import ctypes
    import torch.nn.functional as F
    n, cin = x.shape
    cin_p = max(12, (cin + 3) // 4 * 4)
    xp = (x if cin_p == cin else F.pad(x, (0, cin_p - cin))).contiguous()
    ws_, bns, keep = [], [], []
    prev = cin_p
    for lin, bn in layers:
        w = lin.weight.detach().t()                                   # [in, out]
        if w.shape[0] != prev:
            w = F.pad(w, (0, 0, 0, prev - w.shape[0]))
        w = w.contiguous()
        keep.append(w)
        ws_.append(w)
        tab = (ctypes.c_void_p * 4)(_P(bn.weight), _P(bn.bias), _P(bn.running_mean), _P(bn.running_var))
        bns.append(tab)
        prev = w.shape[1]
    c1, c2, c3 = (w.shape[1] for w in ws_)
    dev = x.device
    work = torch.empty(_lib.size_call("agb_pointnet_mlp_workspace_bytes", n, B, c1, c2, c3), dtype=torch.uint8, device=dev)
    pooled = torch.empty(B, c3, dtype=torch.float32, device=dev)
    m = POOL_MODES[mode]
    arg = torch.empty(B, c3, dtype=torch.int32, device=dev) if m == 2 else None
    _lib.call("agb_pointnet_mlp_fwd", _P(xp), xp.stride(0), n, cin_p, _P(ws_[0]), bns[0], c1, _P(ws_[1]), bns[1], c2,
              _P(ws_[2]), bns[2], c3, ACT_IDS[act], float(layers[0][1].eps), float(momentum), int(bool(training)), _P(ptr),
              B, m, _P(work), _P(pooled), _P(arg), _lib.stream())
    return pooled, arg


# ----------------------------------------------------------------------------------------------------- LayerNorm
_lib.declare("agb_layernorm_chunks", [_lib.c_int])
_lib.declare("agb_layernorm_fwd", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                   _lib.c_float, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p])
_lib.declare("agb_layernorm_bwd", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int,
                                   _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                   _lib.c_void_p, _lib.c_void_p])


class LayerNormFunction(torch.autograd.Function):
    """nn.LayerNorm(C) over the rows of x [N, C] (the reference's MinkowskiLayerNorm, common.py:369-386)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        n, c = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(n, 2, dtype=torch.float32, device=x.device)
        _lib.call("agb_layernorm_fwd", _P(x), x.stride(0), n, c, _P(gamma), _P(beta), float(eps), _P(y), y.stride(0),
                  _P(stats), _lib.stream())
        ctx.save_for_backward(x, stats, gamma if gamma is not None else torch.empty(0))
        ctx.has = (gamma is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma = ctx.saved_tensors
        has_g, has_b = ctx.has
        dy = dy.contiguous()
        n, c = x.shape
        chunks = _lib.load().agb_layernorm_chunks(int(n))
        part = torch.empty(chunks, 2, c, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dg = torch.empty(c, dtype=torch.float32, device=x.device) if has_g else None
        db = torch.empty(c, dtype=torch.float32, device=x.device) if has_b else None
        _lib.call("agb_layernorm_bwd", _P(x), x.stride(0), _P(dy), dy.stride(0), n, c, _P(gamma) if has_g else None,
                  _P(stats), _P(dx), c if dx is None else dx.stride(0), _P(part), _P(dg), _P(db), _lib.stream())
        return dx, dg, db, None


def layer_norm(x, ln: torch.nn.LayerNorm):
    if len(ln.normalized_shape) != 1 or ln.normalized_shape[0] != x.shape[1]:
        raise ValueError("layer_norm: the module normalises one channel axis of the row matrix")
    return LayerNormFunction.apply(x, ln.weight, ln.bias, ln.eps)
