"""autograd bindings of csrc/norm.hip: training-mode BatchNorm over [N, C] rows fused with the following
activation, and the fused residual tail act(a * drop_path_scale[batch] + r)."""
import torch

from . import _lib

_P = _lib.ptr
ACT_IDS = {None: 0, "none": 0, "relu": 1, "gelu": 2}

_lib.declare("agb_bn_chunks", [_lib.c_int])
_lib.declare("agb_bn_stats", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_float, _lib.c_float,
                              _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p,
                              _lib.c_void_p])
_lib.declare("agb_bn_stats_tracked", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_float, _lib.c_float,
                                      _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p,
                                      _lib.c_void_p, _lib.c_void_p, _lib.c_void_p])
_lib.declare("agb_bn_act_fwd", [_lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p])
_lib.declare("agb_bn_act_bwd", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int,
                                _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_int,
                                _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                _lib.c_void_p])
_lib.declare("agb_bn_act_bwd_colsum", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int,
                                       _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_void_p, _lib.c_int,
                                       _lib.c_int, _lib.c_void_p, _lib.c_void_p, _lib.c_int, _lib.c_void_p,
                                       _lib.c_void_p, _lib.c_void_p, _lib.c_void_p])
_lib.declare("agb_add_act_fwd", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                 _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p])
_lib.declare("agb_add_act_bwd", [_lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_int, _lib.c_void_p, _lib.c_void_p,
                                 _lib.c_void_p, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_int, _lib.c_void_p,
                                 _lib.c_void_p, _lib.c_void_p])


def bn_chunks(n):
    return _lib.load().agb_bn_chunks(int(n))


class BatchNormActFunction(torch.autograd.Function):
    """y = act(gamma * (x - mean) * rstd + beta) over the rows of x [N, C] (C % 4 == 0)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, act_id, training, counter=None):
        x = x.contiguous()
        n, c = x.shape
        if c % 4 != 0:
            raise _lib.AgbError("fused batch norm needs a channel count that is a multiple of 4")
        dev = x.device
        stats = torch.empty(2, c, dtype=torch.float32, device=dev)
        part = torch.empty(bn_chunks(n) * 3 * c, dtype=torch.float32, device=dev) if training else None
        _lib.call("agb_bn_stats_tracked", _P(x), x.stride(0), n, c, float(eps), float(momentum), int(bool(training)),
                  _P(part), _P(stats[0]), _P(stats[1]), _P(running_mean), _P(running_var), _P(counter), _lib.stream())
        y = torch.empty_like(x)
        _lib.call("agb_bn_act_fwd", _P(x), x.stride(0), n, c, _P(stats[0]), _P(stats[1]), _P(gamma), _P(beta),
                  act_id, _P(y), y.stride(0), _lib.stream())
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
        _lib.call("agb_bn_act_bwd_colsum", _P(x), x.stride(0), _P(dy), dy.stride(0), n, c, _P(stats[0]), _P(stats[1]),
                  _P(gamma) if has_g else None, _P(beta) if has_b else None, act_id, int(training), _P(part), _P(dx),
                  0 if dx is None else dx.stride(0), _P(dgb[0]), _P(dgb[1]), _P(dgb[2]) if dx is not None else None,
                  _lib.stream())
        if dx is not None:
            # column sums of dx in closed form (0 with batch statistics, gamma * rstd * dbeta with running ones): the
            # convolution that produced x (its backward node receives this very tensor) takes them as its bias gradient
            # instead of reducing dx again
            # The hint is tied to the tensor's version: if autograd accumulates another consumer's gradient into this
            # buffer in place, the version moves and the convolution's backward recomputes the sum itself.
            dx.agb_colsum = (dgb[2], dx._version)
        return dx, (dgb[0] if has_g else None), (dgb[1] if has_b else None), None, None, None, None, None, None, None


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
    return BatchNormActFunction.apply(x, bn.weight, bn.bias, rm, rv, momentum, bn.eps, ACT_IDS[act], use_batch_stats,
                                      counter)


class AddActFunction(torch.autograd.Function):
    """y = act(a * scale[batch] + r); scale: float[B] or None (drop-path keep/(1-p) per batch element)."""

    @staticmethod
    def forward(ctx, a, r, scale, coords, act_id):
        a, r = a.contiguous(), r.contiguous()
        n, c = a.shape
        if c % 4 != 0:
            raise _lib.AgbError("fused residual tail needs a channel count that is a multiple of 4")
        y = torch.empty_like(a)
        _lib.call("agb_add_act_fwd", _P(a), a.stride(0), _P(r), r.stride(0), _P(scale),
                  _P(coords) if scale is not None else None, n, c, act_id, _P(y), y.stride(0), _lib.stream())
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
        _lib.call("agb_add_act_bwd", _P(a), a.stride(0), _P(r), r.stride(0), _P(scale) if has_s else None,
                  _P(coords) if has_s else None, _P(dy), dy.stride(0), n, c, act_id, _P(da), _P(dr), _lib.stream())
        return da, (da if shared else dr), None, None, None
