"""autograd binding of csrc/se.hip: the squeeze-excite excitation MLP (Linear -> act -> Linear -> Sigmoid on the pooled
[B, C] features; reference torch_points3d/modules/MinkowskiEngine/senet_block.py:33-50) as one forward and two backward
launches instead of ~35 library launches on 32-row operands."""
import torch

from . import _lib

_P = _lib.ptr
_R, _sfx = _lib.rows, _lib.sfx
_V, _I = _lib.c_void_p, _lib.c_int
_lib.declare("agb_se_mlp_fwd", [_V, _V, _V, _V, _V, _I, _I, _I, _I, _V, _V, _V])
_lib.declare("agb_se_mlp_bwd", [_V, _V, _V, _I, _I, _I, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V])
ACT_IDS = {"none": 0, "relu": 1, "gelu": 2}
MAX_HIDDEN = 256


class SEExciteFunction(torch.autograd.Function):
    """s = sigmoid(W2 act(W1 p + b1) + b2) for p [B, C]; W1 [H, C], W2 [C, H] (nn.Linear layout)."""

    @staticmethod
    def forward(ctx, p, w1, b1, w2, b2, act_id):
        p, w1, w2 = p.contiguous(), w1.contiguous(), w2.contiguous()
        B, C = p.shape
        H = w1.shape[0]
        h_pre = torch.empty(B, H, dtype=torch.float32, device=p.device)
        s = torch.empty(B, C, dtype=torch.float32, device=p.device)
        _lib.call("agb_se_mlp_fwd", _P(p), _P(w1), _P(b1), _P(w2), _P(b2), B, C, H, act_id, _P(h_pre), _P(s),
                  _lib.stream())
        ctx.save_for_backward(p, w1, w2, h_pre, s)
        ctx.cfg = (act_id, b1 is not None, b2 is not None)
        return s

    @staticmethod
    def backward(ctx, ds):
        p, w1, w2, h_pre, s = ctx.saved_tensors
        act_id, has_b1, has_b2 = ctx.cfg
        ds = ds.contiguous()
        B, C = p.shape
        H = w1.shape[0]
        dev = p.device
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        dz2, dh, dp = f32((C + 511) // 512, B, C), f32(B, H), f32(B, C)
        dw1, dw2 = f32(H, C), f32(C, H)
        db1 = f32(H) if has_b1 else None
        db2 = f32(C) if has_b2 else None
        _lib.call("agb_se_mlp_bwd", _P(p), _P(w1), _P(w2), B, C, H, act_id, _P(h_pre), _P(s), _P(ds), _P(dz2), _P(dh),
                  _P(dp), _P(dw1), _P(db1), _P(dw2), _P(db2), _lib.stream())
        return dp, dw1, db1, dw2, db2, None


class SELayerFunction(torch.autograd.Function):
    """The whole squeeze-excite layer on the rows x [N, C] of a sparse tensor:
    out = x * sigmoid(W2 act(W1 avgpool(x) + b1) + b2)[batch].  Forward: segment average, excitation MLP, broadcast
    multiplication (3 launches).  Backward: ds = segment_sum(dout * x), MLP backward, and ONE pass for
    dx = dout * s[batch] + dpool[batch] / rows(batch) — the unfused graph needs two broadcast kernels plus the addition
    autograd inserts where x feeds both the pooling and the multiplication."""

    @staticmethod
    def forward(ctx, x, coords, ptr, B, w1, b1, w2, b2, act_id):
        from .sparse_ops import segment_reduce
        x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
        n, C = x.shape
        H = w1.shape[0]
        if C % 4 != 0:
            raise _lib.AgbError("fused squeeze-excite needs a channel count that is a multiple of 4")
        p, _ = segment_reduce(x, None, ptr, B, 1)
        h_pre = torch.empty(B, H, dtype=torch.float32, device=x.device)
        s = torch.empty(B, C, dtype=torch.float32, device=x.device)
        _lib.call("agb_se_mlp_fwd", _P(p), _P(w1), _P(b1), _P(w2), _P(b2), B, C, H, act_id, _P(h_pre), _P(s),
                  _lib.stream())
        out = torch.empty_like(x)
        _lib.call("agb_segment_broadcast" + _sfx(x), _P(s), _P(coords), _P(ptr), _R(x), x.stride(0), _R(out), out.stride(0), n,
                  C, 0, _lib.stream())
        ctx.save_for_backward(x, coords, ptr, p, w1, w2, h_pre, s)
        ctx.cfg = (act_id, b1 is not None, b2 is not None, B)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .sparse_ops import segment_reduce
        x, coords, ptr, p, w1, w2, h_pre, s = ctx.saved_tensors
        act_id, has_b1, has_b2, B = ctx.cfg
        dout = dout.contiguous()
        n, C = x.shape
        H = w1.shape[0]
        dev = x.device
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        ds, _ = segment_reduce(dout, x, ptr, B, 0)
        dz2, dh, dp = f32((C + 511) // 512, B, C), f32(B, H), f32(B, C)
        dw1, dw2 = f32(H, C), f32(C, H)
        db1 = f32(H) if has_b1 else None
        db2 = f32(C) if has_b2 else None
        _lib.call("agb_se_mlp_bwd", _P(p), _P(w1), _P(w2), B, C, H, act_id, _P(h_pre), _P(s), _P(ds), _P(dz2), _P(dh),
                  _P(dp), _P(dw1), _P(db1), _P(dw2), _P(db2), _lib.stream())
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("agb_segment_scale_add" + _sfx(dout), _P(s), _P(dp), _P(coords), _P(ptr), _R(dout), dout.stride(0), _R(dx),
                      dx.stride(0), n, C, _lib.stream())
        return dx, None, None, None, dw1, db1, dw2, db2, None


def se_layer(x, coords, ptr, B, lin1: torch.nn.Linear, act_name: str, lin2: torch.nn.Linear):
    return SELayerFunction.apply(x, coords, ptr, B, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ACT_IDS[act_name])


def se_excite(p, lin1: torch.nn.Linear, act_name: str, lin2: torch.nn.Linear):
    return SEExciteFunction.apply(p, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ACT_IDS[act_name])


# ---------------------------------------------------------------------------------------------- fused block tail
_F = _lib.c_float
_lib.declare("agb_se_tail_chunks", [_I, _I, _I])
_lib.declare("agb_se_tail_stats", [_V, _I, _V, _I, _I, _I, _F, _F, _I, _V, _V, _V, _V, _V, _V, _V], rows=True)
_lib.declare("agb_se_tail_pool", [_V, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _V])
_lib.declare("agb_se_tail_fwd", [_V, _I, _V, _I, _V, _V, _V, _V, _V, _V, _V, _I, _I, _I, _V, _I, _V], rows=True)
_lib.declare("agb_se_tail_bwd_sums", [_V, _I, _V, _I, _V, _I, _V, _I, _V, _V, _V, _V, _V, _V, _I, _I, _I, _V, _V], rows=True)
_lib.declare("agb_se_tail_bwd_ds", [_V, _V, _I, _V, _V, _V, _I, _I, _V, _V, _V, _V])
_lib.declare("agb_se_tail_bwd_fold", [_V, _V, _V, _V, _V, _V, _V, _V, _V, _I, _I, _V, _V, _V, _V])
_lib.declare("agb_se_tail_bwd_apply", [_V, _I, _V, _I, _V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _I, _I, _I, _I, _V,
                                       _I, _V, _I, _V], rows=True)
# (whether the blocks take this fused tail is sparse_ops.KernelOptions.fused_tail; tests compare both forms)


class SEBlockTailFunction(torch.autograd.Function):
    """y = act(BatchNorm(z) * s[plot] * keep[plot] + r),  s = sigmoid(W2 act_se(W1 avgpool_plot(BatchNorm(z)) + b1) + b2):
    everything behind the last convolution of an SE residual block (senet_block.py:83-96, 126-147 + resnet_block.py:70-73)
    as one autograd node over csrc/norm.hip k_tail_*: neither BatchNorm(z) nor its product with s is written to memory;
    four passes over [N, C] forward, eight backward (the separate kernels: 9 and 14)."""

    @staticmethod
    def forward(ctx, z, r, gamma, beta, running_mean, running_var, momentum, eps, training, counter, coords, ptr, B, w1,
                b1, w2, b2, se_act, keep, act_id):
        z, r, w1, w2 = z.contiguous(), r.contiguous(), w1.contiguous(), w2.contiguous()
        n, C = z.shape
        H = w1.shape[0]
        dev = z.device
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        stats = f32(2, C)
        part = f32(_lib.load().agb_se_tail_chunks(n, C, B) * 3 * C)
        sf = _sfx(z, r)
        _lib.call("agb_se_tail_stats" + sf, _R(z), z.stride(0), _P(ptr), n, C, B, float(eps), float(momentum),
                  int(bool(training)), _P(part), _P(stats[0]), _P(stats[1]), _P(running_mean), _P(running_var), _P(counter),
                  _lib.stream())
        zbar, p, h_pre, s = f32(B, C), f32(B, C), f32(B, H), f32(B, C)
        _lib.call("agb_se_tail_pool", _P(part), _P(ptr), n, B, C, _P(stats[0]), _P(stats[1]), _P(gamma), _P(beta), _P(zbar),
                  _P(p), _lib.stream())
        _lib.call("agb_se_mlp_fwd", _P(p), _P(w1), _P(b1), _P(w2), _P(b2), B, C, H, se_act, _P(h_pre), _P(s), _lib.stream())
        y = torch.empty_like(z)
        _lib.call("agb_se_tail_fwd" + sf, _R(z), z.stride(0), _R(r), r.stride(0), _P(coords), _P(stats[0]), _P(stats[1]),
                  _P(gamma), _P(beta), _P(s), _P(keep), act_id, n, C, _R(y), y.stride(0), _lib.stream())
        none = torch.empty(0)
        ctx.save_for_backward(z, r, stats, zbar, p, h_pre, s, coords, ptr, w1, w2,
                              gamma if gamma is not None else none, beta if beta is not None else none,
                              keep if keep is not None else none)
        ctx.cfg = (bool(training), B, se_act, act_id, gamma is not None, beta is not None, keep is not None,
                   b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, r, stats, zbar, p, h_pre, s, coords, ptr, w1, w2, gamma, beta, keep = ctx.saved_tensors
        training, B, se_act, act_id, has_g, has_b, has_keep, has_b1, has_b2 = ctx.cfg
        gamma, beta, keep = (gamma if has_g else None), (beta if has_b else None), (keep if has_keep else None)
        dy = dy.contiguous()
        n, C = z.shape
        H = w1.shape[0]
        dev = z.device
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        S = f32(2, B, C)
        bn = (_P(stats[0]), _P(stats[1]), _P(gamma), _P(beta), _P(s), _P(keep))
        common = (_P(coords),) + bn
        spart = f32(_lib.load().agb_se_tail_chunks(n, C, B) * 2 * C)
        sf = _sfx(z, r, dy)
        _lib.call("agb_se_tail_bwd_sums" + sf, _R(z), z.stride(0), _R(r), r.stride(0), _R(dy), dy.stride(0), _P(ptr), B, *bn,
                  act_id, n, C, _P(spart), _lib.stream())
        ds = f32(B, C)
        _lib.call("agb_se_tail_bwd_ds", _P(spart), _P(ptr), n, _P(gamma), _P(beta), _P(keep), B, C, _P(S[0]), _P(S[1]),
                  _P(ds), _lib.stream())
        dz2, dh, dp = f32((C + 511) // 512, B, C), f32(B, H), f32(B, C)
        dw1, dw2 = f32(H, C), f32(C, H)
        db1 = f32(H) if has_b1 else None
        db2 = f32(C) if has_b2 else None
        _lib.call("agb_se_mlp_bwd", _P(p), _P(w1), _P(w2), B, C, H, se_act, _P(h_pre), _P(s), _P(ds), _P(dz2), _P(dh),
                  _P(dp), _P(dw1), _P(db1), _P(dw2), _P(db2), _lib.stream())
        dte, dgb = f32(B, C), f32(2, C)
        _lib.call("agb_se_tail_bwd_fold", _P(S[0]), _P(S[1]), _P(zbar), _P(ptr), _P(dp), _P(s), _P(keep), _P(stats[0]),
                  _P(stats[1]), B, C, _P(dte), _P(dgb[0]), _P(dgb[1]), _lib.stream())
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(r) if ctx.needs_input_grad[1] else None
        _lib.call("agb_se_tail_bwd_apply" + sf, _R(z), z.stride(0), _R(r), r.stride(0), _R(dy), dy.stride(0), *common, _P(dte),
                  _P(dgb[0]), _P(dgb[1]), act_id, int(training), n, C, _R(dz), 0 if dz is None else dz.stride(0), _R(dr),
                  0 if dr is None else dr.stride(0), _lib.stream())
        if dz is not None:
            # column sums of dz = the bias gradient of the convolution that produced z: 0 with batch statistics (the
            # BatchNorm is blind to a constant), gamma * rstd * dbeta with the running ones (norm_ops.BatchNormActFunction)
            colsum = int(C) if training else (stats[1] * dgb[0] * (gamma if gamma is not None else 1.0))
            dz.agb_colsum = (colsum, dz._version)
        return (dz, dr, dgb[1] if has_g else None, dgb[0] if has_b else None, None, None, None, None, None, None, None, None,
                None, dw1, db1, dw2, db2, None, None, None)


def se_block_tail(z, r, bn: torch.nn.BatchNorm1d, coords, ptr, B, lin1, se_act_name, lin2, keep, act_name):
    """nn.BatchNorm1d semantics for `bn` (batch statistics + running update in training, running statistics in eval)."""
    rm, rv = bn.running_mean, bn.running_var
    use_batch_stats = bn.training or rm is None
    momentum, counter = 0.0, None
    if bn.training and rm is not None:
        if bn.momentum is not None:
            momentum, counter = bn.momentum, bn.num_batches_tracked
        else:
            bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
    return SEBlockTailFunction.apply(z, r, bn.weight, bn.bias, rm, rv, momentum, bn.eps, use_batch_stats, counter, coords,
                                     ptr, B, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ACT_IDS[se_act_name], keep,
                                     ACT_IDS[act_name])
