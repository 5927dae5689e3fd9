"""autograd binding of csrc/head.hip: the regression head (one ``nn.Linear(C, 1)`` per target on the pooled features —
reference models/instance/minkowski.py:16-26) together with the loss of models/instance/base.py:154-179 (targets standardised
with the train statistics; smooth-L1 / L2 / L1, mean reduction, summed; weighted by the mean task weight) as ONE launch
forward and one backward instead of ~27 small library kernels between the backbone's forward and backward pass."""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib

_V, _I = _lib.c_void_p, _lib.c_int
_lib.declare("agb_reg_head_fwd", [_V, _I, _I, _I, _I, _V, _V, _V, _V, _V, _V, _I, _V, _V, _V, _V, _V])
_lib.declare("agb_reg_head_bwd", [_V, _I, _I, _I, _I, _V, _V, _V, _V, _V, _V, _I, _V])
MAX_TARGETS = 8
LOSS_BITS = {F.smooth_l1_loss: 1, F.mse_loss: 2, F.l1_loss: 4}


def loss_mask(fns):
    """Bit mask of the configured loss functions, or None when one of them is not built in (or one is listed twice)."""
    mask = 0
    for fn in fns:
        bit = LOSS_BITS.get(fn)
        if bit is None or mask & bit:
            return None
        mask |= bit
    return mask or None


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class RegHeadLossFunction(torch.autograd.Function):
    """(out [B, T], loss_reg, loss) = head + loss of pooled [B, C]; params: weight_0, bias_0, weight_1, bias_1, ...
    Only ``loss`` carries a gradient (``out`` and ``loss_reg`` are reported values)."""

    @staticmethod
    def forward(ctx, pooled, y, center, scale, weights, mask, *params):
        pooled = pooled.contiguous()
        B, C = pooled.shape
        T = len(params) // 2
        ws, bs = list(params[0::2]), list(params[1::2])
        dev = pooled.device
        buf = torch.empty(2 * B * T + 2, dtype=torch.float32, device=dev)
        out, dout = buf[:B * T].view(B, T), buf[B * T:2 * B * T]
        loss_reg, loss = buf[2 * B * T], buf[2 * B * T + 1]
        _lib.call("agb_reg_head_fwd", pooled.data_ptr(), pooled.stride(0), B, C, T, _ptr_array(ws), _ptr_array(bs),
                  y.data_ptr(), center.data_ptr(), scale.data_ptr(), weights.data_ptr(), int(mask), out.data_ptr(),
                  dout.data_ptr(), loss_reg.data_ptr(), loss.data_ptr(), _lib.stream())
        ctx.save_for_backward(pooled, dout, *ws)
        ctx.has_bias = [b is not None for b in bs]
        ctx.mark_non_differentiable(out, loss_reg)
        return out, loss_reg, loss

    @staticmethod
    def backward(ctx, _gout, _greg, gloss):
        pooled, dout, *ws = ctx.saved_tensors
        B, C = pooled.shape
        T = len(ws)
        dev = pooled.device
        gloss = gloss.contiguous()
        flat = torch.empty(T * (C + 1), dtype=torch.float32, device=dev)
        dws = [flat[t * C:(t + 1) * C].view(1, C) for t in range(T)]
        dbs = [flat[T * C + t:T * C + t + 1] if ctx.has_bias[t] else None for t in range(T)]
        dpooled = torch.empty(B, C, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        _lib.call("agb_reg_head_bwd", pooled.data_ptr(), pooled.stride(0), B, C, T, _ptr_array(ws), dout.data_ptr(),
                  gloss.data_ptr(), _ptr_array(dws), _ptr_array(dbs), None if dpooled is None else dpooled.data_ptr(), C,
                  _lib.stream())
        grads = []
        for t in range(T):
            grads += [dws[t], dbs[t]]
        return (dpooled, None, None, None, None, None) + tuple(grads)


def reg_head_loss(pooled, linears, y, center, scale, weights, mask):
    """pooled [B, C] (device fp32); linears: the ``nn.Linear(C, 1)`` modules of the targets; y [B, T] raw targets;
    center / scale [1, T], weights [T] (the model's buffers).  Returns (out [B, T], loss_reg, loss)."""
    params = []
    for lin in linears:
        params += [lin.weight, lin.bias]
    return RegHeadLossFunction.apply(pooled, y, center, scale, weights, mask, *params)
