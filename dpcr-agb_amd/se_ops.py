"""autograd binding of csrc/se.hip: the squeeze-excite excitation MLP (Linear -> act -> Linear -> Sigmoid on the pooled
[B, C] features; reference torch_points3d/modules/MinkowskiEngine/senet_block.py:33-50) as one forward and two backward
launches instead of ~35 library launches on 32-row operands."""
import torch

from . import _lib

_P = _lib.ptr
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
        _lib.call("agb_segment_broadcast", _P(s), _P(coords), _P(ptr), _P(x), x.stride(0), _P(out), out.stride(0), n,
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
            _lib.call("agb_segment_scale_add", _P(s), _P(dp), _P(coords), _P(ptr), _P(dout), dout.stride(0), _P(dx),
                      dx.stride(0), n, C, _lib.stream())
        return dx, None, None, None, dw1, db1, dw2, db2, None


def se_layer(x, coords, ptr, B, lin1: torch.nn.Linear, act_name: str, lin2: torch.nn.Linear):
    return SELayerFunction.apply(x, coords, ptr, B, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ACT_IDS[act_name])


def se_excite(p, lin1: torch.nn.Linear, act_name: str, lin2: torch.nn.Linear):
    return SEExciteFunction.apply(p, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ACT_IDS[act_name])
