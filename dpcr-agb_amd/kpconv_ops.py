"""autograd bindings of csrc/kpconv.hip (KPConv neighbourhood gather, max-pooled shortcut)."""
import torch

from . import _lib

_P = _lib.ptr
_V, _I, _F = _lib.c_void_p, _lib.c_int, _lib.c_float
_lib.declare("agb_kpconv_gather_fwd", [_V, _V, _V, _I, _I, _V, _I, _V, _I, _F, _V, _I, _I, _V])
_lib.declare("agb_kpconv_gather_bwd", [_V, _V, _V, _I, _I, _V, _V, _I, _F, _V, _I, _I, _I, _V])
_lib.declare("agb_kp_maxpool_fwd", [_V, _I, _V, _I, _I, _V, _V, _I, _I, _V])
_lib.declare("agb_kp_maxpool_bwd", [_V, _V, _V, _I, _I, _I, _V])


_lib.declare("agb_kpconv_gather_fwd_csr", [_V, _V, _V, _V, _I, _I, _V, _I, _V, _I, _F, _V, _I, _I, _V])
_lib.declare("agb_kpconv_gather_bwd_csr", [_V, _V, _V, _V, _I, _I, _V, _V, _I, _F, _V, _I, _I, _I, _V])
_lib.declare("agb_kp_maxpool_fwd_csr", [_V, _I, _V, _V, _I, _V, _I, _V, _V, _I, _I, _V])


# the SURVEY 8(b) names: the whole layer as one call (csrc/aliases.hip); the autograd functions below drive the pieces
_lib.declare("agb_hash_build", [_V, _I, _V, _V, _V, _I, _V, _V, _V])
_lib.declare("agb_kpconv_fwd", [_V, _V, _V, _I, _I, _V, _I, _V, _I, _F, _V, _V, _V, _I, _I, _I, _I, _V])
_lib.declare("agb_kpconv_bwd_workspace_bytes", [_I, _I, _I, _I])
_lib.declare("agb_kpconv_bwd", [_V, _V, _V, _I, _I, _V, _V, _I, _V, _I, _F, _V, _V, _I, _V, _I, _I, _I, _V,
                                __import__("ctypes").c_size_t, _V])


# the whole layer as one kernel per direction (csrc/kpfused.hip)
_S = __import__("ctypes").c_size_t
_lib.declare("agb_kpconv_fused_supported", [_I, _I, _I])
_lib.declare("agb_kpconv_fused_bwd_workspace_bytes", [_I, _I, _I, _I])        # (size_t: _lib.size_call)
_lib.declare("agb_kpconv_fused_fwd", [_V, _V, _V, _I, _I, _V, _I, _V, _I, _F, _V, _V, _I, _I, _I, _V])
_lib.declare("agb_kpconv_fused_bwd", [_V, _V, _V, _I, _I, _V, _I, _V, _I, _F, _V, _V, _I, _V, _I, _V, _I, _V, _S, _I, _I, _V])


def is_ragged(idx):
    return hasattr(idx, "row_ptr")


def as_index(idx):
    """Neighbour matrices are int32 on the device; the reference hands int64 (kpconv.py:224-225) — converted once.  Ragged
    neighbour rows (kp_index.Neighbors) pass through: the kernels walk them as they are."""
    if is_ragged(idx):
        return idx
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    return idx.contiguous()


class KPGatherFunction(torch.autograd.Function):
    """wf[n,k,:] = sum_h infl(n,h,k) * x[idx[n,h],:]  (the neighbour-feature gather + kernel-weight correlation)."""

    @staticmethod
    def forward(ctx, x, q_pts, s_pts, idx, kernel_points, extent):
        x = x.contiguous()
        q_pts, s_pts, kernel_points = q_pts.contiguous(), s_pts.contiguous(), kernel_points.contiguous()
        N, H = idx.shape
        Ns, cin = x.shape
        K = kernel_points.shape[0]
        wf = _gather(x, q_pts, s_pts, idx, kernel_points, extent)
        if is_ragged(idx):
            ctx.ragged = idx
            ctx.save_for_backward(q_pts, s_pts, kernel_points)
        else:
            ctx.ragged = None
            ctx.save_for_backward(q_pts, s_pts, idx, kernel_points)
        ctx.cfg = (float(extent), Ns, cin)
        return wf

    @staticmethod
    def backward(ctx, dwf):
        extent, Ns, cin = ctx.cfg
        dwf = dwf.contiguous()
        dx = torch.zeros(Ns, cin, dtype=torch.float32, device=dwf.device)
        if ctx.ragged is not None:
            q_pts, s_pts, kernel_points = ctx.saved_tensors
            r, K = ctx.ragged, kernel_points.shape[0]
            _lib.CALL_NOTE = {"valid": int(r.indices.shape[0])}
            _lib.call("agb_kpconv_gather_bwd_csr", _P(q_pts), _P(s_pts), _P(r.row_ptr), _P(r.indices), r.limit, Ns, _P(dwf),
                      _P(kernel_points), K, extent, _P(dx), dx.stride(0), r.nq, cin, _lib.stream())
            return dx, None, None, None, None, None
        q_pts, s_pts, idx, kernel_points = ctx.saved_tensors
        N, H = idx.shape
        K = kernel_points.shape[0]
        _lib.call("agb_kpconv_gather_bwd", _P(q_pts), _P(s_pts), _P(idx), H, Ns, _P(dwf), _P(kernel_points), K, extent,
                  _P(dx), dx.stride(0), N, cin, _lib.stream())
        return dx, None, None, None, None, None


def _gather(x, q_pts, s_pts, idx, kernel_points, extent):
    N, H = idx.shape
    Ns, cin = x.shape
    K = kernel_points.shape[0]
    wf = torch.empty(N, K, cin, dtype=torch.float32, device=x.device)
    if is_ragged(idx):
        _lib.CALL_NOTE = {"valid": int(idx.indices.shape[0])}      # (for instrumented runs: the gather's pair count)
        _lib.call("agb_kpconv_gather_fwd_csr", _P(q_pts), _P(s_pts), _P(idx.row_ptr), _P(idx.indices), idx.limit, Ns, _P(x),
                  x.stride(0), _P(kernel_points), K, float(extent), _P(wf), N, cin, _lib.stream())
        return wf
    _lib.call("agb_kpconv_gather_fwd", _P(q_pts), _P(s_pts), _P(idx), H, Ns, _P(x), x.stride(0), _P(kernel_points), K,
              float(extent), _P(wf), N, cin, _lib.stream())
    return wf


class KPConvSymmetricFunction(torch.autograd.Function):
    """The whole rigid KPConv (gather + kernel-weight contraction, blocks.py:264-400) of a layer whose query and support
    sets are THE SAME points with a symmetric neighbour relation (j in N(n) <=> n in N(j): an uncropped radius search
    of a point set against itself — every non-strided block of the network).

    Backward without a scatter: dx[j] = sum_{n: j in N(n)} sum_k infl_k(s_j - q_n) W_k dy[n]; with N symmetric the sum
    runs over j's OWN neighbour row, and infl_k(s_j - q_n) = max(0, 1 - |(s_n - q_j) - (-kp_k)| / extent): the forward
    gather applied to dy with the kernel points mirrored, followed by the dense product with W_k^T.  The scatter form
    (agb_kpconv_gather_bwd) issues one 64-byte fp32 atomic per (row, neighbour, 16 channels); those execute at the memory
    side of the fabric at ~19 G requests/s and made the backward gather 2.6x the forward one."""

    @staticmethod
    def supported(K, cin, cout):
        from .sparse_ops import DenseConvFunction
        return DenseConvFunction.supported(K * cin, cout) and DenseConvFunction.supported(K * cout, cin)

    @staticmethod
    def forward(ctx, x, pts, idx, kernel_points, extent, weights):
        from .sparse_ops import current, dense_product
        x, pts, kernel_points = x.contiguous(), pts.contiguous(), kernel_points.contiguous()
        K, cin, cout = weights.shape
        ctx.opts = current()
        wf = _gather(x, pts, pts, idx, kernel_points, extent).view(-1, K * cin)
        out = dense_product(wf, weights.reshape(K * cin, cout), bn_stats=any(ctx.needs_input_grad), opts=ctx.opts)
        # (by the same symmetry dW[k,c,o] = sum_j x[j,c] wfd[j,k,o] with the mirrored gather of dy, which would let the
        # backward keep x instead of the 15x larger wf; measured 0.1 ms/step slower in the [N,16]^T [N,240] product shape)
        if is_ragged(idx):
            ctx.ragged = idx
            ctx.save_for_backward(wf, pts, kernel_points, weights)
        else:
            ctx.ragged = None
            ctx.save_for_backward(wf, pts, idx, kernel_points, weights)
        ctx.extent = float(extent)
        return out

    @staticmethod
    def backward(ctx, dy):
        from .sparse_ops import dense_product, dense_weight_grad
        if ctx.ragged is not None:
            (wf, pts, kernel_points, weights), idx = ctx.saved_tensors, ctx.ragged
        else:
            wf, pts, idx, kernel_points, weights = ctx.saved_tensors
        K, cin, cout = weights.shape
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[5]:
            dw = dense_weight_grad(wf, dy, ctx.opts).view(K, cin, cout)
        if ctx.needs_input_grad[0]:
            wfd = _gather(dy, pts, pts, idx, (-kernel_points).contiguous(), ctx.extent).view(-1, K * cout)
            dx = dense_product(wfd, weights.permute(0, 2, 1).reshape(K * cout, cin), "dgrad1x1", opts=ctx.opts)
        return dx, None, None, None, None, dw


class KPConvFusedFunction(torch.autograd.Function):
    """The same layer as KPConvSymmetricFunction (rigid KPConv on ONE point set with a symmetric, ragged neighbour relation,
    blocks.py:264-400) as ONE kernel per direction: csrc/kpfused.hip gathers a tile of 16 / 32 query rows' weighted
    neighbourhood features into LDS and contracts them with the kernel weights held in registers — wf[N, 15, Cin] is never
    written.  Backward: one launch gathers dy with mirrored kernel points and feeds both dx = wfd . W^T and
    dW[k, c, o] = sum_j x[j, c] wfd[j, k, o] (the symmetry again: the layer keeps x, not the 15x larger wf).  Fixed summation
    order in both directions (no atomics)."""

    @staticmethod
    def supported(K, cin, cout, idx, opts):
        return (opts.fused_kpconv and not opts.low_precision and is_ragged(idx)
                and bool(_lib.load().agb_kpconv_fused_supported(int(K), int(cin), int(cout))))

    @staticmethod
    def forward(ctx, x, pts, idx, kernel_points, extent, weights):
        x, pts, kernel_points, weights = x.contiguous(), pts.contiguous(), kernel_points.contiguous(), weights.contiguous()
        K, cin, cout = weights.shape
        N = x.shape[0]
        out = torch.empty(N, cout, dtype=torch.float32, device=x.device)
        _lib.CALL_NOTE = {"valid": int(idx.indices.shape[0]), "cin": cin, "cout": cout}
        _lib.call("agb_kpconv_fused_fwd", _P(pts), _P(idx.row_ptr), _P(idx.indices), idx.limit, N, _P(x), x.stride(0),
                  _P(kernel_points), K, float(extent), _P(weights), _P(out), out.stride(0), cin, cout, _lib.stream())
        ctx.ragged = idx
        ctx.save_for_backward(x, pts, kernel_points, weights)
        ctx.extent = float(extent)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, pts, kernel_points, weights = ctx.saved_tensors
        idx = ctx.ragged
        K, cin, cout = weights.shape
        N = x.shape[0]
        dy = dy.contiguous()
        want_dx, want_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[5]
        dx = torch.empty(N, cin, dtype=torch.float32, device=dy.device) if want_dx else None
        dw = ws = None
        nbytes = 0
        if want_dw:
            dw = torch.empty(K, cin, cout, dtype=torch.float32, device=dy.device)
            nbytes = _lib.size_call("agb_kpconv_fused_bwd_workspace_bytes", N, K, cin, cout)
            ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=dy.device)
        _lib.CALL_NOTE = {"valid": int(idx.indices.shape[0]), "cin": cin, "cout": cout, "dx": bool(want_dx), "dw": bool(want_dw)}
        _lib.call("agb_kpconv_fused_bwd", _P(pts), _P(idx.row_ptr), _P(idx.indices), idx.limit, N, _P(dy), dy.stride(0),
                  _P(kernel_points), K, ctx.extent, _P(weights), _P(x), x.stride(0), _P(dx), cin, _P(dw), 0, _P(ws), nbytes,
                  cin, cout, _lib.stream())
        return dx, None, None, None, None, dw


class KPMaxPoolFunction(torch.autograd.Function):
    """max over the neighbour matrix with a zero shadow row (blocks.py:98-114)."""

    @staticmethod
    def forward(ctx, x, idx):
        x = x.contiguous()
        N, H = idx.shape
        Ns, c = x.shape
        y = torch.empty(N, c, dtype=torch.float32, device=x.device)
        arg = torch.empty(N, c, dtype=torch.int32, device=x.device)
        if is_ragged(idx) and c % 4 == 0 and x.stride(0) % 4 == 0:
            _lib.call("agb_kp_maxpool_fwd_csr", _P(x), x.stride(0), _P(idx.row_ptr), _P(idx.indices), idx.limit,
                      _P(idx.max_count_dev), Ns, _P(y), _P(arg), N, c, _lib.stream())
        else:
            if is_ragged(idx):
                idx = idx.padded()
            _lib.call("agb_kp_maxpool_fwd", _P(x), x.stride(0), _P(idx), H, Ns, _P(y), _P(arg), N, c, _lib.stream())
        ctx.save_for_backward(arg)
        ctx.cfg = (Ns, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        Ns, c = ctx.cfg
        dy = dy.contiguous()
        dx = torch.zeros(Ns, c, dtype=torch.float32, device=dy.device)
        _lib.call("agb_kp_maxpool_bwd", _P(dy), _P(arg), _P(dx), dx.stride(0), dy.shape[0], c, _lib.stream())
        return dx, None
