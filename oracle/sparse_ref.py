"""ORACLE (test infrastructure only — never imported by the product path under dpcr-agb_amd/).

CPU restatement (torch-CPU / numpy, dictionary-based) of the sparse-voxel arithmetic that the reference
delegates to MinkowskiEngine, and of the reference's own network wiring on top of it.

PARITY UNPINNED for the MinkowskiEngine part: ME is a third-party dependency that is absent from
/root/reference (``pip install -U git+https://github.com/NVIDIA/MinkowskiEngine``, HEAD, unpinned —
/root/reference/README.md:100,107; latest release 0.5.4) and cannot be built here (needs CUDA).  The reference
holds no test or golden vector at this boundary.  This file therefore restates ME's documented semantics and is
anchored by known-answer tests in tests/test_oracle_sparse.py (k=1 conv == Linear; fully occupied grid ==
torch.nn.functional.conv3d / max_pool3d; fp64 gradcheck) and by the reference's call sites:
  torch_points3d/modules/MinkowskiEngine/SENet.py:47-70,113-118      (stem, max-pool, stages, global pool, head)
  torch_points3d/modules/MinkowskiEngine/resnet_block.py:48-75,95-133 (BasicBlock / Bottleneck wiring)
  torch_points3d/modules/MinkowskiEngine/senet_block.py:33-50,80-96,127-147 (SE layer and SE blocks)
  torch_points3d/modules/MinkowskiEngine/common.py:215-226,344-366    (ConvNormActivation, drop-path)
  torch_points3d/modules/MinkowskiEngine/PointNet.py:16-49            (MinkowskiPointNet)
  torch_points3d/models/instance/minkowski.py:17-29,67-89             (SeparateLinear head, SparseTensor input)

ME semantics restated here:
  * stride-s output coordinates = unique(floor(c / (s*ts)) * (s*ts)) (order: first occurrence — ME does not
    define an order; tests compare per coordinate);
  * kernel offsets of a hyper-cube kernel of size k: odd k -> {-(k//2)..k//2}, even k -> {0..k-1}, times
    tensor_stride*dilation, enumerated x fastest; kernel tensor [k^3, Cin, Cout]
    ([Cin, Cout] if k^3 == 1 and stride == 1); bias [1, Cout];
  * max pooling takes the max over PRESENT inputs only; global sum/avg/max pool per batch index;
    MinkowskiGlobalPooling == average; BatchNorm == nn.BatchNorm1d over all rows of the batch.
"""
import random

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- coordinates
def _pack(c: np.ndarray) -> np.ndarray:
    c = c.astype(np.int64)
    return (c[:, 0] << 48) | ((c[:, 3] + 32768) << 32) | ((c[:, 2] + 32768) << 16) | (c[:, 1] + 32768)


def floor_stride(coords: np.ndarray, ts_out: int) -> np.ndarray:
    """coords int [N,4] (b,x,y,z) -> unique strided coords, first-occurrence order."""
    c = np.asarray(coords, dtype=np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], ts_out) * ts_out
    _, first = np.unique(_pack(c), return_index=True)
    return c[np.sort(first)].reshape(-1, 4)


def kernel_offsets(K: int, step: int) -> np.ndarray:
    half = K // 2 if K % 2 == 1 else 0
    offs = []
    for iz in range(K):
        for iy in range(K):
            for ix in range(K):
                offs.append(((ix - half) * step, (iy - half) * step, (iz - half) * step))
    return np.array(offs, dtype=np.int64)


def kernel_map(coords_in: np.ndarray, coords_out: np.ndarray, K: int, step: int) -> np.ndarray:
    """nbr[k, r] = row of (coords_out[r] + offset_k) in coords_in, or -1 (sorted-key dictionary lookup)."""
    coords_in = np.asarray(coords_in, dtype=np.int64)
    coords_out = np.asarray(coords_out, dtype=np.int64)
    keys = _pack(coords_in)
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    offs = kernel_offsets(K, step)
    nbr = np.full((len(offs), len(coords_out)), -1, dtype=np.int64)
    if len(sk) == 0:
        return nbr
    for k in range(len(offs)):
        q = coords_out.copy()
        q[:, 1:] += offs[k]
        qk = _pack(q)
        pos = np.searchsorted(sk, qk)
        pos[pos >= len(sk)] = 0
        hit = sk[pos] == qk
        nbr[k] = np.where(hit, order[pos], -1)
    return nbr


class Coords:
    """Tiny coordinate manager: levels by tensor stride, cached kernel maps."""

    def __init__(self, coords: np.ndarray, batch_size: int = None):
        coords = np.asarray(coords, dtype=np.int64)
        self.levels = {1: coords}
        self.maps = {}
        self._pairs = {}
        self.cache_pairs = False
        self.B = int(coords[:, 0].max()) + 1 if batch_size is None else batch_size

    def level(self, ts, stride=1):
        ts_out = ts * stride
        if ts_out not in self.levels:
            self.levels[ts_out] = floor_stride(self.levels[ts], ts_out)
        return ts_out

    def map(self, ts, K, stride=1, dilation=1):
        key = (ts, K, stride, dilation)
        if key not in self.maps:
            ts_out = self.level(ts, stride)
            self.maps[key] = torch.from_numpy(kernel_map(self.levels[ts], self.levels[ts_out], K, ts * dilation))
        return self.maps[key]

    def pairs(self, ts, K, stride=1, dilation=1):
        """Per kernel offset: (output rows, input rows) of the present pairs — the index lists ``conv`` derives from the
        map, kept so that a Coords object reused across training steps (same batch, next epoch) does not rebuild them."""
        key = (ts, K, stride, dilation)
        if key not in self._pairs:
            nbr = self.map(ts, K, stride, dilation)
            out = []
            for k in range(nbr.shape[0]):
                rows = torch.nonzero(nbr[k] >= 0).squeeze(1)
                out.append((rows, nbr[k][rows]))
            self._pairs[key] = out
        return self._pairs[key]

    def batch_index(self, ts):
        return torch.from_numpy(self.levels[ts][:, 0].copy())


# --------------------------------------------------------------------------- ops (autograd-friendly)
def conv(feats, nbr, kernel, bias=None, pairs=None):
    """out[r] = bias + sum_k feats[nbr[k, r]] @ kernel[k].  pairs (optional): Coords.pairs(...) of the same map — the
    very index lists computed below, cached."""
    if kernel.dim() == 2:
        kernel = kernel.unsqueeze(0)
    n_out = nbr.shape[1]
    out = feats.new_zeros(n_out, kernel.shape[2])
    for k in range(nbr.shape[0]):
        if pairs is not None:
            rows, idx = pairs[k]
            if len(rows):
                out = out.index_add(0, rows, feats[idx] @ kernel[k])
            continue
        m = nbr[k] >= 0
        if m.any():
            rows = torch.nonzero(m).squeeze(1)
            out = out.index_add(0, rows, feats[nbr[k][rows]] @ kernel[k])
    if bias is not None:
        out = out + bias.reshape(1, -1)
    return out


def max_pool(feats, nbr):
    K3, n_out = nbr.shape
    pad = torch.cat([feats, feats.new_full((1, feats.shape[1]), float("-inf"))], 0)
    idx = torch.where(nbr >= 0, nbr, torch.full_like(nbr, feats.shape[0]))  # [K3, n_out]
    gathered = pad[idx]  # [K3, n_out, C]
    out = gathered.max(0).values
    return torch.where(torch.isinf(out), torch.zeros_like(out), out)


def global_pool(feats, batch_index, B, mode):
    C = feats.shape[1]
    if mode == "max":
        rows = [feats[batch_index == b].max(0).values if (batch_index == b).any() else feats.new_zeros(C)
                for b in range(B)]
        return torch.stack(rows)
    out = feats.new_zeros(B, C).index_add(0, batch_index, feats)
    if mode in ("avg", "mean"):
        cnt = torch.bincount(batch_index, minlength=B).clamp(min=1).to(feats.dtype).unsqueeze(1)
        out = out / cnt
    return out


def batch_norm(feats, sd, prefix, training, momentum, eps=1e-5, update=None):
    w, b = sd.get(prefix + ".bn.weight"), sd.get(prefix + ".bn.bias")
    rm, rv = sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"]
    if training:
        rm_, rv_ = rm.clone().to(feats.dtype), rv.clone().to(feats.dtype)
        out = F.batch_norm(feats, rm_, rv_, w, b, True, momentum, eps)
        if update is not None:
            update[prefix + ".bn.running_mean"] = rm_.detach()
            update[prefix + ".bn.running_var"] = rv_.detach()
        return out
    return F.batch_norm(feats, rm.to(feats.dtype), rv.to(feats.dtype), w, b, False, momentum, eps)


ACT = {"relu": F.relu, "gelu": F.gelu, "silu": F.silu, "sigmoid": torch.sigmoid, "tanh": torch.tanh,
       "elu": lambda x: F.elu(x, 0.54), "celu": lambda x: F.celu(x, 0.54)}


def drop_path(feats, batch_index, B, drop_prob, training):
    """common.py:344-366 — one random.uniform(0,1) per batch element, in batch order; scale kept by 1/keep."""
    if not training or drop_prob <= 0:
        return feats
    keep = torch.tensor([1.0 if random.uniform(0, 1) > drop_prob else 0.0 for _ in range(B)], dtype=feats.dtype)
    keep = keep / (1 - drop_prob)
    return feats * keep[batch_index].unsqueeze(1)


# --------------------------------------------------------------------------- networks
def _conv_module(x, ts, cm, sd, prefix, K, stride):
    kernel = sd[prefix + ".kernel"]
    bias = sd.get(prefix + ".bias")
    if kernel.dim() == 2:  # k=1, stride=1: plain matmul (ME use_mm)
        out = x @ kernel
        return (out + bias if bias is not None else out), ts
    nbr = cm.map(ts, K, stride)
    return conv(x, nbr, kernel, bias, cm.pairs(ts, K, stride) if cm.cache_pairs else None), ts * stride


def _block(x, ts, cm, sd, p, stride, act, training, momentum, dp, update):
    """Basic / Bottleneck / SE variants, told apart by the keys present (resnet_block.py, senet_block.py)."""
    B = cm.B
    residual, rts = x, ts
    if p + ".conv3.kernel" in sd:  # bottleneck
        out, ts1 = _conv_module(x, ts, cm, sd, p + ".conv1", 1, 1)
        out = act(batch_norm(out, sd, p + ".norm1", training, momentum, update=update))
        out, ts1 = _conv_module(out, ts1, cm, sd, p + ".conv2", 3, stride)
        out = act(batch_norm(out, sd, p + ".norm2", training, momentum, update=update))
        out, ts1 = _conv_module(out, ts1, cm, sd, p + ".conv3", 1, 1)
        out = batch_norm(out, sd, p + ".norm3", training, momentum, update=update)
    else:
        out, ts1 = _conv_module(x, ts, cm, sd, p + ".conv1", 3, stride)
        out = act(batch_norm(out, sd, p + ".norm1", training, momentum, update=update))
        out, ts1 = _conv_module(out, ts1, cm, sd, p + ".conv2", 3, 1)
        out = batch_norm(out, sd, p + ".norm2", training, momentum, update=update)
    bidx = cm.batch_index(ts1)
    if p + ".se.fc.0.linear.weight" in sd:
        y = global_pool(out, bidx, B, "avg")
        y = act(F.linear(y, sd[p + ".se.fc.0.linear.weight"], sd[p + ".se.fc.0.linear.bias"]))
        y = torch.sigmoid(F.linear(y, sd[p + ".se.fc.2.linear.weight"], sd[p + ".se.fc.2.linear.bias"]))
        out = out * y[bidx]
    if p + ".downsample.0.kernel" in sd:
        residual, _ = _conv_module(residual, rts, cm, sd, p + ".downsample.0", 1, stride)
        residual = batch_norm(residual, sd, p + ".downsample.1", training, momentum, update=update)
    out = drop_path(out, bidx, B, dp, training) + residual
    return act(out), ts1


def resnet_forward(sd, coords, feats, layers, strides=(1, 2, 2, 2), activation="gelu", first_stride=1,
                   global_pool_mode="sum", training=True, momentum=0.1, drop_path_prob=0.0, batch_size=None,
                   update=None, cm=None):
    """SENet.py:113-118 with the head of models/instance/minkowski.py:17-29 when ``final.linears.*`` exist.
    sd: state_dict (torch tensors, any float dtype). Returns [B, n_out].
    cm (optional): a Coords object of these very coordinates kept from an earlier call (a training loop that revisits its
    batches every epoch): levels, kernel maps and pair lists are reused instead of rebuilt — same arithmetic."""
    act = ACT[activation]
    if cm is None:
        cm = Coords(np.asarray(coords), batch_size)
    x, ts = feats, 1
    x, ts = _conv_module(x, ts, cm, sd, "blocks.0.0.conv", 7, first_stride)
    x = act(batch_norm(x, sd, "blocks.0.0.norm", training, momentum, update=update))
    x = max_pool(x, cm.map(ts, 3, 2))
    ts *= 2
    for s, (n_blocks, stride) in enumerate(zip(layers, strides), start=1):
        for i in range(n_blocks):
            x, ts = _block(x, ts, cm, sd, f"blocks.{s}.{i}", stride if i == 0 else 1, act, training, momentum,
                           drop_path_prob, update)
    pooled = global_pool(x, cm.batch_index(ts), cm.B, global_pool_mode)
    if "final.linear.weight" in sd:
        return F.linear(pooled, sd["final.linear.weight"], sd["final.linear.bias"])
    outs, i = [], 0
    while f"final.linears.{i}.weight" in sd:
        outs.append(F.linear(pooled, sd[f"final.linears.{i}.weight"], sd[f"final.linears.{i}.bias"]))
        i += 1
    return torch.cat(outs, 1)


def pointnet_forward(sd, batch_index, feats, B, activation="gelu", global_pool_mode="sum", training=True,
                     momentum=0.1, update=None, pool_rows=None, keep=None):
    """PointNet.py:43-49: blocks (3x Linear no-bias + BN + act) -> global pool -> mlp -> final.
    pool_rows (optional, long [B, C], max pooling only): take these rows as the winners instead of the arg-max (a test
    can pin the selection where two candidates tie within rounding).  keep (optional dict): receives the pre-pool
    activation under "embedding"."""
    act = ACT[activation]
    x = feats
    for lin, bn in ((0, 1), (3, 4), (6, 7)):
        x = F.linear(x, sd[f"blocks.{lin}.linear.weight"])
        x = act(batch_norm(x, sd, f"blocks.{bn}", training, momentum, update=update))
    if keep is not None:
        keep["embedding"] = x
    if pool_rows is not None:
        x = x[pool_rows, torch.arange(x.shape[1]).unsqueeze(0).expand_as(pool_rows)]
    else:
        x = global_pool(x, batch_index, B, global_pool_mode)
    for lin, bn in ((0, 1), (3, 4)):
        x = F.linear(x, sd[f"mlp.{lin}.linear.weight"])
        x = act(batch_norm(x, sd, f"mlp.{bn}", training, momentum, update=update))
    if "final.linear.weight" in sd:
        return F.linear(x, sd["final.linear.weight"], sd["final.linear.bias"])
    outs, i = [], 0
    while f"final.linears.{i}.weight" in sd:
        outs.append(F.linear(x, sd[f"final.linears.{i}.weight"], sd[f"final.linears.{i}.bias"]))
        i += 1
    return torch.cat(outs, 1)


def reg_loss(outputs, y, center, scale, weights):
    """models/instance/base.py:154-179 — smooth-L1 on standardised targets times mean(task weights)."""
    labels = (y - center) / scale
    return weights.mean() * F.smooth_l1_loss(outputs, labels)
