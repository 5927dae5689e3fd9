"""An anchor for the HIP sparse path that does NOT go through the dictionary oracle: on a fully occupied grid a
generalized sparse convolution IS torch.nn.functional.conv3d with zero padding (cross-correlation, offsets
{-(k//2)..k//2}, x fastest), a strided one is conv3d with that stride on the even lattice, max pooling is
F.max_pool3d, and a SEBasicBlock (torch_points3d/modules/MinkowskiEngine/senet_block.py:53-96, resnet_block.py:48-75)
is its dense nn.Conv3d / BatchNorm3d twin.  MinkowskiEngine itself is absent: these identities are the
part of its documented semantics that plain PyTorch can state independently."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def full_grid(B, G, origin=0):
    """All cells of a G^3 cube per batch element, rows in a random order (batch-contiguous)."""
    rng = np.random.default_rng(G + B)
    rows = []
    for b in range(B):
        z, y, x = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
        c = np.stack([x.ravel(), y.ravel(), z.ravel()], 1) + origin
        rng.shuffle(c)
        rows.append(np.concatenate([np.full((len(c), 1), b), c], 1))
    return np.concatenate(rows).astype(np.int64)


def to_dense(feats, coords, B, G, origin=0, step=1):
    """rows [N, C] at coords (b, x, y, z) -> [B, C, Z, Y, X]"""
    C = feats.shape[1]
    d = feats.new_zeros(B, G, G, G, C)
    c = torch.as_tensor(coords)
    d[c[:, 0], (c[:, 3] - origin) // step, (c[:, 2] - origin) // step, (c[:, 1] - origin) // step] = feats
    return d.permute(0, 4, 1, 2, 3).contiguous()


def from_dense(d, coords, origin=0, step=1):
    c = torch.as_tensor(coords)
    return d.permute(0, 2, 3, 4, 1)[c[:, 0], (c[:, 3] - origin) // step, (c[:, 2] - origin) // step,
                                    (c[:, 1] - origin) // step]


def dense_weight(kernel, K):
    """ME kernel [K^3, Cin, Cout] (offset index x fastest) -> conv3d weight [Cout, Cin, kz, ky, kx]"""
    K3, cin, cout = kernel.shape
    return kernel.reshape(K, K, K, cin, cout).permute(4, 3, 0, 1, 2).contiguous()


@pytest.mark.parametrize("cin,cout,K,stride,G,origin", [(64, 64, 3, 1, 10, 0), (3, 64, 7, 1, 9, 0), (64, 128, 3, 2, 10, 0),
                                                         (16, 32, 3, 1, 8, -3), (32, 32, 2, 2, 8, 0),
                                                         (64, 96, 3, 2, 12, -4), (128, 128, 3, 1, 12, 0)])
def test_sparse_conv_equals_conv3d_on_full_grid(device, cin, cout, K, stride, G, origin):
    import dpcr_agb_amd.me_compat as ME
    B = 2
    torch.manual_seed(cin + cout + K)
    coords = full_grid(B, G, origin)
    x = torch.randn(len(coords), cin)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=K, stride=stride, bias=True, dimension=3).to(device)
    xg = x.to(device).requires_grad_(True)
    st = ME.SparseTensor(torch.zeros(len(coords), 1), coordinates=torch.from_numpy(coords).int(), device=device)
    out = conv(ME.SparseTensor(xg, coordinate_map_key=st.coordinate_map_key, coordinate_manager=st.coordinate_manager))
    oc = out.C.cpu().long()
    g = torch.randn(out.F.shape[0], cout)
    out.F.backward(g.to(device))

    xd = to_dense(x.double(), coords, B, G, origin).requires_grad_(True)
    w = conv.kernel.detach().cpu().double().requires_grad_(True)
    b = conv.bias.detach().cpu().double().requires_grad_(True)
    pad = K // 2 if K % 2 == 1 else 0
    yd = F.conv3d(xd, dense_weight(w, K), b.reshape(-1), stride=stride, padding=pad)
    ref = from_dense(yd, oc, origin, stride)
    assert ref.shape == out.F.shape
    ref.backward(g.double())
    assert rel_err(out.F, ref) < RTOL
    assert rel_err(xg.grad, from_dense(xd.grad, coords, origin)) < RTOL
    assert rel_err(conv.kernel.grad, w.grad) < RTOL
    assert rel_err(conv.bias.grad, b.grad) < RTOL


@pytest.mark.parametrize("K,stride,G", [(3, 2, 10), (2, 2, 8), (3, 1, 7)])
def test_sparse_maxpool_equals_max_pool3d_on_full_grid(device, K, stride, G):
    import dpcr_agb_amd.me_compat as ME
    B, C = 2, 64
    torch.manual_seed(K + G)
    coords = full_grid(B, G)
    x = torch.randn(len(coords), C)
    xg = x.to(device).requires_grad_(True)
    st = ME.SparseTensor(torch.zeros(len(coords), 1), coordinates=torch.from_numpy(coords).int(), device=device)
    out = ME.MinkowskiMaxPooling(K, stride, dimension=3)(
        ME.SparseTensor(xg, coordinate_map_key=st.coordinate_map_key, coordinate_manager=st.coordinate_manager))
    oc = out.C.cpu().long()
    g = torch.randn(*out.F.shape)
    out.F.backward(g.to(device))
    xd = to_dense(x.double(), coords, B, G).requires_grad_(True)
    yd = F.max_pool3d(xd, K, stride, padding=K // 2 if K % 2 == 1 else 0)
    ref = from_dense(yd, oc, 0, stride)
    ref.backward(g.double())
    assert rel_err(out.F, ref) < 1e-6
    assert rel_err(xg.grad, from_dense(xd.grad, coords)) < 1e-6


@pytest.mark.parametrize("stride", [1, 2])
def test_se_basic_block_equals_dense_twin(device, stride):
    """SEBasicBlock (conv3-BN-GELU-conv3-BN-SE, 1x1 strided downsample + BN, add, GELU) in training mode vs its dense
    conv3d / batch_norm twin in fp64."""
    import dpcr_agb_amd.me_compat as ME
    from dpcr_agb_amd.backbones.sparse import SENet14
    torch.manual_seed(5 + stride)
    B, G, cin, planes = 2, 8, 64, 128
    net = SENet14(3, 2, activation="gelu", first_stride=1, global_pool="sum", drop_path=0.0)
    blk = net.blocks[2][0]                     # SEBasicBlock(64 -> 128, stride 2, downsample)
    if stride == 1:
        from dpcr_agb_amd.backbones.sparse import SEBasicBlock
        from functools import partial
        import torch.nn as nn
        down = nn.Sequential(ME.MinkowskiConvolution(cin, planes, kernel_size=1, stride=1, dimension=3, bias=True),
                             ME.MinkowskiBatchNorm(planes))
        blk = SEBasicBlock(cin, planes, ME.MinkowskiGELU(), partial(ME.MinkowskiBatchNorm, momentum=0.1), stride=1,
                           downsample=down, dimension=3)
    for p in blk.parameters():                 # non-trivial BN affine parameters and biases
        if p.dim() == 1 or p.shape[0] == 1:
            torch.nn.init.normal_(p, 0.3 if p.dim() == 1 else 0.0, 0.2)
    sd = {k: v.detach().clone().double() for k, v in blk.state_dict().items()}
    blk.to(device).train()
    coords = full_grid(B, G)
    x = torch.randn(len(coords), cin)
    xg = x.to(device).requires_grad_(True)
    st = ME.SparseTensor(torch.zeros(len(coords), 1), coordinates=torch.from_numpy(coords).int(), device=device)
    out = blk(ME.SparseTensor(xg, coordinate_map_key=st.coordinate_map_key, coordinate_manager=st.coordinate_manager))
    oc = out.C.cpu().long()
    g = torch.randn(*out.F.shape)
    out.F.backward(g.to(device))

    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k)
         for k, v in sd.items()}

    def bn(t, prefix):
        return F.batch_norm(t, None, None, P[prefix + ".bn.weight"], P[prefix + ".bn.bias"], True, 0.1, 1e-5)

    xd = to_dense(x.double(), coords, B, G).requires_grad_(True)
    h = F.conv3d(xd, dense_weight(P["conv1.kernel"], 3), P["conv1.bias"].reshape(-1), stride=stride, padding=1)
    h = F.gelu(bn(h, "norm1"))
    h = bn(F.conv3d(h, dense_weight(P["conv2.kernel"], 3), P["conv2.bias"].reshape(-1), padding=1), "norm2")
    y = h.mean((2, 3, 4))
    y = F.gelu(F.linear(y, P["se.fc.0.linear.weight"], P["se.fc.0.linear.bias"]))
    y = torch.sigmoid(F.linear(y, P["se.fc.2.linear.weight"], P["se.fc.2.linear.bias"]))
    h = h * y[:, :, None, None, None]
    wd = P["downsample.0.kernel"]
    wd = wd.reshape(1, *wd.shape[-2:])
    r = bn(F.conv3d(xd, dense_weight(wd, 1), P["downsample.0.bias"].reshape(-1), stride=stride), "downsample.1")
    ref = from_dense(F.gelu(h + r), oc, 0, stride)
    ref.backward(g.double())
    assert rel_err(out.F, ref) < RTOL
    assert rel_err(xg.grad, from_dense(xd.grad, coords)) < RTOL
    gmax = max(float(v.grad.abs().max()) for v in P.values() if v.requires_grad)
    named = dict(blk.named_parameters())
    for k, v in P.items():
        if not v.requires_grad:
            continue
        got = named[k].grad.detach().cpu().double().reshape(v.grad.shape)
        e = float((got - v.grad).abs().max()) / max(float(v.grad.abs().max()), 1e-3 * gmax)
        assert e < RTOL, (k, e)
