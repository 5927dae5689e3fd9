"""Known-answer tests that anchor the (parity-unpinned) sparse-voxel oracle: k=1 conv == Linear, a fully
occupied grid == dense conv3d / max_pool3d, strided coordinates, fp64 gradcheck."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sparse_ref as R


def _dense_grid(B, n):
    g = np.stack(np.meshgrid(np.arange(B), np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1)
    return g.reshape(-1, 4)  # (b, x, y, z), z fastest


def test_k1_conv_is_linear():
    torch.manual_seed(0)
    coords = _dense_grid(1, 3)
    x = torch.randn(len(coords), 5, dtype=torch.float64)
    w = torch.randn(1, 5, 7, dtype=torch.float64)
    cm = R.Coords(coords)
    out = R.conv(x, cm.map(1, 1), w, None)
    assert torch.allclose(out, x @ w[0])


def test_full_grid_matches_conv3d():
    torch.manual_seed(1)
    B, n, cin, cout, K = 2, 5, 3, 4, 3
    coords = _dense_grid(B, n)
    x = torch.randn(len(coords), cin, dtype=torch.float64)
    w = torch.randn(K ** 3, cin, cout, dtype=torch.float64)
    b = torch.randn(1, cout, dtype=torch.float64)
    cm = R.Coords(coords)
    out = R.conv(x, cm.map(1, K), w, b)
    # dense: input [B, C, X, Y, Z]; kernel offset k = ix + K*(iy + K*iz) -> weight[co, ci, ix, iy, iz]
    xd = x.view(B, n, n, n, cin).permute(0, 4, 1, 2, 3)
    wd = w.view(K, K, K, cin, cout).permute(4, 3, 2, 1, 0)  # [iz,iy,ix,ci,co] -> [co,ci,ix,iy,iz]
    ref = F.conv3d(xd, wd, b.view(-1), padding=K // 2).permute(0, 2, 3, 4, 1).reshape(-1, cout)
    assert torch.allclose(out, ref, atol=1e-10)


def test_full_grid_stride2_maxpool_matches_dense():
    torch.manual_seed(2)
    B, n, c = 1, 6, 3
    coords = _dense_grid(B, n)
    x = torch.randn(len(coords), c, dtype=torch.float64)
    cm = R.Coords(coords)
    out = R.max_pool(x, cm.map(1, 3, stride=2))
    oc = cm.levels[2]
    # output voxel at even coordinate c looks at c-1, c, c+1
    xd = x.view(n, n, n, c)
    for r, (b, X, Y, Z) in enumerate(oc.tolist()):
        sl = xd[max(X - 1, 0):X + 2, max(Y - 1, 0):Y + 2, max(Z - 1, 0):Z + 2].reshape(-1, c)
        assert torch.allclose(out[r], sl.max(0).values)
    assert len(oc) == (n // 2) ** 3


def test_floor_stride_negative_and_order():
    coords = np.array([[0, -1, 0, 3], [0, 1, 1, 2], [0, -2, 1, 3], [1, 0, 0, 0], [0, 0, 0, 2]])
    out = R.floor_stride(coords, 2)
    assert out.tolist() == [[0, -2, 0, 2], [0, 0, 0, 2], [1, 0, 0, 0]]


def test_even_kernel_offsets():
    offs = R.kernel_offsets(2, 1)
    assert offs.tolist()[0] == [0, 0, 0] and offs.tolist()[1] == [1, 0, 0] and offs.tolist()[-1] == [1, 1, 1]
    offs3 = R.kernel_offsets(3, 2)
    assert offs3.tolist()[0] == [-2, -2, -2] and offs3.tolist()[1] == [0, -2, -2] and offs3.tolist()[13] == [0, 0, 0]


def test_conv_gradcheck_fp64():
    torch.manual_seed(3)
    rng = np.random.default_rng(0)
    pts = np.unique(rng.integers(0, 4, size=(30, 3)), axis=0)
    coords = np.concatenate([np.zeros((len(pts), 1), dtype=np.int64), pts], 1)
    cm = R.Coords(coords)
    nbr = cm.map(1, 3, stride=2)
    x = torch.randn(len(coords), 2, dtype=torch.float64, requires_grad=True)
    w = torch.randn(27, 2, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: R.conv(a, nbr, b, None), (x, w))


def test_global_pool_modes():
    x = torch.tensor([[1.0, 2.0], [3.0, -1.0], [5.0, 0.0]])
    bi = torch.tensor([0, 0, 1])
    assert R.global_pool(x, bi, 2, "sum").tolist() == [[4.0, 1.0], [5.0, 0.0]]
    assert R.global_pool(x, bi, 2, "avg").tolist() == [[2.0, 0.5], [5.0, 0.0]]
    assert R.global_pool(x, bi, 2, "max").tolist() == [[3.0, 2.0], [5.0, 0.0]]
