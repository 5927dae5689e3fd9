"""Empty and ragged inputs through the product path (HIP) against the oracles: zero rows, an EMPTY cloud in the middle of a
batch, single-point clouds.  The reference's own behaviour is the bar where it is defined (radius search: the compiled
reference returns a padded matrix for an empty batch element and its Python wrapper raises on an empty result,
cpp_neighbors/wrapper.cpp:201-205); where it is undefined (grid subsampling / voxelisation of a cloud without points read
uninitialised min / max corners) the product returns empty results instead of failing."""
import numpy as np
import pytest
import torch

from oracle import kpconv_index as K
from oracle import sparse_ref as R
from oracle import voxelize_ref as VR

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_zero_row_products(device):
    from dpcr_agb_amd import norm_ops, sparse_ops
    w = torch.randn(64, 32, device=device)
    assert tuple(sparse_ops.dense_product(torch.zeros(0, 64, device=device), w).shape) == (0, 32)
    dw = sparse_ops.dense_weight_grad(torch.zeros(0, 64, device=device), torch.zeros(0, 32, device=device))
    assert tuple(dw.shape) == (64, 32) and float(dw.abs().max()) == 0.0
    nbr = torch.full((27, 0), -1, dtype=torch.int32, device=device)
    y = sparse_ops.spconv_forward_raw(torch.zeros(0, 64, device=device), torch.randn(27 * 64, 64, device=device), nbr, 0, None,
                                      0, 27, 64, 64)
    assert tuple(y.shape) == (0, 64)
    bn = torch.nn.BatchNorm1d(32).to(device)
    assert tuple(norm_ops.batch_norm_act(torch.zeros(0, 32, device=device), bn, None).shape) == (0, 32)
    # one row: the biased variance is 0, the output is beta, the running variance takes the unbiased estimate's n/(n-1)
    # guard (torch raises for a single value per channel in training mode; the fused kernel normalises with var = 0)
    x1 = torch.randn(1, 32, device=device)
    y1 = norm_ops.batch_norm_act(x1, bn, None)
    assert torch.isfinite(y1).all() and float(y1.detach().abs().max()) < 1e-3
    pooled = sparse_ops.segment_reduce(torch.zeros(0, 16, device=device), None,
                                       torch.zeros(3, dtype=torch.int32, device=device), 2, 0)[0]
    assert tuple(pooled.shape) == (2, 16) and float(pooled.abs().max()) == 0.0
    torch.cuda.synchronize()


def test_voxelize_empty_clouds(device):
    from dpcr_agb_amd import voxelize
    c, k, nl, b = voxelize.voxelize_last(torch.zeros(0, 3), np.array([0, 0], dtype=np.int64), 0.1)
    assert tuple(c.shape) == (0, 3) and tuple(k.shape) == (0,) and nl.tolist() == [0, 0]
    g = torch.Generator().manual_seed(3)
    pos = torch.rand(300, 3, generator=g)
    lens = np.array([120, 0, 180], dtype=np.int64)
    perm = voxelize.draw_permutations(lens)
    c, k, nl, _ = voxelize.voxelize_last(pos, lens, 0.1, perm=perm)
    assert nl[1] == 0
    # the two non-empty clouds, each against the oracle on its own
    off_in, off_out = 0, 0
    for n_in, n_out in zip(lens, nl):
        if n_in:
            rc, rk = VR.grid_sampling_last(pos[off_in:off_in + n_in].numpy(), perm[off_in:off_in + n_in].numpy(), 0.1)
            assert n_out == len(rk)
            assert np.array_equal(c[off_out:off_out + n_out].cpu().numpy(), rc)
            assert np.array_equal(k[off_out:off_out + n_out].cpu().numpy() - off_in, rk)
        off_in, off_out = off_in + n_in, off_out + n_out


@pytest.mark.parametrize("ql,sl", [([25, 0, 15], [20, 0, 30]), ([40, 0], [20, 30]), ([0, 40], [50, 0]), ([1, 39], [1, 49])])
def test_neighbors_with_empty_batch_elements(device, ql, sl):
    from dpcr_agb_amd import kp_index
    rng = np.random.default_rng(0)
    q = rng.random((sum(ql), 3)).astype(np.float32)
    s = rng.random((sum(sl), 3)).astype(np.float32)
    ref = K.batch_neighbors(q, s, ql, sl, 0.3)
    if ref.shape[1] == 0:      # no neighbour anywhere: the reference's wrapper raises (wrapper.cpp:201-205)
        with pytest.raises(RuntimeError):
            kp_index.batch_neighbors(q, s, ql, sl, 0.3)
        return
    got = kp_index.batch_neighbors(q, s, ql, sl, 0.3)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_neighbors_of_an_empty_batch_raise_like_the_reference(device):
    from dpcr_agb_amd import kp_index
    s = np.random.default_rng(0).random((50, 3)).astype(np.float32)
    with pytest.raises(RuntimeError):
        kp_index.batch_neighbors(np.zeros((0, 3), np.float32), s, [0], [50], 0.2)
    with pytest.raises(RuntimeError):
        kp_index.batch_neighbors(s, np.zeros((0, 3), np.float32), [50], [0], 0.2)


def test_grid_subsampling_with_empty_clouds(device):
    from dpcr_agb_amd import kp_index
    rng = np.random.default_rng(1)
    P = rng.random((200, 3)).astype(np.float32)
    F = rng.random((200, 2)).astype(np.float32)
    got = kp_index.batch_grid_subsampling(P, [90, 0, 110], features=F, sampleDl=0.2, random_grid_orient=False)
    ref = K.batch_grid_subsampling(P, [90, 0, 110], features=F, sampleDl=0.2)
    assert np.asarray(got[1]).tolist() == np.asarray(ref[1]).tolist() and got[1][1] == 0
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2])
    e = kp_index.batch_grid_subsampling(np.zeros((0, 3), np.float32), [0, 0], sampleDl=0.2, random_grid_orient=False)
    assert e[0].shape == (0, 3) and np.asarray(e[1]).tolist() == [0, 0]


def _ragged_sparse_batch(sizes, seeds):
    from dpcr_agb_amd import synthetic
    parts = [synthetic.make_sparse_batch([s], n_points=n) for s, n in zip(seeds, sizes)]
    batch = torch.cat([torch.full_like(p.batch, b) for b, p in enumerate(parts)])
    cat = lambda name: torch.cat([getattr(p, name) for p in parts])  # noqa: E731
    return synthetic.PlotBatch(batch, cat("coords"), cat("x"), cat("pos"), cat("y_reg"),
                               torch.ones(len(parts), 2, dtype=torch.bool), len(parts))


def test_network_on_ragged_plots_matches_oracle(device):
    """SENet14 forward + backward on plots of 1500, 1 and 40 points (the one-point plot is a single voxel at every level)."""
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    torch.manual_seed(0)
    ds = synthetic.SyntheticDataset(feature_dimension=3, stat_seeds=range(10_000, 10_032))
    opt = Opt(MODEL_OPTIONS["SENet14"])
    opt["drop_path"] = 0.0
    model = MinkowskiBaselineModel(opt, "minkowski", ds)
    batch = _ragged_sparse_batch([1500, 1, 40], [0, 1, 2])
    assert int((batch.batch == 1).sum()) == 1
    sd32 = {k: v.detach().clone() for k, v in model.model.state_dict().items()}
    model.to(device).train()
    model.set_input(batch, device)
    model.forward()
    model.loss.backward()
    sd = {k: (v.double().requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in sd32.items()}
    coords = torch.cat([batch.batch[:, None], batch.coords.long()], 1).numpy()
    out = R.resnet_forward(sd, coords, batch.x.double(), (1, 1, 1, 1), batch_size=len(batch))
    loss = R.reg_loss(out, batch.y_reg.double(), model.reg_center_targets.cpu().double(),
                      model.reg_scale_targets.cpu().double(), model.reg_weights.cpu().double())
    loss.backward()
    assert rel_err(model.output, out) < RTOL
    gmax = max(float(sd[k].grad.abs().max()) for k, _ in model.model.named_parameters())
    worst, worst_name = 0.0, None
    for k, p in model.model.named_parameters():
        ref_g = sd[k].grad
        e = float((p.grad.detach().cpu().double() - ref_g).abs().max()) / max(float(ref_g.abs().max()), 1e-3 * gmax)
        if e > worst:
            worst, worst_name = e, k
    print(f"ragged SENet14: output rel err {rel_err(model.output, out):.2e}, worst gradient rel err {worst:.2e} ({worst_name})")
    assert worst < RTOL, (worst, worst_name)


def test_kpconv_pyramid_on_ragged_plots_matches_oracle(device):
    """KPConv input pyramid (subsampling + radius searches, 5 levels) for plots of 6144, 3 and 200 points."""
    from dpcr_agb_amd import kp_index, synthetic
    parts = [synthetic.make_point_batch([s], n_points=n) for s, n in zip([0, 1, 2], [2048, 3, 200])]
    pos = torch.cat([p.pos for p in parts]).numpy().astype(np.float32)
    lens = [len(p.pos) for p in parts]
    dl, r = 0.02 * 2.5 / 2.5, 0.05
    for level in range(3):
        got_nb = kp_index.batch_neighbors(pos, pos, lens, lens, r)
        ref_nb = K.batch_neighbors(pos, pos, lens, lens, r)
        assert got_nb.shape == ref_nb.shape and np.array_equal(got_nb, ref_nb), level
        dl2 = dl * 2
        got = kp_index.batch_grid_subsampling(pos, lens, sampleDl=dl2, random_grid_orient=False)
        ref = K.batch_grid_subsampling(pos, lens, sampleDl=dl2)
        assert np.asarray(got[1]).tolist() == np.asarray(ref[1]).tolist(), level
        assert np.array_equal(got[0], ref[0]), level
        pos, lens, dl, r = np.asarray(got[0]), [int(v) for v in np.asarray(got[1])], dl2, r * 2


def test_kpconv_model_on_ragged_plots_matches_oracle(device):
    """Full KPConv model (5-level pyramid with the reference configuration) on plots of 3000, 5 and 250 points: pyramid vs
    the pinned index oracle with the same grid orientations, network output vs the fp64 layer oracle, finite gradients."""
    from oracle import kpconv_ref as KR
    from dpcr_agb_amd import kp_index, synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import KPConvModel
    torch.manual_seed(0)
    np.random.seed(3)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016))
    opt = Opt(MODEL_OPTIONS["KPConv"])
    model = KPConvModel(opt, "kpconv", ds).to(device).train()
    parts = [synthetic.make_point_batch([s], n_points=n) for s, n in zip([40, 41, 42], [3000, 5, 250])]
    pos = torch.cat([p.pos for p in parts])
    x = torch.cat([p.x for p in parts])
    lens = np.array([len(p.pos) for p in parts], dtype=np.int64)
    rots = [kp_index.random_grid_rotations(3) for _ in range(4)]
    inp = model.prepare_inputs(pos, x, lens, device, rotations=rots)
    cfg = opt.config
    pts, ln, r = pos.numpy(), lens, cfg.first_subsampling_dl * cfg.conv_radius
    for lvl in range(5):
        assert np.array_equal(inp["points"][lvl].cpu().numpy(), pts), lvl
        assert np.array_equal(inp["neighbors"][lvl].cpu().numpy(), K.batch_neighbors(pts, pts, ln, ln, r)), lvl
        assert [int(v) for v in inp["lengths"][lvl]] == [int(v) for v in ln], lvl
        if lvl == 4:
            break
        rot = pts.copy()
        i0 = 0
        for bi, n in enumerate(ln):
            rot[i0:i0 + n] = np.sum(np.expand_dims(pts[i0:i0 + n], 2) * rots[lvl][bi], axis=1)
            i0 += n
        sp, sb = K.batch_grid_subsampling(rot, ln, sampleDl=2 * r / cfg.conv_radius, order="canonical")
        i0 = 0
        for bi, n in enumerate(sb):
            sp[i0:i0 + n] = np.sum(np.expand_dims(sp[i0:i0 + n], 2) * rots[lvl][bi].T, axis=1)
            i0 += n
        assert K.same_up_to_ties(inp["pools"][lvl].cpu().numpy(), K.batch_neighbors(sp, pts, sb, ln, r), sp, pts), lvl
        pts, ln, r = sp, sb.astype(np.int64), r * 2
    assert min(int(v) for v in inp["lengths"][4]) >= 1
    out = model.model(Opt(inp))
    sd = {k: v.detach().cpu().double() for k, v in model.model.state_dict().items()}
    ob = dict(features=x.double(), points=[p.cpu().double() for p in inp["points"]],
              neighbors=[n.cpu().long() for n in inp["neighbors"]], pools=[p.cpu().long() for p in inp["pools"]],
              lengths=[l.numpy() for l in inp["lengths"]])
    ocfg = dict(first_subsampling_dl=cfg.first_subsampling_dl, conv_radius=cfg.conv_radius, KP_extent=cfg.KP_extent,
                in_features_dim=3, first_features_dim=cfg.first_features_dim, architecture=list(cfg.architecture),
                batch_norm_momentum=cfg.batch_norm_momentum)
    ref = KR.kpcnn_forward(sd, ocfg, ob, training=True)
    assert tuple(out.shape) == (3, ref.shape[1])
    assert rel_err(out, ref) < RTOL
    out.backward(torch.randn_like(out))
    assert all(torch.isfinite(p.grad).all() for p in model.model.parameters() if p.grad is not None)


def test_tensors_beyond_two_giga_elements(device):
    """Maximum sizes: a [2.2 M, 1024] activation (2.25e9 elements, 9 GB) through the dense product, the fused BatchNorm +
    activation, per-plot pooling and their backward passes — every row offset is a 64-bit product.  Checked on row blocks
    at the start, across the 2^31-element boundary (row 2 097 152) and at the end."""
    from dpcr_agb_amd import norm_ops, sparse_ops
    if torch.cuda.get_device_properties(0).total_memory < 80 * 2**30:
        pytest.skip("needs ~45 GB of device memory")
    torch.manual_seed(0)
    n, cin, cout = 2_200_000, 16, 1024
    x = torch.randn(n, cin, device=device)
    w = (torch.randn(cout, cin, device=device) * 0.2)
    blocks = (0, 1_000_000, 2_097_152 - 8, n - 64)
    y = sparse_ops.dense_product(x, w.t().contiguous())
    for r0 in blocks:
        ref = x[r0:r0 + 64].double() @ w.double().t()
        assert rel_err(y[r0:r0 + 64], ref) < 1e-5, r0
    bn = torch.nn.BatchNorm1d(cout).to(device).train()
    z = norm_ops.batch_norm_act(y, bn, "relu")
    mean = torch.zeros(cout, dtype=torch.float64, device=device)
    sq = torch.zeros_like(mean)
    for r0 in range(0, n, 200_000):
        blk = y[r0:r0 + 200_000].double()
        mean += blk.sum(0)
        sq += (blk * blk).sum(0)
    mean /= n
    var = sq / n - mean * mean
    for r0 in blocks:
        ref = torch.relu((y[r0:r0 + 64].double() - mean) / torch.sqrt(var + bn.eps))
        assert rel_err(z[r0:r0 + 64], ref) < 1e-5, r0
    assert float((bn.running_mean.double() - 0.1 * mean).abs().max()) < 1e-6
    del y, z
    # backward: linear -> BatchNorm + ReLU -> sum pooling per plot, two plots of n/2 rows
    xg, wg = x.requires_grad_(True), w.requires_grad_(True)
    lin = sparse_ops.dense_linear(xg, wg)
    lin.retain_grad()
    z = norm_ops.batch_norm_act(lin, bn, "relu")
    ptr = torch.tensor([0, n // 2, n], dtype=torch.int32, device=device)
    pooled = sparse_ops.segment_reduce(z.detach(), None, ptr, 2, 0)[0]
    ref_pool = torch.stack([z[:n // 2].detach().double().sum(0), z[n // 2:].detach().double().sum(0)])
    assert rel_err(pooled, ref_pool) < 1e-5
    gz = torch.zeros(cout, device=device)
    gz[::7] = 1e-3
    z.backward(gz.expand(n, cout))
    dy = lin.grad
    # BatchNorm's backward removes the column mean of the gradient: sums over ALL rows vanish (a mis-addressed region
    # would not), and so do the column sums of dx = dy W
    scale = float(dy.abs().max())
    assert float(dy.double().sum(0).abs().max()) < 1e-4 * scale * np.sqrt(n)
    # dx = dy W on the row blocks, dW = dy^T x accumulated over every row in fp64
    for r0 in blocks:
        assert rel_err(xg.grad[r0:r0 + 64], dy[r0:r0 + 64].double() @ w.detach().double()) < 1e-5, r0
    dw = torch.zeros(cout, cin, dtype=torch.float64, device=device)
    for r0 in range(0, n, 200_000):
        dw += dy[r0:r0 + 200_000].double().t() @ x.detach()[r0:r0 + 200_000].double()
    assert rel_err(wg.grad, dw) < 1e-4


@pytest.mark.parametrize("prefetch", [False, True])
def test_network_in_hash_mode_matches_grid_mode(device, prefetch, monkeypatch):
    """Coordinate levels and kernel maps through the HASH tables (what a batch takes whose extent is too large for the dense
    lookup grids) give the same training step as the grid mode: loss, output and every gradient, at 6 000-point plots (the
    first level is large enough for interleaved / work-balanced tiles), with and without the side-stream input pipeline."""
    import dpcr_agb_amd.coords as C
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel

    def run(hash_mode):
        monkeypatch.setattr(C, "GRID_MAX_CELLS", 0 if hash_mode else 1 << 28)
        torch.manual_seed(0)
        ds = synthetic.SyntheticDataset(feature_dimension=3, stat_seeds=range(10_000, 10_032))
        opt = Opt(MODEL_OPTIONS["SENet14"])
        opt["drop_path"] = 0.0
        model = MinkowskiBaselineModel(opt, "minkowski", ds).to(device).train()
        batch = synthetic.make_sparse_batch([0, 1, 2, 3], n_points=6000)
        if prefetch:
            model.prefetch_input(batch, device)
        model.set_input(batch, device)
        modes = {lvl_ts: model.input.coordinate_manager.mode for lvl_ts in (1,)}
        model.forward()
        model.loss.backward()
        torch.cuda.synchronize()
        return (modes[1], float(model.loss.detach()), model.output.detach().clone(),
                {k: p.grad.detach().clone() for k, p in model.model.named_parameters() if p.grad is not None})

    mode_g, loss_g, out_g, grads_g = run(False)
    mode_h, loss_h, out_h, grads_h = run(True)
    assert (mode_g, mode_h) == ("grid", "hash")
    assert abs(loss_g - loss_h) < 1e-5 * max(1.0, abs(loss_g))
    assert rel_err(out_h, out_g) < 1e-5
    gmax = max(float(v.abs().max()) for v in grads_g.values())
    bad = {k: float((grads_h[k] - v).abs().max()) / max(float(v.abs().max()), 1e-3 * gmax) for k, v in grads_g.items()}
    bad = {k: e for k, e in bad.items() if e >= RTOL}
    assert not bad, bad
