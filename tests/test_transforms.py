"""NFI sparse transform chain: the CPU restatement (oracle/transforms_ref.py) against known answers and against the real
third-party calls the reference makes (matplotlib Path.contains_points, torch PairwiseDistance); the device pipeline
(dpcr_agb_amd.transforms -> agb_plot_prepare / agb_voxelize_last / agb_coords_augment) against the oracle, bit-exact
for positions, kept points, voxel coordinates and flips."""
import random

import numpy as np
import pytest
import torch

from oracle import transforms_ref as T
from oracle import voxelize_ref as V


def point_in_polygon_np(x, y, poly):
    """numpy copy of csrc/transform.hip:point_in_polygon (matplotlib's crossing test), vectorised over points."""
    poly = np.asarray(poly, dtype=np.float64)
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    vx0, vy0 = poly[-1]
    yflag0 = vy0 >= y
    inside = np.zeros(len(x), dtype=bool)
    for vx1, vy1 in poly:
        yflag1 = vy1 >= y
        cross = ((vy1 - y) * (vx0 - vx1) >= (vx1 - x) * (vy0 - vy1)) == yflag1
        inside ^= (yflag0 != yflag1) & cross
        yflag0, vx0, vy0 = yflag1, vx1, vy1
    return inside


def raw_plot(seed, n):
    """A synthetic plot in the raw frame of the reference (metres, centred on the plot centre)."""
    from dpcr_agb_amd import synthetic
    rng = np.random.default_rng(seed)
    pos, _, y = synthetic.make_plot(seed, n)
    # undo the normalisation and spill some points outside the hexagon / below the lowest return
    raw = np.stack([(pos[:, 0] - 0.5) * 30.0, (pos[:, 1] - 0.5) * 30.0, pos[:, 2] * 40.0 + 3.25], 1)
    raw[:, :2] *= rng.uniform(1.0, 1.25)
    return raw.astype(np.float32), y


# ------------------------------------------------------------------------------------------------------------- CPU
def test_known_answers():
    pos = torch.tensor([[15.0, -15.0, 44.0], [0.0, 0.0, 4.0], [30.0, 0.0, 24.0]])
    p = T.start_z_from_zero(T.move_center(T.scale_pos(pos, (30.0, 30.0, 40.0), "div"), 0.5, 0.5))
    assert torch.equal(p, torch.tensor([[1.0, 0.0, 1.0], [0.5, 0.5, 0.0], [1.5, 0.5, 0.5]]))
    m = T.polygon_mask(p)
    assert m.tolist() == [False, True, False]
    x = T.features(p[m])
    assert x.shape == (1, 3) and x[0, 0] == 1 and x[0, 1] == 0
    assert abs(float(x[0, 2]) - np.sqrt(2) * 1e-6) < 1e-9          # PairwiseDistance adds eps before the norm


def test_crossing_test_is_matplotlibs():
    from matplotlib.path import Path
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.2, 1.2, size=(200_000, 2))
    # plus points hugging the edges (1e-9 .. 1e-4 away)
    poly = np.asarray(T.HEXAGON)
    a, b = poly, np.roll(poly, -1, 0)
    t = rng.uniform(0, 1, size=(6, 2000, 1))
    on = a[:, None] + t * (b - a)[:, None]
    nrm = np.stack([(b - a)[:, 1], -(b - a)[:, 0]], 1)[:, None]
    near = (on + nrm * rng.choice([-1, 1], size=(6, 2000, 1)) * 10 ** rng.uniform(-9, -4, size=(6, 2000, 1))).reshape(-1, 2)
    pts = np.concatenate([pts, near]).astype(np.float32).astype(np.float64)   # the kernel sees fp32 coordinates
    assert np.array_equal(point_in_polygon_np(pts[:, 0], pts[:, 1], T.HEXAGON), Path(T.HEXAGON).contains_points(pts))


def test_coordinate_augmentations_known_answers():
    c = torch.tensor([[0, 1, 2], [3, 0, 2], [1, 5, 0]], dtype=torch.int32)
    random.seed(4)
    draws = [random.random() for _ in range(3)]
    random.seed(4)
    torch.manual_seed(1)
    flags = T.random_coords_flip(c, p=0.5)
    assert flags == [int(draws[0] < 0.5), int(draws[1] < 0.5), 0]
    exp = torch.tensor([[0, 1, 2], [3, 0, 2], [1, 5, 0]], dtype=torch.int32)
    if flags[0]:
        exp[:, 0] = 3 - exp[:, 0]
    if flags[1]:
        exp[:, 1] = 5 - exp[:, 1]
    assert torch.equal(c, exp)
    shift = T.shift_voxels(c, p=0.5)
    torch.manual_seed(1)
    want = (torch.rand(3) * 100).to(torch.int32) if draws[2] < 0.5 else torch.zeros(3, dtype=torch.int32)
    assert torch.equal(shift, want) and torch.equal(c, exp + want)


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_prepare_matches_oracle(device):
    from dpcr_agb_amd.transforms import SparsePlotPipeline, nfi_test_transform
    sizes = [16000, 30000, 300, 9000, 1]      # one plot above MaxPoints after the crop, one below MinPoints, one point
    plots = [raw_plot(200 + i, n)[0] for i, n in enumerate(sizes)]
    plots[4] = np.array([[0.1, -0.2, 7.0]], np.float32)
    pipe = SparsePlotPipeline(nfi_test_transform()[:-1])     # everything before GridSampling3D
    torch.manual_seed(11)
    pos, x, src, lens = pipe.prepare(plots, device)
    torch.manual_seed(11)
    off, o_off = 0, 0
    for b, raw in enumerate(plots):
        op, ox, osrc = T.test_transform_sample(torch.from_numpy(raw))
        n = len(op)
        assert lens[b] == n, (b, lens[b], n)
        sl = slice(o_off, o_off + n)
        assert torch.equal(pos[sl].cpu(), op), b
        assert torch.equal(src[sl].cpu() - off, osrc), b
        assert torch.equal(x[sl, :2].cpu(), ox[:, :2]), b
        assert torch.allclose(x[sl, 2].cpu(), ox[:, 2], rtol=2e-7, atol=0), b   # sqrt of a 2-term sum: 1 ulp
        off += len(raw)
        o_off += n
    assert lens.tolist()[1] == 16000 and lens.tolist()[2] in (500, 0) and o_off == len(pos)


@pytest.mark.gpu
def test_device_chain_with_voxelisation_and_coord_augmentation(device):
    from dpcr_agb_amd.transforms import SparsePlotPipeline, nfi_coord_augmentation, nfi_test_transform
    sizes = [12000, 16000, 7000]
    plots = [raw_plot(300 + i, n)[0] for i, n in enumerate(sizes)]
    # oracle: per sample
    torch.manual_seed(3)
    ops = [T.test_transform_sample(torch.from_numpy(r)) for r in plots]
    lens = [len(o[0]) for o in ops]
    rng = np.random.default_rng(9)
    perms = [rng.permutation(n) for n in lens]
    oc, ok, ol = V.batch_grid_sampling_last(np.concatenate([o[0].numpy() for o in ops]), lens, perms, 0.0125)
    random.seed(1)
    torch.manual_seed(22)
    ocs, o0 = [], 0
    for n in ol:
        c = torch.from_numpy(oc[o0:o0 + n].copy())
        T.random_coords_flip(c, p=0.5)
        T.shift_voxels(c, p=0.5)
        ocs.append(c)
        o0 += n
    # device
    pipe = SparsePlotPipeline(nfi_test_transform() + nfi_coord_augmentation())
    torch.manual_seed(3)
    random.seed(1)
    # the flips/shifts are drawn after the voxelisation: re-seed torch there through the perms argument path
    pos, x, src, plens = pipe.prepare(plots, device)
    assert plens.tolist() == lens
    torch.manual_seed(22)
    out = pipe(plots, device, y_reg=np.ones((3, 2), np.float32), perms=torch.from_numpy(np.concatenate(perms)))
    assert torch.equal(out.coords.cpu(), torch.cat(ocs))
    # the box of the augmented coordinates, stated without reading them back, is the exact one
    allc = torch.cat(ocs)
    assert out.coord_bounds == tuple(allc.min(0).values.tolist()) + tuple(allc.max(0).values.tolist())
    assert np.array_equal(np.bincount(out.batch.cpu().numpy(), minlength=3), ol)
    ox = torch.cat([o[1] for o in ops])[torch.from_numpy(ok)]
    assert torch.equal(out.x[:, :2].cpu(), ox[:, :2])
    # the batch feeds the model
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    ds = synthetic.SyntheticDataset(feature_dimension=3, stat_seeds=range(10_000, 10_016))
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds).to(device).eval()
    with torch.no_grad():
        model.set_input(out, device)
        model.forward()
    assert model.output.shape == (3, 2) and torch.isfinite(model.output).all()


# ------------------------------------------------------------------------------------------------------- train chain
def _seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def test_train_draws_follow_the_oracles_generator_order():
    """The product's host-side draw function consumes random / numpy.random / torch exactly like the sequential oracle
    (which applies the transforms while drawing, as the reference does): same selections, noise, rotation, polygon."""
    import importlib.util
    import os
    from dpcr_agb_amd.train_transforms import NFITrainConfig, draw_sample
    raws = [torch.from_numpy(raw_plot(400 + i, n)[0]) for i, n in enumerate([9000, 16000, 700, 300])]
    for seed in (0, 1, 2, 3, 4, 5):      # different seeds hit different branches (ground removal p = 0.1 ...)
        _seed_all(seed)
        ora = [T.train_transform_sample(r) for r in raws]
        after = (random.random(), float(np.random.rand()), float(torch.rand(1)))
        _seed_all(seed)
        drs = [draw_sample(r, NFITrainConfig()) for r in raws]
        assert after == (random.random(), float(np.random.rand()), float(torch.rand(1))), seed
        for (pos, src, pre, orig), d in zip(ora, drs):
            assert torch.equal(d["sel"], orig)
            n_cj = 0 if d["cj_idx"] is None else len(d["cj_idx"])
            assert len(pre) == len(d["sel"]) + d["n_add"] + n_cj
    # the rotation helper is the reference's (when the reference tree is mounted: never on the GPU box)
    geo = "/root/reference/torch-points3d/torch_points3d/utils/geometry.py"
    if os.path.exists(geo):
        spec = importlib.util.spec_from_file_location("ref_geometry", geo)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        th = torch.tensor([0.3, -1.1, 2.0])
        random.seed(9)
        a = mod.euler_angles_to_rotation_matrix(th, random_order=True)
        random.seed(9)
        b = T.euler_angles_to_rotation_matrix(th, random_order=True)
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_device_train_chain_matches_oracle(device):
    from dpcr_agb_amd.train_transforms import NFITrainConfig, SparseTrainPipeline, draw_sample
    raws = [raw_plot(500 + i, n)[0] for i, n in enumerate([9000, 16000, 700, 11000, 5000, 300])]
    pipe = SparseTrainPipeline()
    hit = dict(ground=0, drop=0, add=0, cj=0, shift=0)
    for seed in range(6):
        _seed_all(seed)
        ora = [T.train_transform_sample(torch.from_numpy(r)) for r in raws]
        _seed_all(seed)
        draws = [draw_sample(torch.from_numpy(r), NFITrainConfig()) for r in raws]
        pos, x, src, out_ptr = pipe.augment(raws, draws, device)
        optr = out_ptr.cpu().numpy()
        pos, src = pos.cpu(), src.cpu()
        pre_off = 0
        for b, ((opos, osrc, pre, orig), d) in enumerate(zip(ora, draws)):
            dp, ds = pos[optr[b]:optr[b + 1]], src[optr[b]:optr[b + 1]] - pre_off
            pre_off += len(pre)
            # the rotation is a 3x3 torch.mm on the CPU and three multiply-adds on the device: positions agree to a few
            # ulp, so a point within that distance of a polygon edge may be classified differently
            common, ia, ib = np.intersect1d(ds.numpy(), osrc.numpy(), return_indices=True)
            assert len(ds) + len(osrc) - 2 * len(common) <= 2, (seed, b)
            assert torch.allclose(dp[ia], opos[ib], rtol=0, atol=2e-6), (seed, b)
            hit["ground"] += d["zsub"] > 0
            hit["drop"] += len(d["sel"]) < len(raws[b])
            hit["add"] += d["n_add"] > 0
            hit["cj"] += d["cj_idx"] is not None
            hit["shift"] += d["shift"] is not None
    assert all(v > 0 for v in hit.values()), hit      # every branch of the chain was exercised
    # end to end: a voxelised, coordinate-augmented batch that the model accepts
    _seed_all(11)
    out = pipe(raws, device, y_reg=np.ones((len(raws), 2), np.float32))
    assert out.coords.shape[0] == out.x.shape[0] == out.batch.shape[0] and out.coords.dtype == torch.int32
    assert int(out.batch.max()) == len(raws) - 1


def test_collated_draws_and_loader_workers():
    """collate_draws stacks a batch's draws without changing them, and SampleDraws behind a DataLoader with worker processes
    (where the reference runs its transforms) yields such batches with a different generator state in every worker."""
    from functools import partial
    from torch.utils.data import DataLoader
    from dpcr_agb_amd.train_transforms import NFITrainConfig, SampleDraws, collate_draws, draw_sample
    cfg = NFITrainConfig()
    raws = [torch.from_numpy(raw_plot(700 + i, n)[0]) for i, n in enumerate([3000, 5000, 700, 300])]
    _seed_all(3)
    drs = [draw_sample(r, cfg) for r in raws]
    col = collate_draws(drs, cfg)
    n_raw = [len(r) for r in raws]
    assert col["B"] == 4 and col["n_raw"].tolist() == n_raw
    off = np.concatenate([[0], np.cumsum(n_raw)])
    assert torch.equal(col["sel"], torch.cat([d["sel"] + int(off[b]) for b, d in enumerate(drs)]))
    assert col["n1s"].tolist() == [len(d["sel"]) for d in drs] and col["noise"].shape == (int(col["n1s"].sum()), 3)
    assert col["n_add"].tolist() == [d["n_add"] for d in drs]
    assert col["cj_idx"].shape[0] == int(col["n_cj"].sum()) + 1 == col["cj_noise"].shape[0]
    assert torch.equal(col["aug"][:, 4:13].reshape(4, 3, 3), torch.stack([d["M"] for d in drs]))
    assert col["polys"].shape == (4, 2 * col["nv"])
    loader = DataLoader(SampleDraws(raws, cfg, length=16), batch_size=4, num_workers=2, collate_fn=partial(collate_draws, cfg=cfg),
                        worker_init_fn=SampleDraws.seed_worker)
    got = list(loader)
    assert len(got) == 4 and all(g["n_raw"].tolist() == n_raw for g in got)
    # batches 0 and 1 come from different workers: different rotations and polygons
    assert not torch.equal(got[0]["aug"], got[1]["aug"]) and not torch.equal(got[0]["polys"], got[1]["polys"])


@pytest.mark.gpu
def test_device_shuffle_is_a_permutation_per_cloud_and_collated_draws_give_the_same_rows(device):
    from dpcr_agb_amd.train_transforms import NFITrainConfig, SparseTrainPipeline, collate_draws, draw_sample
    from dpcr_agb_amd.voxelize import device_permutations
    lens = [5000, 0, 1, 13000, 77]
    a, b = device_permutations(lens, device).cpu(), device_permutations(lens, device).cpu()
    assert a.shape[0] == sum(lens) and not torch.equal(a, b)
    o = 0
    for n in lens:
        assert torch.equal(torch.sort(a[o:o + n]).values, torch.arange(n))
        o += n
    # a uniform shuffle: the first shuffled position of the 13 000-row cloud over 200 draws covers the range evenly
    firsts = torch.stack([device_permutations(lens, device)[5001] for _ in range(200)]).float().cpu()
    assert 0.40 * 13000 < float(firsts.mean()) < 0.60 * 13000 and float(firsts.std()) > 0.2 * 13000
    raws = [raw_plot(800 + i, n)[0] for i, n in enumerate([9000, 16000, 700])]
    cfg = NFITrainConfig()
    _seed_all(5)
    draws = [draw_sample(torch.from_numpy(r), cfg) for r in raws]
    pipe = SparseTrainPipeline(cfg, device_shuffle=True)
    r1 = pipe.augment(raws, draws, device)
    r2 = pipe.augment([torch.from_numpy(r).to(device) for r in raws], collate_draws(draws, cfg), device)
    m = int(r1[3][-1])
    assert torch.equal(r1[3], r2[3]) and torch.equal(r1[0][:m], r2[0][:m]) and torch.equal(r1[2][:m], r2[2][:m])
    _seed_all(6)
    out = pipe(raws, device, y_reg=np.ones((3, 2), np.float32))
    assert out.coords.shape[0] == out.x.shape[0] and int(out.batch.max()) == 2
    with pytest.raises(ValueError):
        pipe.augment(raws[:2] + [raws[0]], draws, device)



@pytest.mark.gpu
def test_staged_train_chain_equals_the_waiting_one(device):
    """SparseTrainPipeline.staged / StagedBatches (the chain's two count read-backs picked up a step later instead of
    waited for): same draws, same voxel shuffle, same flips / shifts -> the same PlotBatch as ``__call__``, bit for bit,
    also with several batches in flight on a side stream."""
    from dpcr_agb_amd.train_transforms import NFITrainConfig, SparseTrainPipeline, StagedBatches, draw_sample
    cfg = NFITrainConfig()
    # (plots that stay between MinPoints and MaxPoints after the crop: those two draw from torch's generator in the MIDDLE stage,
    # which a pipelined loop runs for all batches in flight before any batch's last stage — a different, equally valid order)
    sets = [[raw_plot(900 + 10 * j + i, n)[0] for i, n in enumerate([9000, 16000, 4000, 12000])] for j in range(3)]
    _seed_all(7)
    draws = [[draw_sample(torch.from_numpy(r), cfg) for r in raws] for raws in sets]
    ys = [np.full((4, 2), float(j), np.float32) for j in range(3)]

    def perms_for(pipe, raws, d):       # the voxel shuffle of this batch, fixed: lengths after the crop / MaxPoints
        _seed_all(11)
        pos, x, src, out_ptr = pipe.augment(raws, d, device)
        lens = pipe.tail.fix_counts(pos, x, src, out_ptr, with_extent=True)[3]
        g = torch.Generator().manual_seed(3)
        return torch.cat([torch.randperm(int(n), generator=g) for n in lens])

    pipe = SparseTrainPipeline(cfg)
    perms = [perms_for(pipe, sets[j], draws[j]) for j in range(3)]
    # reference: the three batches one after the other; the flips / shifts (the only draws left: no plot exceeds MaxPoints
    # after the crop, the voxel shuffle is given) come from one seeded sequence
    _seed_all(20)
    want = [pipe(sets[j], device, y_reg=ys[j], draws=draws[j], perms=perms[j]) for j in range(3)]
    # the staged form: three batches in flight on a side stream; first advance = the middle stages (no draws), second = the
    # final stages in submission order, drawing the same sequence
    side = torch.cuda.Stream()
    flight = StagedBatches(SparseTrainPipeline(cfg), device, side)
    for j in range(3):
        flight.submit(sets[j], y_reg=ys[j], draws=draws[j], perms=perms[j])
    assert len(flight) == 3
    assert flight.advance() == [] and len(flight) == 3
    _seed_all(20)
    order = flight.advance()
    side.synchronize()
    assert len(order) == 3 and len(flight) == 0
    for j, out in enumerate(order):
        ref = want[j]
        assert torch.equal(out.coords, ref.coords) and torch.equal(out.x, ref.x) and torch.equal(out.batch, ref.batch), j
        assert torch.equal(out.y_reg, ref.y_reg) and out.coord_bounds == ref.coord_bounds, j


@pytest.mark.gpu
def test_point_train_chain_of_the_kpconv_models(device):
    """PointTrainPipeline = xy.yaml:4-75 (the sparse chain without the voxel tail, MaxPoints 6144): the rows it keeps are rows
    of the augmented cloud (same draws as the sparse chain, checked against the oracle in
    test_device_train_chain_matches_oracle), counts obey MaxPoints / MinPoints, features are [1, z, xy distance to the centre],
    the row offsets are known on the host, the staged form gives the same batch, and the KPConv model trains on it."""
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import KPConvModel
    from dpcr_agb_amd.train_transforms import NFITrainConfig, PointTrainPipeline, StagedBatches, draw_sample
    cfg = NFITrainConfig(voxel=None, max_points=6144)
    raws = [raw_plot(700 + i, n)[0] for i, n in enumerate([9000, 16000, 700, 12000, 300])]
    _seed_all(5)
    draws = [draw_sample(torch.from_numpy(r), cfg) for r in raws]
    pipe = PointTrainPipeline(cfg)
    assert pipe.tail.grid is None and pipe.tail.max_points == 6144 and pipe.tail.flip is None
    ys = np.arange(10, dtype=np.float32).reshape(5, 2)
    _seed_all(6)
    out = pipe(raws, device, y_reg=ys, draws=draws)
    pos_a, x_a, src_a, optr = pipe.augment(raws, draws, device)
    optr = optr.cpu().numpy()
    lens = np.diff(out.ptr.numpy())
    assert out.coords is None and out.ptr.device.type == "cpu" and int(out.ptr[-1]) == out.pos.shape[0]
    assert np.array_equal(lens, np.bincount(out.batch.cpu().numpy(), minlength=5))
    for b in range(5):
        kept = optr[b + 1] - optr[b]
        assert lens[b] == (6144 if kept > 6144 else 500 if 0 < kept < 500 else kept), (b, kept, lens[b])
    assert lens.max() == 6144 and lens.min() == 500            # both branches ran
    # the rows are rows of the cropped, augmented cloud of their own plot
    src, rows = out.src.cpu(), src_a.cpu()
    lookup = {int(s): i for i, s in enumerate(rows[:optr[-1]].tolist())}
    at = torch.tensor([lookup[int(s)] for s in src.tolist()])
    assert torch.equal(out.pos.cpu(), pos_a.cpu()[at]) and torch.equal(out.x.cpu(), x_a.cpu()[at])
    p, x = out.pos.cpu(), out.x.cpu()
    assert torch.equal(x[:, 0], torch.ones(len(x))) and torch.equal(x[:, 1], p[:, 2])
    assert torch.allclose(x, T.features(p), rtol=0, atol=1e-6)      # (torch.nn.PairwiseDistance: eps inside the norm)
    assert torch.equal(out.y_reg.cpu(), torch.from_numpy(ys))
    # staged: the count read-back picked up a step later — the same batch
    flight = StagedBatches(PointTrainPipeline(cfg), device, torch.cuda.Stream())
    flight.submit(raws, y_reg=ys, draws=draws)
    _seed_all(6)
    done = flight.advance()
    torch.cuda.synchronize()
    assert len(done) == 1 and torch.equal(done[0].pos, out.pos) and torch.equal(done[0].x, out.x)
    assert torch.equal(done[0].batch, out.batch) and torch.equal(done[0].ptr, out.ptr)
    # the KPConv model takes the batch as it is
    torch.manual_seed(0)
    model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016)))
    model.to(device).train()
    model.init_train_objects(TRAINING_NFI)
    model.set_input(out, device)
    model.optimize_parameters(epoch=0, batch_size=5, num_batches=10)
    assert np.isfinite(float(model.loss.detach()))
