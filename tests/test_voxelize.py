"""GridSampling3D(mode='last'): oracle known answers on CPU, HIP kernel vs oracle on the GPU (bit-exact indices)."""
import numpy as np
import pytest
import torch

from oracle import voxelize_ref as V


def test_oracle_known_answer():
    # 5 points, voxel 1.0: points 0,2 share a voxel; 1,4 share a voxel; half-to-even rounding at x = 0.5 / 1.5 / 2.5
    pos = np.array([[0.2, 0.1, 0.0], [1.5, 0.0, 0.0], [0.4, -0.2, 0.1], [2.5, 0.0, 0.0], [2.4, 0.0, 0.0]], np.float32)
    perm = np.array([3, 0, 4, 2, 1])  # shuffled order: 3, 0, 4, 2, 1
    coords, keep = V.grid_sampling_last(pos, perm, 1.0)
    # round: [0,0,0], [2,0,0] (1.5 -> 2), [0,0,0], [2,0,0] (2.5 -> 2), [2,0,0]: two voxels
    assert coords.tolist() == [[0, 0, 0], [2, 0, 0]]
    # last shuffled point of voxel 0 is original 2 (position 3); of voxel 2 is original 1 (position 4)
    assert keep.tolist() == [2, 1]


def test_oracle_matches_generator_helper():
    from dpcr_agb_amd import synthetic
    pos, _, _ = synthetic.make_plot(3, 4000)
    perm = np.random.default_rng(0).permutation(len(pos))
    c1, k1 = V.grid_sampling_last(pos, perm, 0.0125)
    c2, k2 = synthetic.voxelize_host(pos, perm, 0.0125)
    assert np.array_equal(c1, c2) and np.array_equal(k1, k2)
    # sorted by key: z-major, x fastest
    key = (c1[:, 2].astype(np.int64) * 10**6 + c1[:, 1]) * 10**6 + c1[:, 0]
    assert (np.diff(key) > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,vox", [([16000, 16000, 9000], 0.0125), ([500, 1, 2000, 37], 0.05)])
def test_hip_voxelize_matches_oracle(device, sizes, vox):
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.voxelize import voxelize_last
    rng = np.random.default_rng(5)
    pos = np.concatenate([synthetic.make_plot(70 + i, n)[0] for i, n in enumerate(sizes)])
    sizes = [min(n, len(synthetic.make_plot(70 + i, n)[0])) for i, n in enumerate(sizes)]
    perms = [rng.permutation(n) for n in sizes]
    coords, keep, lens, bounds = voxelize_last(torch.from_numpy(pos), sizes, vox, perm=torch.from_numpy(np.concatenate(perms)))
    oc, ok, ol = V.batch_grid_sampling_last(pos, sizes, perms, vox)
    assert np.array_equal(lens, ol)
    assert np.array_equal(coords.cpu().numpy(), oc)
    assert np.array_equal(keep.cpu().numpy(), ok)
    assert bounds == tuple(oc.min(0).tolist()) + tuple(oc.max(0).tolist())
    # idempotence: voxelising one point per voxel again (identity permutation) changes nothing
    c2, k2, l2, _ = voxelize_last(torch.from_numpy(pos[ok]), ol, vox,
                                  perm=torch.cat([torch.arange(int(n)) for n in ol]))
    assert np.array_equal(c2.cpu().numpy(), oc) and np.array_equal(l2, ol)


@pytest.mark.gpu
def test_gridsampling_transform_feeds_the_sparse_model(device):
    """Raw points -> GridSampling3D on the GPU -> MSENet14 forward equals the host-voxelised path bit for bit."""
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    from dpcr_agb_amd.voxelize import GridSampling3D
    seeds = [3, 4]
    raw = synthetic.make_point_batch(seeds, n_points=3000)
    perms = [np.random.default_rng(s + 7919).permutation(int(n)) for s, n in
             zip(seeds, np.bincount(raw.batch.numpy()))]
    vox = GridSampling3D(0.0125, quantize_coords=True, mode="last")(raw, perm=torch.from_numpy(np.concatenate(perms)))
    host = synthetic.make_sparse_batch(seeds, n_points=3000)
    assert np.array_equal(vox.coords.cpu().numpy(), host.coords.numpy())
    assert np.array_equal(vox.x.cpu().numpy(), host.x.numpy())
    assert vox.coord_bounds == host.coord_bounds
    torch.manual_seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016))
    opt = Opt(MODEL_OPTIONS["SENet14"])
    opt["drop_path"] = 0.0
    model = MinkowskiBaselineModel(opt, "minkowski", ds).to(device).train()
    outs = []
    for batch in (vox, host):
        model.set_input(batch, device)
        model.forward()
        outs.append(model.output.detach().clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.gpu
def test_seeded_device_shuffle_picks_uniform_representatives(device):
    """agb_voxelize_last_seeded_ws: the shuffle drawn inside the voxeliser from a seed (a keyed pseudo-random bijection per
    cloud).  (i) it IS a bijection: with one point per voxel every point is kept, for many cloud sizes; (ii) same voxels as the
    permutation-driven form, the representative a point of its voxel; (iii) reproducible per seed, different across seeds;
    (iv) every point of a voxel is picked about equally often over many seeds (GridSampling3D(mode="last") asks no more of the
    shuffle: grid_transform.py:118-121)."""
    from dpcr_agb_amd.voxelize import voxelize_last
    # (i) one point per voxel: sizes around powers of two and four (the Feistel domain changes there), several clouds
    for sizes in ([1, 2, 3, 4, 5], [15, 16, 17, 63, 64, 65], [255, 256, 257, 1000, 4096, 4097], [16000, 9999]):
        pts = np.concatenate([np.stack([np.arange(n) * 1.0, np.full(n, 3.0 * b), np.zeros(n)], 1) for b, n in enumerate(sizes)])
        c, keep, lens, _ = voxelize_last(torch.from_numpy(pts.astype(np.float32)).to(device), sizes, 1.0, seed=12345)
        assert lens.tolist() == list(sizes)
        assert torch.equal(torch.sort(keep.cpu()).values, torch.arange(sum(sizes)))
    # (ii) + (iii) on a real cloud
    rng = np.random.default_rng(0)
    pos = torch.from_numpy(rng.uniform(0, 1, size=(12000, 3)).astype(np.float32)).to(device)
    lens = [7000, 5000]
    c0, k0, l0, b0 = voxelize_last(pos, lens, 0.05, seed=7)
    c1, k1, l1, _ = voxelize_last(pos, lens, 0.05, seed=7)
    c2, k2, l2, _ = voxelize_last(pos, lens, 0.05, seed=8)
    cp, kp, lp, bp = voxelize_last(pos, lens, 0.05)                      # host permutation
    assert torch.equal(c0, c1) and torch.equal(k0, k1) and torch.equal(c0, cp) and l0.tolist() == lp.tolist() and b0 == bp
    assert torch.equal(c0, c2) and not torch.equal(k0, k2)
    cell_of = torch.round(pos[k0] / np.float32(0.05)).to(torch.int32)      # the kept point lies in the voxel it stands for
    assert torch.equal(cell_of, c0)
    # (iv) one voxel of 8 points, 4000 seeds: every point wins ~1/8 of the time (binomial sd 0.005)
    pts = torch.tensor([[0.1 + 0.01 * i, 0.2, 0.3] for i in range(8)] + [[5.0 + i, 0.0, 0.0] for i in range(50)],
                       dtype=torch.float32, device=device)
    wins = np.zeros(8)
    for s in range(4000):
        _, keep, _, _ = voxelize_last(pts, [58], 1.0, seed=1000 + s)
        w = [int(k) for k in keep.tolist() if k < 8]
        assert len(w) == 1
        wins[w[0]] += 1
    assert np.abs(wins / 4000 - 0.125).max() < 0.03, wins
