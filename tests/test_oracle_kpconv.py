"""Pins the KPConv index oracle (oracle/kpconv_index_ref.cpp): against the committed golden vectors generated from
the reference's own C++, and — where /root/reference is mounted — against that C++ directly on fresh inputs."""
import os

import numpy as np
import pytest

from oracle import kpconv_index as K

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kpconv_index_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_neighbors_match_golden(gold):
    pts, lens = gold["A_points"], gold["A_lens"]
    for r in (0.03125, 0.0625):
        got = K.batch_neighbors(pts, pts, lens, lens, r)
        assert np.array_equal(got, gold[f"A_neighbors_r{r}"])
    # shadow index = number of supports; a 1-point cloud has itself as only neighbour
    assert gold["A_neighbors_r0.03125"][-1, 0] == len(pts) - 1
    assert (gold["A_neighbors_r0.03125"][-1, 1:] == len(pts)).all()


def test_grid_subsampling_matches_golden(gold):
    pts, feats, lens = gold["A_points"], gold["A_feats"], gold["A_lens"]
    for dl in (0.025, 0.05):
        sp, sb, sf = K.batch_grid_subsampling(pts, lens, features=feats, sampleDl=dl, order="reference")
        assert np.array_equal(sb, gold[f"B_sub_lens_dl{dl}"])
        assert np.array_equal(sp, gold[f"B_sub_points_dl{dl}"])      # bit-equal barycentres, reference order
        assert np.array_equal(sf, gold[f"B_sub_feats_dl{dl}"])
        r = 0.03125 if dl == 0.025 else 0.0625
        # barycentres of 2-point cells are exactly equidistant from both points: tie order is unspecified in the
        # reference (std::sort on d2), everything else must be bit-identical
        assert K.same_up_to_ties(K.batch_neighbors(sp, pts, sb, lens, r), gold[f"B_pool_neighbors_dl{dl}"], sp, pts)
        # canonical order = the same cells, key-sorted per cloud
        cp, cb, cf, keys = K.batch_grid_subsampling(pts, lens, features=feats, sampleDl=dl, return_keys=True)
        assert np.array_equal(cb, sb)
        off = 0
        for n in cb:
            assert (np.diff(keys[off:off + n]) > 0).all()
            a = {tuple(x) for x in np.concatenate([sp[off:off + n], sf[off:off + n]], 1).tolist()}
            b = {tuple(x) for x in np.concatenate([cp[off:off + n], cf[off:off + n]], 1).tolist()}
            assert a == b
            off += n
    sp, sb = K.batch_grid_subsampling(pts, lens, sampleDl=0.025, max_p=200, order="reference")
    assert np.array_equal(sp, gold["B_sub_points_maxp200"]) and np.array_equal(sb, gold["B_sub_lens_maxp200"])


def test_duplicates_and_outlier(gold):
    pts, lens = gold["C_points"], gold["C_lens"]
    got, ref = K.batch_neighbors(pts, pts, lens, lens, 0.05), gold["C_neighbors_r0.05"]
    assert not np.array_equal(got, ref)          # duplicates DO tie ...
    assert K.same_up_to_ties(got, ref, pts, pts)  # ... and only the order inside ties differs
    swapped = got.copy()
    swapped[0, :2] = swapped[0, 1::-1]
    assert not K.same_up_to_ties(swapped, ref, pts, pts) or \
        K.neighbor_d2(pts, pts, got)[0, 0] == K.neighbor_d2(pts, pts, got)[0, 1]
    sp, sb = K.batch_grid_subsampling(pts, lens, sampleDl=0.04, order="reference")
    assert np.array_equal(sp, gold["C_sub_points_dl0.04"]) and np.array_equal(sb, gold["C_sub_lens_dl0.04"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not mounted")
def test_against_reference_build_on_fresh_inputs():
    from dpcr_agb_amd import synthetic
    for seed, n, r, dl in [(1, 1500, 0.03125, 0.025), (2, 900, 0.125, 0.1), (3, 2500, 0.0625, 0.05)]:
        b = synthetic.make_point_batch([seed, seed + 50], n_points=n)
        pts = b.pos.numpy()
        lens = np.bincount(b.batch.numpy()).astype(np.int32)
        assert np.array_equal(K.batch_neighbors(pts, pts, lens, lens, r), K.ref_batch_neighbors(pts, pts, lens, lens, r))
        rp, rb = K.ref_batch_grid_subsampling(pts, lens, sampleDl=dl)
        mp, mb = K.batch_grid_subsampling(pts, lens, sampleDl=dl, order="reference")
        assert np.array_equal(rp, mp) and np.array_equal(rb, mb)
        assert K.same_up_to_ties(K.batch_neighbors(mp, pts, mb, lens, r), K.ref_batch_neighbors(rp, pts, rb, lens, r),
                                 mp, pts)
