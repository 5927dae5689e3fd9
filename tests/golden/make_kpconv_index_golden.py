"""Generates tests/golden/kpconv_index_golden.npz from the REFERENCE's own C++ (oracle/_ref/libref_kpconv.so, built by
oracle/Makefile from /root/reference — only possible where that tree is mounted).  The .npz holds inputs and the
reference's outputs only (data, no reference source).  Run:  python tests/golden/make_kpconv_index_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kpconv_index as K  # noqa: E402
from dpcr_agb_amd import synthetic  # noqa: E402


def main():
    out = {}
    # case A: three ragged plots, self-neighbours at two radii (levels 0/1 of kpconv.py:148,234)
    sizes = [700, 350, 1]
    pts, feats = [], []
    for i, n in enumerate(sizes):
        p, x, _ = synthetic.make_plot(100 + i, n_points=n)
        pts.append(p)
        feats.append(x)
    pts, feats = np.concatenate(pts), np.concatenate(feats)
    lens = np.array(sizes, dtype=np.int32)
    out["A_points"], out["A_feats"], out["A_lens"] = pts, feats, lens
    for r in (0.03125, 0.0625):
        out[f"A_neighbors_r{r}"] = K.ref_batch_neighbors(pts, pts, lens, lens, r)
    # case B: grid subsampling (reference emission order) + pooled neighbours (queries = subsampled points)
    for dl in (0.025, 0.05):
        sp, sb, sf = K.ref_batch_grid_subsampling(pts, lens, features=feats, sampleDl=dl)
        out[f"B_sub_points_dl{dl}"], out[f"B_sub_lens_dl{dl}"], out[f"B_sub_feats_dl{dl}"] = sp, sb, sf
        out[f"B_pool_neighbors_dl{dl}"] = K.ref_batch_neighbors(sp, pts, sb, lens, 0.03125 if dl == 0.025 else 0.0625)
    sp, sb = K.ref_batch_grid_subsampling(pts, lens, sampleDl=0.025, max_p=200)
    out["B_sub_points_maxp200"], out["B_sub_lens_maxp200"] = sp, sb
    # case C: exact duplicates (MinPoints duplicates points, transforms.py:1742-1769) and a far outlier
    rng = np.random.default_rng(7)
    base = rng.uniform(0, 0.2, size=(120, 3)).astype(np.float32)
    dup = np.concatenate([base, base[:30], np.array([[5, 5, 5]], dtype=np.float32)])
    lens_c = np.array([len(dup)], dtype=np.int32)
    out["C_points"], out["C_lens"] = dup, lens_c
    out["C_neighbors_r0.05"] = K.ref_batch_neighbors(dup, dup, lens_c, lens_c, 0.05)
    sp, sb = K.ref_batch_grid_subsampling(dup, lens_c, sampleDl=0.04)
    out["C_sub_points_dl0.04"], out["C_sub_lens_dl0.04"] = sp, sb
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kpconv_index_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
