"""Plot readers and the Las / LasDataset interface (dpcr_agb_amd/plots.py; reference: datasets/instance/las_dataset.py:32-71
read_pt, :421-435 Las.get, :477-512 the cached sample record, :894-940 per-area target statistics).  laspy / plyfile are
not installed: the LAS reader is checked against a byte-level fixture written here field by field from the public LAS 1.2
layout (not by the module's own writer) and by write -> read round trips."""
import os
import struct

import numpy as np
import pytest
import torch

from dpcr_agb_amd import plots
from dpcr_agb_amd.config import NFI_TARGETS


def _handmade_las(path, pts, intensity, rn, nr, gps):
    """LAS 1.2, point format 1, 28-byte records + 2 extra bytes per record, assembled with struct only."""
    n = len(pts)
    scale, off = (0.01, 0.01, 0.001), (100.0, -50.0, 2.0)
    head = bytearray(227)
    head[0:4] = b"LASF"
    head[24:26] = bytes([1, 2])
    struct.pack_into("<H", head, 94, 227)
    struct.pack_into("<I", head, 96, 227 + 10)             # 10 bytes of padding before the points
    head[104] = 1
    struct.pack_into("<H", head, 105, 30)
    struct.pack_into("<I", head, 107, n)
    struct.pack_into("<6d", head, 131, *scale, *off)
    body = bytearray()
    for i in range(n):
        q = [int(round((pts[i][d] - off[d]) / scale[d])) for d in range(3)]
        body += struct.pack("<iiiHBBbBHd", q[0], q[1], q[2], intensity[i], (rn[i] & 7) | ((nr[i] & 7) << 3), 2, -5, 0, 7,
                            gps[i])
        body += b"\xAA\xBB"
    with open(path, "wb") as f:
        f.write(bytes(head) + b"\x00" * 10 + bytes(body))


def test_read_las_against_handmade_bytes(tmp_path):
    rng = np.random.default_rng(0)
    pts = np.round(rng.uniform([100, -50, 2], [130, -20, 40], size=(57, 3)), 2)
    inten, rn, nr = rng.integers(0, 60000, 57), rng.integers(1, 5, 57), rng.integers(1, 6, 57)
    gps = rng.uniform(0, 1e5, 57)
    p = str(tmp_path / "a.las")
    _handmade_las(p, pts, inten, rn, nr, gps)
    pos, feats, crs = plots.read_pt(p, ["intensity", "return_number", "number_of_returns", "gps_time", "classification"])
    assert crs is None and pos.shape == (57, 3)
    assert np.allclose(pos, pts, atol=1e-9)
    assert np.array_equal(feats[:, 0], inten) and np.array_equal(feats[:, 1], rn) and np.array_equal(feats[:, 2], nr)
    assert np.allclose(feats[:, 3], gps) and np.all(feats[:, 4] == 2)


@pytest.mark.parametrize("fmt", [0, 1, 2, 3])
def test_las_round_trip_and_errors(tmp_path, fmt):
    rng = np.random.default_rng(fmt)
    pts = rng.uniform(-10, 10, size=(200, 3))
    p = str(tmp_path / "b.las")
    plots.write_las(p, pts, scale=0.001, point_format=fmt, intensity=rng.integers(0, 100, 200),
                    return_number=rng.integers(1, 4, 200))
    pos, dims = plots.read_las(p)
    assert np.abs(pos - pts).max() <= 0.0005 + 1e-12 and "intensity" in dims
    with open(p, "r+b") as f:           # flag the records as compressed: must be refused, not mis-read
        f.seek(104)
        f.write(bytes([fmt | 0x80]))
    with pytest.raises(ValueError):
        plots.read_las(p)
    with pytest.raises(ValueError):
        plots.read_pt(str(tmp_path / "x.laz"))
    empty = str(tmp_path / "e.las")
    plots.write_las(empty, np.zeros((0, 3)))
    assert plots.read_las(empty)[0].shape == (0, 3)


def test_read_ply_and_csv(tmp_path):
    rng = np.random.default_rng(3)
    v = rng.uniform(0, 5, size=(11, 4)).astype(np.float32)
    a = tmp_path / "a.ply"
    a.write_text("ply\nformat ascii 1.0\ncomment x\nelement vertex 11\nproperty float x\nproperty float y\n"
                 "property float z\nproperty float intensity\nelement face 0\nproperty list uchar int vertex_indices\n"
                 "end_header\n" + "\n".join(" ".join(f"{t:.7g}" for t in r) for r in v) + "\n")
    pos, feats, _ = plots.read_pt(str(a), ["intensity"])
    assert np.allclose(pos, v[:, :3], rtol=1e-6) and np.allclose(feats[:, 0], v[:, 3], rtol=1e-6)
    b = tmp_path / "b.ply"
    with open(b, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 11\nproperty double x\nproperty double y\n"
                b"property double z\nproperty uchar cls\nend_header\n")
        rec = np.zeros(11, dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("cls", "u1")])
        rec["x"], rec["y"], rec["z"], rec["cls"] = v[:, 0], v[:, 1], v[:, 2], np.arange(11)
        rec.tofile(f)
    pos, feats, _ = plots.read_pt(str(b), ["cls"])
    assert np.allclose(pos, v[:, :3].astype(np.float64)) and np.array_equal(feats[:, 0], np.arange(11))
    c = tmp_path / "c.csv"
    np.savetxt(c, v, delimiter=";", fmt="%.6f")
    pos, feats, _ = plots.read_pt(str(c), [3], delimiter=";")
    assert np.allclose(pos, v[:, :3], atol=1e-5) and np.allclose(feats[:, 0], v[:, 3], atol=1e-5)


def test_las_dataset_interface(tmp_path):
    rng = np.random.default_rng(5)
    files, labels = [], []
    for i in range(7):
        pts = rng.uniform([500, 900, 3], [530, 930, 30], size=(40 + i, 3))
        f = str(tmp_path / f"p{i}.las")
        plots.write_las(f, pts, intensity=rng.integers(0, 100, len(pts)))
        files.append(f)
        labels.append(dict(y_reg=[100.0 + i, np.nan if i == 2 else 200.0 + 2 * i], area_name="n" if i < 4 else "s",
                           x=515.0, y=915.0))
    paths = plots.process_plot_files(files, labels, str(tmp_path / "processed"), feature_cols=["intensity"])
    ds = plots.LasDataset({"train": paths[:5], "val": paths[5:]}, NFI_TARGETS, feature_dimension=1)
    assert ds.reg_targets == ["BMag_ha", "V_ha"] and ds.num_reg_classes == 2 and list(ds.areas) == ["n", "s"]
    m = ds.get_mean_targets()
    assert np.allclose(m["total"]["train"], [102.0, np.mean([200, 202, 206, 208])])
    assert np.allclose(m["n"]["train"], [101.5, np.mean([200, 202, 206])]) and "val" not in m["n"]
    assert np.allclose(ds.get_std_targets()["s"]["val"], [0.5, 1.0])
    s0 = ds.train_dataset.get(0)
    assert s0["is_double"] is False and ds.train_dataset.get(0)["is_double"] is True
    assert ds.train_dataset.get(1)["is_double"] is False
    assert s0["pos"].dtype == torch.float32 and abs(float(s0["pos"][:, 2].min())) < 1e-6    # z from the plot's minimum
    assert abs(float(s0["pos"][:, 0].mean())) < 5.0                                         # xy centred on the label
    s2 = ds.train_dataset.get(2)
    assert s2["y_reg_mask"].tolist() == [True, False]
    batch = plots.collate([ds.train_dataset.get(i) for i in range(3)])
    assert len(batch) == 3 and batch.pos.shape[0] == 40 + 41 + 42 and batch.x.shape[1] == 1
    assert batch.y_reg_mask_all is False and batch.area_name == ["n", "n", "n"]
    # the model contract reads its target statistics from this dataset (models/instance/base.py:86-114)
    from dpcr_agb_amd.config import Opt
    from dpcr_agb_amd.instance.base import InstanceBase
    mb = InstanceBase(Opt(), "x", ds)
    assert torch.allclose(mb.reg_center_targets, torch.tensor([[np.mean([102.0, 101.5, 104.0]),
                                                                np.mean([204.0, 608 / 3.0, 208.0])]], dtype=torch.float))
