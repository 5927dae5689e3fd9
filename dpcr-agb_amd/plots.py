"""Plot files -> samples -> batches: the data side of the drop-in boundary.

Restates, without laspy / plyfile / torch_geometric / geopandas (none of them is installed here):
  * ``read_pt(pt_file, feature_cols, delimiter)``  (datasets/instance/las_dataset.py:32-71): positions (+ named per-point
    feature columns) from an uncompressed ``.las`` file (LAS 1.0-1.4, point formats 0-10), a ``.ply`` file (ascii or
    binary_little_endian) or a headerless CSV (first three columns = x, y, z, as the reference assumes);
  * the per-sample record the reference caches as ``.pt`` (``covert_to_data_``, las_dataset.py:477-512): float32
    ``pos`` / ``x``, ``y_reg`` / ``y_reg_mask``, ``area_name``, ``label_idx`` — stored here as a plain dict through
    ``torch.save`` (the reference pickles a torch_geometric ``Data``; ``load_sample`` also accepts such an object when
    torch_geometric happens to be importable);
  * ``Las.get(idx)`` (las_dataset.py:421-435): in-memory cache of the processed files, ``is_double`` = "same index as
    the previous call";
  * the dataset attributes the models and the tracker read (``LasDataset``: feature_dimension, num_classes,
    num_reg_classes, reg_targets, reg_targets_idx, targets, areas, get_{mean,std,min,max}_targets()), with the target
    statistics computed per area and split like ``get_stat_targets_`` (las_dataset.py:894-940), NaN-aware.
Radius cropping around plot centres, GIS joins and the 40 CPU augmentations stay out of scope.
"""
import os
import struct
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

# LAS point-record layouts: (name, numpy dtype) of the fixed part of every point format
_LAS_CORE_OLD = [("X", "<i4"), ("Y", "<i4"), ("Z", "<i4"), ("intensity", "<u2"), ("flags", "u1"),
                 ("classification", "u1"), ("scan_angle_rank", "i1"), ("user_data", "u1"), ("point_source_id", "<u2")]
_LAS_CORE_NEW = [("X", "<i4"), ("Y", "<i4"), ("Z", "<i4"), ("intensity", "<u2"), ("flags", "u1"), ("flags2", "u1"),
                 ("classification", "u1"), ("user_data", "u1"), ("scan_angle", "<i2"), ("point_source_id", "<u2"),
                 ("gps_time", "<f8")]
_GPS = [("gps_time", "<f8")]
_RGB = [("red", "<u2"), ("green", "<u2"), ("blue", "<u2")]
_NIR = [("nir", "<u2")]
_WAVE = [("wave_desc", "u1"), ("wave_offset", "<u8"), ("wave_size", "<u4"), ("wave_loc", "<f4"), ("wave_xt", "<f4"),
         ("wave_yt", "<f4"), ("wave_zt", "<f4")]
LAS_FORMATS = {0: _LAS_CORE_OLD, 1: _LAS_CORE_OLD + _GPS, 2: _LAS_CORE_OLD + _RGB, 3: _LAS_CORE_OLD + _GPS + _RGB,
               4: _LAS_CORE_OLD + _GPS + _WAVE, 5: _LAS_CORE_OLD + _GPS + _RGB + _WAVE, 6: _LAS_CORE_NEW,
               7: _LAS_CORE_NEW + _RGB, 8: _LAS_CORE_NEW + _RGB + _NIR, 9: _LAS_CORE_NEW + _WAVE,
               10: _LAS_CORE_NEW + _RGB + _NIR + _WAVE}


def read_las(path):
    """Uncompressed LAS -> (pos float64 [n, 3] in file units, dict of per-point dimensions)."""
    with open(path, "rb") as f:
        head = f.read(375)
        if head[:4] != b"LASF":
            raise ValueError(f"{path}: not a LAS file")
        minor = head[25]
        header_size, offset_to_points = struct.unpack_from("<HI", head, 94)
        fmt_raw, rec_len, legacy_count = struct.unpack_from("<BHI", head, 104)
        if fmt_raw & 0x80 or fmt_raw & 0x40:
            raise ValueError(f"{path}: LAZ-compressed point records need laspy + a LAZ backend (not available here)")
        fmt = fmt_raw & 0x3F
        if fmt not in LAS_FORMATS:
            raise ValueError(f"{path}: unsupported point format {fmt}")
        sx, sy, sz, ox, oy, oz = struct.unpack_from("<6d", head, 131)
        count = legacy_count
        if minor >= 4 and header_size >= 375:
            count64 = struct.unpack_from("<Q", head, 247)[0]
            count = count64 or legacy_count
        base = np.dtype(LAS_FORMATS[fmt])
        if rec_len < base.itemsize:
            raise ValueError(f"{path}: record length {rec_len} < {base.itemsize} of point format {fmt}")
        dt = np.dtype({"names": base.names, "formats": [base.fields[n][0] for n in base.names],
                       "offsets": [base.fields[n][1] for n in base.names], "itemsize": rec_len})   # extra bytes skipped
        f.seek(offset_to_points)
        rec = np.fromfile(f, dtype=dt, count=count)
    if len(rec) != count:
        raise ValueError(f"{path}: header announces {count} points, file holds {len(rec)}")
    pos = np.stack([rec["X"] * sx + ox, rec["Y"] * sy + oy, rec["Z"] * sz + oz], 1)
    dims = {n: rec[n] for n in rec.dtype.names if n not in ("X", "Y", "Z")}
    fl = rec["flags"]
    if fmt < 6:
        dims["return_number"], dims["number_of_returns"] = fl & 7, (fl >> 3) & 7
        dims["scan_angle"] = rec["scan_angle_rank"]
    else:
        dims["return_number"], dims["number_of_returns"] = fl & 15, (fl >> 4) & 15
    dims["x"], dims["y"], dims["z"] = pos[:, 0], pos[:, 1], pos[:, 2]
    return pos, dims


def write_las(path, pos, scale=0.001, point_format=1, **dims):
    """Minimal LAS 1.2 writer (point formats 0-3) — fixtures and round-trip tests."""
    pos = np.asarray(pos, dtype=np.float64)
    off = np.floor(pos.min(0)) if len(pos) else np.zeros(3)
    dt = np.dtype(LAS_FORMATS[point_format])
    rec = np.zeros(len(pos), dtype=dt)
    q = np.rint((pos - off) / scale).astype(np.int64)
    rec["X"], rec["Y"], rec["Z"] = q[:, 0], q[:, 1], q[:, 2]
    rn = np.asarray(dims.pop("return_number", np.ones(len(pos))), dtype=np.uint8)
    nr = np.asarray(dims.pop("number_of_returns", np.ones(len(pos))), dtype=np.uint8)
    rec["flags"] = (rn & 7) | ((nr & 7) << 3)
    for k, v in dims.items():
        rec[k] = v
    head = bytearray(227)
    head[:4] = b"LASF"
    head[24], head[25] = 1, 2
    struct.pack_into("<HI", head, 94, 227, 227)
    struct.pack_into("<BHI", head, 104, point_format, dt.itemsize, len(pos))
    struct.pack_into("<6d", head, 131, scale, scale, scale, *off)
    mx, mn = (pos.max(0), pos.min(0)) if len(pos) else (np.zeros(3), np.zeros(3))
    struct.pack_into("<6d", head, 179, mx[0], mn[0], mx[1], mn[1], mx[2], mn[2])
    with open(path, "wb") as f:
        f.write(bytes(head))
        rec.tofile(f)


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def read_ply(path):
    """First element of an ascii / binary_little_endian PLY file -> structured array of its scalar properties."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first = None, None, [], False
        while True:
            line = f.readline().decode("ascii").strip().split()
            if not line:
                continue
            if line[0] == "format":
                fmt = line[1]
            elif line[0] == "element":
                if count is not None:
                    in_first = False
                else:
                    count, in_first = int(line[2]), True
            elif line[0] == "property" and in_first:
                if line[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((line[2], _PLY_TYPES[line[1]]))
            elif line[0] == "end_header":
                break
        if fmt == "ascii":
            arr = np.loadtxt(f, max_rows=count, ndmin=2)
            out = np.zeros(count, dtype=[(n, "<" + t if t[1:] != "1" else t) for n, t in props])
            for i, (n, _) in enumerate(props):
                out[n] = arr[:, i]
            return out
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: PLY format '{fmt}' is not supported")
        return np.fromfile(f, dtype=[(n, "<" + t if t[1:] != "1" else t) for n, t in props], count=count)


def read_pt(pt_file, feature_cols=(), delimiter=","):
    """las_dataset.py:32-71: (pos [n,3], features [n,F] or None, crs).  crs is always None (pyproj is not used here)."""
    feature_cols = list(feature_cols)
    suffix = Path(pt_file).suffix.lower()
    if suffix == ".laz":
        raise ValueError(f"{pt_file}: LAZ needs laspy with a LAZ backend; decompress to .las first")
    if suffix == ".las":
        pos, dims = read_las(pt_file)
        feats = np.stack([np.asarray(dims[c]) for c in feature_cols], 1) if feature_cols else None
    elif suffix == ".ply":
        v = read_ply(pt_file)
        pos = np.stack([v["x"], v["y"], v["z"]], 1)
        feats = np.stack([v[c] for c in feature_cols], 1) if feature_cols else None
    else:   # headerless CSV, first three columns are the position (the reference's assumption)
        import pandas as pd
        df = pd.read_csv(pt_file, header=None, delimiter=delimiter, dtype=np.float32, skip_blank_lines=True)
        pos = df.values[:, :3]
        feats = df[feature_cols].values if feature_cols else None
    return pos, feats, None


# ------------------------------------------------------------------------------------------------ samples
def make_sample(pos, features, y_reg, area_name, label_idx, stats=()):
    """The per-plot record of covert_to_data_ (las_dataset.py:477-512); NaN targets are masked."""
    y = torch.as_tensor(np.asarray(y_reg, dtype=np.float32))
    return dict(pos=torch.as_tensor(np.asarray(pos, dtype=np.float32)),
                x=None if features is None else torch.as_tensor(np.asarray(features, dtype=np.float32)),
                y_reg=y, y_reg_mask=~torch.isnan(y), area_name=str(area_name), label_idx=[int(label_idx)],
                stats=torch.as_tensor(np.asarray(stats, dtype=np.float32)))


def load_sample(path):
    obj = torch.load(path, weights_only=False)
    if isinstance(obj, dict):
        return obj
    keys = ("pos", "x", "y_reg", "y_reg_mask", "area_name", "label_idx", "stats")   # a torch_geometric Data object
    return {k: getattr(obj, k, None) for k in keys}


class Las:
    """Processed samples of one split: ``get(idx)`` like las_dataset.py:421-435 (memory cache, is_double flag)."""

    def __init__(self, processed_files, in_memory=True, transform=None):
        self.processed_file_names = list(processed_files)
        self.in_memory, self.transform = in_memory, transform
        self.memory, self.prev_idx = {}, None

    def __len__(self):
        return len(self.processed_file_names)

    def get(self, idx):
        if self.in_memory and idx in self.memory:
            data = dict(self.memory[idx])
        else:
            data = load_sample(self.processed_file_names[idx])
            if self.in_memory:
                self.memory[idx] = dict(data)
        data["is_double"] = self.prev_idx == idx
        self.prev_idx = idx
        return data

    def __getitem__(self, idx):
        data = self.get(idx)
        return self.transform(data) if self.transform is not None else data


class LasDataset:
    """What the models / tracker read from the reference's LasDataset, over processed sample files.

    splits: {"train": [paths], "val": [...], "test": [...]};  targets: config.NFI_TARGETS-like mapping."""

    def __init__(self, splits, targets, feature_dimension, num_points=16000, in_memory=True):
        from .config import Opt
        self.targets = targets
        self.reg_targets = [t for t in targets if targets[t]["task"] == "regression"]
        self.reg_targets_idx = np.array([targets[t]["task"] == "regression" for t in targets])
        self.num_reg_classes = int(self.reg_targets_idx.sum())
        self.num_classes = self.num_reg_classes
        self.has_reg_targets = self.num_reg_classes > 0
        self.feature_dimension = feature_dimension
        self.double_batch = False
        self.dataset_opt = Opt(fixed=Opt(num_points=num_points))
        self.sets = {k: Las(v, in_memory) for k, v in splits.items()}
        self.train_dataset, self.val_dataset, self.test_dataset = (self.sets.get(k) for k in ("train", "val", "test"))
        labels = {k: [(s["area_name"], s["y_reg"].double().numpy()) for s in (ds.get(i) for i in range(len(ds)))]
                  for k, ds in self.sets.items()}
        self.areas = OrderedDict((a, None) for k in labels for a, _ in labels[k])
        self._stat_cache = {}
        self._labels = labels

    def _stat(self, fn):
        """{area | "total": {split: array over targets}} like get_stat_targets_ (NaN-aware)."""
        if fn not in self._stat_cache:
            out = OrderedDict((a, {}) for a in ["total"] + list(self.areas))
            for split, rows in self._labels.items():
                for a in out:
                    ys = [y for area, y in rows if a == "total" or area == a]
                    if ys:
                        with np.errstate(all="ignore"):
                            out[a][split] = getattr(np, "nan" + fn)(np.stack(ys), 0)
            self._stat_cache[fn] = out
        return self._stat_cache[fn]

    def get_mean_targets(self):
        return self._stat("mean")

    def get_std_targets(self):
        return self._stat("std")

    def get_min_targets(self):
        return self._stat("min")

    def get_max_targets(self):
        return self._stat("max")


def collate(samples):
    """Plot samples -> the stacked point batch ``set_input`` / the device transform chain take (a stand-in for
    torch_geometric's Batch.from_data_list over the fields this path reads)."""
    from .synthetic import PlotBatch
    n = [len(s["pos"]) for s in samples]
    batch = torch.repeat_interleave(torch.arange(len(samples)), torch.tensor(n))
    x = None if samples[0]["x"] is None else torch.cat([s["x"] for s in samples])
    out = PlotBatch(batch, None, x, torch.cat([s["pos"] for s in samples]), torch.stack([s["y_reg"] for s in samples]),
                    torch.stack([s["y_reg_mask"] for s in samples]), len(samples))
    out.area_name = [s["area_name"] for s in samples]
    return out


def process_plot_files(files, labels, out_dir, feature_cols=(), delimiter=","):
    """Plot point files -> processed ``.pt`` samples (positions moved to the plot's minimum corner in z and to the label's
    plot centre in xy when given, as center_pos does: las_dataset.py:525-531).  labels: list of dicts with ``y_reg``
    (sequence), ``area_name`` and optional ``x``, ``y`` (plot centre)."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for i, (f, lab) in enumerate(zip(files, labels)):
        pos, feats, _ = read_pt(f, feature_cols, delimiter)
        pos = np.asarray(pos, dtype=np.float64)
        centre = pos.min(0, keepdims=True)
        if "x" in lab and "y" in lab:
            centre[0, 0], centre[0, 1] = lab["x"], lab["y"]
        sample = make_sample(pos - centre, feats, lab["y_reg"], lab.get("area_name", "area"), i)
        p = os.path.join(out_dir, f"sample_{i}.pt")
        torch.save(sample, p)
        paths.append(p)
    return paths
