"""ORACLE (test infrastructure only). numpy restatement of GridSampling3D(mode="last", quantize_coords=True):
torch_points3d/core/data_transform/grid_transform.py:112-128 (+ shuffle_data :22-29, group_data :32-80).

PARITY UNPINNED: the clustering arithmetic lives in third-party packages absent from /root/reference and not
installable offline — torch_cluster.grid_cluster (pyg::pytorch-cluster, version unpinned, env.yml:43) and
torch_geometric.nn.pool.consecutive.consecutive_cluster (pyg 2.3.1, env.yml:42); the reference holds no test for it.
Restated from their documented behaviour:
  coords  = round(pos / size)                    (torch.round: half to even, float32)              :116
  cluster = sum_d floor((c_d - min_d) / 1) * prod_{e<d} (floor((max_e - min_e) / 1) + 1)          :118 (x fastest)
  unique sorted clusters; representative = the LAST shuffled point of each cluster (scatter_ overwrite) :121
  data[key] = item[unique_pos_indices]; coords = coords[unique_pos_indices].int()                  :123-125
"""
import numpy as np


def grid_sampling_last(pos: np.ndarray, perm: np.ndarray, size: float):
    """pos float32 [N,3], perm int [N] (shuffle: shuffled[i] = original[perm[i]]).
    Returns (coords int32 [M,3], keep int64 [M] indices into the ORIGINAL order), ascending cluster id."""
    p = np.asarray(pos, dtype=np.float32)[perm]
    c = np.rint(p / np.float32(size)).astype(np.float32)
    lo, hi = c.min(0), c.max(0)
    span = (np.floor(hi - lo) + 1).astype(np.int64)
    rel = np.floor(c - lo).astype(np.int64)
    cluster = rel[:, 0] + span[0] * (rel[:, 1] + span[1] * rel[:, 2])
    uniq, inv = np.unique(cluster, return_inverse=True)
    last = np.full(len(uniq), -1, dtype=np.int64)
    for i, u in enumerate(inv):   # sequential scatter: later shuffled points overwrite earlier ones
        last[u] = i
    return c[last].astype(np.int32), np.asarray(perm, dtype=np.int64)[last]


def batch_grid_sampling_last(pos, lengths, perms, size):
    """Per-cloud application + stacking (what the DataLoader collation does)."""
    coords, keep, lens, off = [], [], [], 0
    for n, perm in zip(lengths, perms):
        c, k = grid_sampling_last(pos[off:off + n], perm, size)
        coords.append(c)
        keep.append(k + off)
        lens.append(len(k))
        off += n
    return np.concatenate(coords), np.concatenate(keep), np.asarray(lens, dtype=np.int64)
