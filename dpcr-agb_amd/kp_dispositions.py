"""Kernel-point dispositions for KPConv (initialisation only — not on the hot path).

Restates the reference's procedure (torch_points3d/modules/KPConv/kernel_points.py:204-335 potential optimisation,
:338-413 load_kernels) with the same ``np.random`` consumption, so a seeded construction reproduces the reference's
kernel points: 100 candidate dispositions of K points are relaxed under a mutual-repulsion + centring potential,
the one with the smallest final gradient is kept, rescaled to ``ratio`` of the radius, jittered by N(0, 0.01),
scaled by the radius and rotated about z by a random angle.  Trained models carry their kernel points in the
state_dict (``block_ops.<i>.KPConv.kernel_points``), which takes precedence over anything generated here.
The optimisation is cached per (K, fixed) within the process: the reference repeats it for every layer (~8 s each).
"""
import numpy as np

_CACHE = {}


def _relax(num_points, num_kernels, dimension, fixed, ratio=0.66):
    radius0, diameter0 = 1.0, 2.0
    moving_factor, decay, thresh, clip = 1e-2, 0.9995, 1e-5, 0.05 * 1.0
    total = num_kernels * num_points
    kp = np.random.rand(total - 1, dimension) * diameter0 - radius0
    while kp.shape[0] < total:
        fresh = np.random.rand(total - 1, dimension) * diameter0 - radius0
        kp = np.vstack((kp, fresh))
        kp = kp[np.sum(np.power(kp, 2), axis=1) < 0.5 * radius0 * radius0, :]
    kp = kp[:total, :].reshape((num_kernels, num_points, -1))
    if fixed == "center":
        kp[:, 0, :] *= 0
    elif fixed == "verticals":
        kp[:, :3, :] *= 0
        kp[:, 1, -1] += 2 * radius0 / 3
        kp[:, 2, -1] -= 2 * radius0 / 3
    skip = {"center": 1, "verticals": 3}.get(fixed, 0)
    last_norms = np.zeros((num_kernels, num_points))
    final = np.zeros(num_kernels)
    for _ in range(10001):
        a, b = np.expand_dims(kp, 2), np.expand_dims(kp, 1)
        d2 = np.sum(np.power(a - b, 2), axis=-1)
        grad = np.sum((a - b) / (np.power(np.expand_dims(d2, -1), 3 / 2) + 1e-6), axis=1) + 10 * kp
        if fixed == "verticals":
            grad[:, 1:3, :-1] = 0
        norms = np.sqrt(np.sum(np.power(grad, 2), axis=-1))
        final = np.max(norms, axis=1)
        if np.max(np.abs(last_norms[:, skip:] - norms[:, skip:])) < thresh:
            break
        last_norms = norms
        move = np.minimum(moving_factor * norms, clip)
        if fixed in ("center", "verticals"):
            move[:, 0] = 0
        kp -= np.expand_dims(move, -1) * grad / np.expand_dims(norms + 1e-6, -1)
        moving_factor *= decay
    r = np.sqrt(np.sum(np.power(kp, 2), axis=-1))
    kp *= ratio / np.mean(r[:, 1:])
    return kp, final


def kernel_disposition(radius, num_kpoints=15, dimension=3, fixed="center", cache=True):
    """float32 [K, dimension] kernel points for a convolution of the given radius."""
    if num_kpoints > 30:
        raise NotImplementedError("more than 30 kernel points (the reference's Lloyd path) is not on the AGB path")
    key = (num_kpoints, dimension, fixed)
    if cache and key in _CACHE:
        base = _CACHE[key]
    else:
        cands, final_grad = _relax(num_kpoints, 100, dimension, fixed)
        base = cands[int(np.argmin(final_grad))]
        if cache:
            _CACHE[key] = base
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    if dimension == 3:
        rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    else:
        rot = np.array([[c, -s], [s, c]], dtype=np.float32)
    pts = base + np.random.normal(scale=0.01, size=base.shape)
    return np.matmul(radius * pts, rot).astype(np.float32)
