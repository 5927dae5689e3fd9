"""Pins oracle/kpconv_ref.py against golden vectors produced by the reference's own Python
(tests/golden/kpconv_layer_golden.npz, generator: tests/golden/make_kpconv_layer_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import kpconv_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kpconv_layer_golden.npz")
CFG = dict(first_subsampling_dl=0.032, conv_radius=2.5, KP_extent=1.0, in_features_dim=3, first_features_dim=16,
           architecture=["simple", "resnetb", "resnetb_strided", "resnetb", "global_sum"], batch_norm_momentum=0.02)


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def T(a, grad=False):
    t = torch.from_numpy(np.asarray(a))
    if t.dtype == torch.int32:
        t = t.long()
    return t.requires_grad_(grad) if grad else t


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30)


def test_kpconv_layer_forward_backward(g):
    for tag, q, idx in (("L", g["points0"], g["neighbors0"]), ("S", g["points1"], g["pools0"])):
        x, w = T(g[f"{tag}_x"], True), T(g["L_w"], True)
        y = R.kpconv(T(q), T(g["points0"]), T(idx), x, T(g["L_kp"]), w, float(g["L_ext"]))
        y.backward(T(g[f"{tag}_g"]))
        assert close(y.detach(), g[f"{tag}_y"])
        assert close(x.grad, g[f"{tag}_dx"])
        assert close(w.grad, g[f"{tag}_dw"])


def test_pool_helpers(g):
    assert np.array_equal(R.max_pool(T(g["P_x"]), T(g["pools0"])).numpy(), g["P_maxpool"])
    assert close(R.global_sum(T(g["P_x"]), g["lens0"]), g["P_globalsum"], 1e-6)


def test_kpcnn_end_to_end(g):
    sd = {k[5:]: T(g[k], grad=("running" not in k and "num_batches" not in k and "kernel_points" not in k))
          for k in g.files if k.startswith("N_sd/")}
    batch = dict(features=T(g["N_feats"]), points=[T(g["points0"]), T(g["points1"])],
                 neighbors=[T(g["neighbors0"]), T(g["neighbors1"])], pools=[T(g["pools0"])],
                 lengths=[g["lens0"], g["lens1"]])
    upd = {}
    y = R.kpcnn_forward(sd, CFG, batch, training=True, update=upd)
    y.backward(T(g["N_g"]))
    assert close(y.detach(), g["N_y"], 1e-4)
    n_checked = 0
    for k in g.files:
        if k.startswith("N_grad/"):
            name = k[7:]
            assert close(sd[name].grad, g[k], 2e-3), name
            n_checked += 1
        if k.startswith("N_after/"):
            assert close(upd[k[8:]], g[k], 1e-5), k
    assert n_checked > 20
