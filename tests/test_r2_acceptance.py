"""R2 acceptance (BASELINE.json north_star "test-set R2 within +-0.005"; SURVEY.md §8c/§8d scope it to the HIP path
against this repo's CPU restatement on one synthetic labelled set): the HIP path trains MSENet14 on the schedule of
tests/golden/make_r2_cpu_leg.py (256 train / 64 val plots x 4000 points, 10 epochs = one cosine cycle, identical initial
weights, batch order and drop-path draws) and its validation R2 / RMSE — metric definitions of
metrics/meters/r2meter.py:15-26 and instance_tracker.py:85-87 — are compared with the committed CPU leg
(oracle/sparse_ref.py, torch-CPU fp32)."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "r2_cpu_leg.json")


def test_cpu_leg_fixture_is_sane():
    ref = json.load(open(GOLDEN))
    assert ref["config"]["model"] == "SENet14" and len(ref["history"]) == ref["config"]["epochs"]
    assert len(ref["val_predictions"]) == ref["config"]["val"]
    # the run learned something: the last epoch is far better than predicting the mean
    assert min(ref["final"]["r2_bs"]) > 0.0


@pytest.mark.gpu
def test_r2_within_0p005_of_cpu_leg(device):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from train_eval import acceptance_gpu_leg
    ref = json.load(open(GOLDEN))
    got = acceptance_gpu_leg(ref["config"], device)
    for t in range(2):
        assert abs(got["final"]["r2_bs"][t] - ref["final"]["r2_bs"][t]) <= 0.005, (t, got["final"], ref["final"])
        assert abs(got["final"]["rmse_bs"][t] - ref["final"]["rmse_bs"][t]) <= 0.02 * ref["final"]["rmse_bs"][t]
