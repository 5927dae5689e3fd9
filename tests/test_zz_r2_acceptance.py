"""R2 acceptance (BASELINE.json north_star "test-set R2 within +-0.005"; SURVEY.md §8c/§8d scope it to the HIP path against
this repo's CPU restatement on one synthetic labelled set; metric definitions of metrics/meters/r2meter.py:15-26 and
instance_tracker.py:85-87, pinned by tests/test_metrics.py).  Collected LAST (tests/conftest.py): a training-outcome test
never sits in front of a kernel parity file.

Protocol = the reference's own: every R2 it publishes is the MEDIAN OF 5 TRIALS (README.md:24-56); its evaluation is seeded to
be repeatable (eval.py:12-21).  The acceptance set and schedule are those of tests/golden/make_r2_cpu_leg.py (MSENet14,
reference recipe, 256 / 128 plots, 150 epochs = the end of the fourth cosine cycle, calibrate_bn, running-statistics
evaluation): both targets plateau at R2 ~ 0.77.

Since round 4 EVERY HIP leg — fp32, bf16 operands, bf16 operands on bf16 row storage — trains with fixed-order weight-gradient
sums (KernelOptions.deterministic_wgrad, honoured in every operand precision): a trial is a pure function of (tree, seed), the
same on every run and every MI355X.  Three kinds of statement, kept apart (round 5):

  * ACCEPTANCE, one-sided bounds on the committed 13 x 3 per-trial table tests/golden/r2_hip_expected.json
    (tools/make_r2_hip_expected.py) against the 13 committed CPU trials (CPU suite, nothing is drawn): plateau regime, paired-seed
    bias, gap of the medians within 0.005 + 2 s.e. — and, on the GPU, the sharp form of "+-0.005": the same weights give the same
    R2 under either implementation (1e-8 in fp32);
  * REGRESSION PIN (labelled as such): the GPU suite reproduces the protocol's five seeds of every leg of that table to 1e-9.
    It says the tree still computes what the table holds, not that the numbers are right; a change of a summation order — or a
    ROCm update that reorders a reduction — needs `python tools/make_r2_hip_expected.py` (seeds 5-12 are reproduced there);
  * REPORT: whether the bare |median_HIP - median_CPU| <= 0.005 happens to hold per leg is PRINTED, never asserted: +-0.005 is
    0.4 standard errors of a difference of two medians of 13, every leg met and missed it across three trees of round 4 that
    differed only in a summation order — a kernel choice must not hang on that coin.
"""
import glob
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "r2_cpu_leg.json")
HIP_EXPECTED = os.path.join(ROOT, "tests", "golden", "r2_hip_expected.json")
LEGS = ("fp32", "bf16", "bf16rows")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _ref():
    with open(GOLDEN) as f:
        ref = json.load(f)
    assert "trials" in ref, "tests/golden/r2_cpu_leg.json predates the 5-trial protocol: run tests/golden/make_r2_cpu_leg.py"
    return ref


def _cpu_trials():
    """R2 (biomass, volume) of the committed CPU trials, by seed: tests/golden/r2_cpu_trials/trial_<seed>.json."""
    out = {}
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "r2_cpu_trials", "trial_*.json")):
        d = json.load(open(f))
        out[int(d["trial"])] = d["final"]["r2_rs"]
    return out


def _hip_expected():
    with open(HIP_EXPECTED) as f:
        return json.load(f)


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
    # the first five per-seed files are the protocol's five trials
    cpu = _cpu_trials()
    assert sorted(cpu) == list(range(13))
    assert np.allclose(np.array([cpu[t] for t in range(5)]), r2)


def _median_se(r2):
    """Standard error of the median of n normal draws: 1.2533 s / sqrt(n), per target."""
    return 1.2533 * r2.std(0, ddof=1) / math.sqrt(len(r2))


def _gap_table():
    exp, cpu = _hip_expected(), _cpu_trials()
    assert exp["config"] == _ref()["config"] and exp["trials"] == 13
    c = np.array([cpu[t] for t in range(13)])
    rows = {}
    for leg in LEGS:
        h = np.array(exp["legs"][leg])
        assert h.shape == (13, 2), leg
        d = h - c
        rows[leg] = dict(hip=h, median=np.median(h, 0), gap13=np.median(h, 0) - np.median(c, 0),
                         gap5=np.median(h[:5], 0) - np.median(c[:5], 0),
                         se13=np.sqrt(_median_se(h) ** 2 + _median_se(c) ** 2),
                         paired_mean=d.mean(0), paired_se=d.std(0, ddof=1) / math.sqrt(13), paired_sd=d.std(0, ddof=1))
    return c, rows


def test_r2_gap_on_committed_trials():
    """The HIP-vs-CPU gap on the committed, reproducible per-trial numbers (no GPU needed: the GPU half of this file checks
    that the tree still produces exactly these numbers).

    ASSERTED, one-sided (an improvement never fails; each a deterministic fact of the tree — nothing is drawn here):
      (a) every leg is in the plateau regime: median R2 >= 0.6 on both targets, every trial finite and > 0.5;
      (b) paired by seed (same initial weights, batch order, drop-path draws on both sides), the mean HIP - CPU difference
          over the 13 seeds is within +-0.02 on both targets — a bound on BIAS (the s.e. of that mean is 0.004-0.008: the sd
          of a paired difference is 0.02-0.03, rounding-level differences send a trial to another of equally good minima,
          DESIGN.md section 6).  A kernel that costs 0.03 of R2 fails here;
      (c) the gap of the 13-seed medians is within 0.005 + 2 s.e. of a difference of two medians of 13 (~0.02-0.03).
    PRINTED, not asserted: the outcome of the bare north-star criterion |median_HIP - median_CPU| <= 0.005 per leg and target,
    and the five-seed medians of the reference's protocol (with sd 0.015-0.03 a bare +-0.005 between two medians is met by chance
    about one time in three for identical implementations).  The sharp, draw-free form of the criterion is
    test_r2_same_weights."""
    c, rows = _gap_table()
    print()
    print(f"cpu (oracle, fp32)  13-seed median {np.median(c, 0).round(4).tolist()}  sd {c.std(0, ddof=1).round(4).tolist()}  "
          f"5-seed median {np.median(c[:5], 0).round(4).tolist()}")
    for leg, r in rows.items():
        h = r["hip"]
        print(f"hip {leg:9s} 13-seed median {r['median'].round(4).tolist()}  sd {h.std(0, ddof=1).round(4).tolist()}  gap "
              f"{r['gap13'].round(4).tolist()} = {(np.abs(r['gap13']) / r['se13']).round(2).tolist()} s.e.;  5-seed gap "
              f"{r['gap5'].round(4).tolist()};  paired HIP - CPU mean {r['paired_mean'].round(4).tolist()} +- "
              f"{r['paired_se'].round(4).tolist()}")
        assert np.isfinite(h).all() and (h > 0.5).all() and (r["median"] >= 0.6).all(), leg                  # (a)
        assert (np.abs(r["paired_mean"]) <= 0.02).all(), (leg, r["paired_mean"])                             # (b)
        assert (np.abs(r["gap13"]) <= 0.005 + 2.0 * r["se13"]).all(), (leg, r["gap13"], r["se13"])           # (c)
        print(f"    bare |median gap| <= 0.005 on this table (reported, not a gate): "
              f"{['met' if v else 'NOT met' for v in (np.abs(r['gap13']) <= 0.005)]}")
    assert (np.median(c, 0) >= 0.6).all()


@pytest.mark.gpu
@pytest.mark.parametrize("leg", LEGS)
def test_r2_regression_pin_five_seeds(device, leg):
    """REGRESSION PIN, not an acceptance statement: seeds 0-4 (the reference protocol's five, README.md:24-56) of this leg on
    the HIP path equal tests/golden/r2_hip_expected.json to 1e-9 — the committed table is what this tree computes, on this box
    too.  Fails after ANY change of a summation order (kernel work, a ROCm update): regenerate with
    tools/make_r2_hip_expected.py, which also reproduces seeds 5-12.  (~11 s per trial.)"""
    from train_eval import acceptance_data, acceptance_gpu_trial
    exp = _hip_expected()
    cfg = exp["config"]
    data = acceptance_data(cfg, device)
    want = np.array(exp["legs"][leg])[:5]
    got = np.array([acceptance_gpu_trial(cfg, t, device, leg, data)["final"]["r2_rs"] for t in range(5)])
    print()
    for t in range(len(got)):
        print(f"hip {leg} seed {t}: R2 {got[t].tolist()}  expected {want[t].tolist()}  d = {(got[t] - want[t]).tolist()}")
    assert np.abs(got - want).max() <= 1e-9, (leg, np.abs(got - want).max(axis=1).tolist())


@pytest.mark.gpu
@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("AGB_R2_FULL", "0") == "0", reason="the full 13-seed reproduction of every leg (~7 GPU-minutes): "
                    "AGB_R2_FULL=1; the table is regenerated on the round's last tree (its `tree` field names the commit)")
@pytest.mark.parametrize("leg", LEGS)
def test_r2_regression_pin_all_thirteen_seeds(device, leg):
    """Every row the CPU half of this file consumes (seeds 0-12 of every leg) reproduced by THIS tree to 1e-9 — the slow form
    of test_r2_regression_pin_five_seeds (round-5 advisor finding: rows 5-12 were pinned by no test).  Run with AGB_R2_FULL=1
    after kernel work; tools/make_r2_hip_expected.py regenerates the table and records the commit it was generated on."""
    from train_eval import acceptance_data, acceptance_gpu_trial
    exp = _hip_expected()
    cfg = exp["config"]
    data = acceptance_data(cfg, device)
    want = np.array(exp["legs"][leg])
    got = np.array([acceptance_gpu_trial(cfg, t, device, leg, data)["final"]["r2_rs"] for t in range(len(want))])
    assert np.abs(got - want).max() <= 1e-9, (leg, np.abs(got - want).max(axis=1).tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("leg,r2_tol,rmse_rtol", [("fp32", 1e-5, 1e-5), ("bf16", 5e-3, 2e-2), ("bf16rows", 5e-3, 2e-2)])
def test_r2_same_weights(device, leg, r2_tol, rmse_rtol):
    """One HIP trial's trained weights (30 epochs; any trained state serves), evaluated (i) by the HIP path in the leg's own
    precision and (ii) by the CPU restatement in fp32 on the same validation plots — what eval.py computes must not depend
    on which implementation computes it.  fp32: measured 4e-8 (the sharp form of +-0.005); bf16 / bf16 rows: the HIP
    forward rounds operands (and rows) to 8 significant bits, the bar is the criterion's own 0.005 on R2 and 2 % on RMSE."""
    from train_eval import acceptance_data, acceptance_gpu_trial
    from oracle import sparse_ref as R
    from dpcr_agb_amd.metrics import RegressionMeter
    cfg = dict(_ref()["config"], epochs=30, eval_every=30)
    keep = {}
    got = acceptance_gpu_trial(cfg, 0, device, leg, acceptance_data(cfg, device), keep=keep)
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
        print(f"same weights ({leg}), target {t}: R2 hip {got['final']['r2_rs'][t]:.6f} cpu {cpu['r2'][t]:.6f} (d = {d:+.2e}); "
              f"RMSE hip {got['final']['rmse_rs'][t]:.4f} cpu {cpu['rmse'][t]:.4f}")
        assert abs(d) <= r2_tol, (leg, t, d)
        assert abs(got["final"]["rmse_rs"][t] - cpu["rmse"][t]) <= rmse_rtol * cpu["rmse"][t]


@pytest.mark.gpu
def test_default_atomic_weight_gradients_reach_the_plateau(device):
    """The DEFAULT weight-gradient kernels (fp32 atomic accumulation: the faster form inside the step, not reproducible from
    run to run) are what bench.py times; one trial with them lands in the same plateau as the reproducible trials.  Not a
    sharp gate by construction (the outcome is a draw): plateau regime and inside the committed trials' range +-0.08 (the
    trial-to-trial sd is 0.027; seed-0 runs of rounds 3-4 with atomics: 0.748, 0.763, 0.767)."""
    from train_eval import acceptance_data, acceptance_gpu_trial
    exp = _hip_expected()
    cfg = exp["config"]
    r2 = np.array(acceptance_gpu_trial(cfg, 0, device, "fp32", acceptance_data(cfg, device), deterministic=False)["final"]["r2_rs"])
    h = np.array(exp["legs"]["fp32"])
    print(f"default (atomic) weight gradients, seed 0: R2 {r2.round(4).tolist()}; reproducible trials span "
          f"{h.min(0).round(4).tolist()} .. {h.max(0).round(4).tolist()}")
    assert (r2 >= 0.6).all() and (r2 >= h.min(0) - 0.08).all() and (r2 <= h.max(0) + 0.08).all()
