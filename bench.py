"""Headline benchmark: MSENet14 biomass-regression TRAINING throughput (plots/s) on synthetic 16k-point plots.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W)

One step = one pass of the hot path over one batch: set_input (coordinate hash + kernel maps on device),
forward, backward, gradient all-reduce (N>1), clip, AdaBelief, LR schedule — BASELINE.json config
"MSENet14 sparse-voxel, 1xMI355X, 0.5m voxel, batch=32" per GPU (weak scaling).  Inputs are pre-voxelised and
resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0  # bf16 MFMA dense peak (split-bf16x3 issues 3 bf16 MFMAs per product: priced at 2500 / 3)
PRECISION = "fp32"         # operand precision of the convolution MFMAs (--precision)
PMC_CONFIG = None          # "config5" for the MSENet50 bf16-row line: which committed PMC pass `roofline.traffic` reads


def mfma_peak_tf():
    return {"fp32": MFMA_F32_PEAK_TF, "bf16": MFMA_BF16_PEAK_TF, "bf16x3": MFMA_BF16_PEAK_TF / 3.0}[PRECISION]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # (1.8 s of timed region: the 40-step default of rounds 1-3 was 0.4 s)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="plots per GPU")
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--model", default="SENet14")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="operand precision of the convolution MFMAs (fp32 accumulate; BN / SE / index kernels fp32). The "
                         "headline (BASELINE config 4) is fp32; bf16 is BASELINE config 5's mode")
    ap.add_argument("--bf16-rows", action="store_true",
                    help="with --precision bf16: store every activation / gradient row matrix of the backbone in bf16 "
                         "(KernelOptions.bf16_activations; accumulators, statistics, parameters fp32)")
    ap.add_argument("--pool", type=int, default=4, help="distinct pre-generated batches cycled per rank")
    ap.add_argument("--features", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--deterministic", action="store_true", help="fixed-order weight-gradient sums (KernelOptions."
                    "deterministic_wgrad): bitwise reproducible steps — what tests compare parameter checksums on; not the "
                    "default the headline is measured with")
    ap.add_argument("--no-prefetch", action="store_true", help="build each batch's coordinate plan inside set_input")
    ap.add_argument("--force-prefetch", action="store_true", help="keep the side-stream input pipeline even where the ranks "
                    "of this node have fewer than 2.5 usable cores each (tools/host_budget.py measures both forms)")
    ap.add_argument("--cpu-plots", type=int, default=1)
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BASELINE configs 2, 3 and 5 (single-GPU leg) that the default N = 1 headline run measures "
                         "afterwards as fresh child processes and reports under `other_configs`")
    ap.add_argument("--reserve-gib", type=int, default=16, help="allocator pool reserved up front on the compute stream "
                    "(half of it again on the input pipeline's side stream); 0 = grow on demand")
    return ap.parse_args()


def conv_cost(rec, pairs):
    """Algorithmic bytes / FLOPs of one convolution launch (activations and weights are fp32 in HBM in every operand
    mode: e = 4).  bytes: SURVEY.md §8(d)'s per-pair formula P*(Cin+Cout)*e + 8P + K^3*Cin*Cout*e; compulsory: its
    output-stationary lower bound N_in*Cin*e + N_out*Cout*e + K^3*Cin*Cout*e + 8P (every row read / written once) — the
    figure an HBM-bound kernel is priced against (the per-pair figure counts each gathered row once per pair although
    the gathers are served by L2 / MALL: it can exceed the HBM peak and is not a bound)."""
    cin, cout, K3 = rec["cin"], rec["cout"], rec["K3"]
    flops = 2.0 * pairs * cin * cout
    byts = pairs * (cin + cout) * 4 + pairs * 8 + K3 * cin * cout * 4
    n_in = rec.get("rows_in") or rec["rows"]
    comp = (n_in * cin + rec["rows"] * cout) * 4 + K3 * cin * cout * 4 + pairs * 8
    return byts, flops, comp


def kernel_of(rec):
    """Name of the HIP kernel a recorded launch ran: what the library itself noted when it launched it (agb_last_kernel,
    csrc/agb_common.h AGB_LAUNCH) — the names rocprof shows — with " (dense)" appended for identity-map products."""
    name = rec.get("kernel") or "unknown"
    return name + (" (dense)" if rec["kind"].endswith("1x1") else "")


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, file it was read from) from the newest committed PMC pass
    (tools/collect_pmc.sh <tag> -> profiles/<tag>_pmc_traffic.json; separate --pmc runs of this same command, corrected as
    MI355X_MICROARCH.md prescribes) — counters cannot be collected from inside a running benchmark — or (None, None)."""
    names = pmc_files()
    for name in names:
        rel = os.path.join("profiles", name)
        try:
            data = json.load(open(os.path.join(ROOT, rel)))["kernels"]
        except Exception:
            continue
        for kname, v in data.items():
            if kname.replace("void ", "").strip() == kernel.split(" (")[0]:
                return round(v["hbm_bytes_per_launch"]), rel
    return None, None


def pmc_files():
    names = [f"{tag}_pmc_traffic.json" for tag in ("r06", "r05", "r04", "r03", "r02", "r01")]
    if PMC_CONFIG:      # (another model / precision than the headline: its own PMC pass, tools/collect_pmc_configs.sh)
        names = [f"{tag}_pmc_traffic_{PMC_CONFIG}.json" for tag in ("r06", "r05", "r04")]
    return names


def pmc_step_bytes():
    """(HBM bytes per training step, file) of the newest committed PMC pass that recorded it (tools/pmc_summary.py: every
    kernel of the profiled run over its steps), or (None, None)."""
    for name in pmc_files():
        rel = os.path.join("profiles", name)
        try:
            v = json.load(open(os.path.join(ROOT, rel))).get("hbm_bytes_per_step")
        except Exception:
            continue
        if v:
            return float(v), rel
    return None, None


def group_profile(prof, table=False):
    """Group the recorded launches by kernel flavour: name -> dict(ms, n, bytes, flops, comp)."""
    groups = {}
    pair_cache = {}

    def pairs_of(rec):
        p = rec["pairs"]
        if p is None:
            return None
        if isinstance(p, int):      # dense 1x1 products: one pair per row
            return p
        if id(p) not in pair_cache:
            pair_cache[id(p)] = int(p.sum().item())
        return pair_cache[id(p)]

    for rec in prof:
        pairs = pairs_of(rec)
        if pairs is None:
            continue
        ms = rec["start"].elapsed_time(rec["end"])
        byts, flops, comp = conv_cost(rec, pairs)
        name = kernel_of(rec)
        g = groups.setdefault(name, dict(ms=0.0, n=0, bytes=0.0, flops=0.0, comp=0.0))
        g["ms"] += ms
        g["n"] += 1
        g["bytes"] += byts
        g["flops"] += flops
        g["comp"] += comp
    if table:   # per-layer table (stderr): where the conv time goes
        layers = {}
        for rec in prof:
            pairs = pairs_of(rec)
            if pairs is None:
                continue
            key = (rec["kind"], rec["K3"], rec["cin"], rec["cout"], rec["rows"] // 1000)
            e = layers.setdefault(key, [0, 0.0, 0.0, 0])
            e[0] += 1
            e[1] += rec["start"].elapsed_time(rec["end"])
            e[2] += 2.0 * pairs * rec["cin"] * rec["cout"]
            e[3] += pairs
        for key, (cnt, ms, fl, pr) in sorted(layers.items(), key=lambda kv: -kv[1][1])[:40]:
            kind, K3, cin, cout, krows = key
            log(f"  {kind:8s} K3={K3:3d} {cin:4d}->{cout:4d} rows~{krows:4d}k  n={cnt:3d}  {ms / cnt * 1e3:8.1f} us/launch  "
                f"{fl / (ms / 1e3) / 1e12:6.1f} TF  density={pr / cnt / (K3 * max(krows, 1) * 1000.0):.2f}")
    return groups


def dominant_kernel(groups):
    return max(groups, key=lambda k: groups[k]["ms"]) if groups else None


def roofline_of(dom, g):
    """Roofline entry of one kernel group over its measured time.  MFMA-bound (FLOPs / compulsory bytes above the
    ridge of the operand mode): achieved = algorithmic FLOP/s against the dense MFMA peak of that mode.  HBM-bound:
    achieved = compulsory bytes/s against the HBM peak (see conv_cost)."""
    secs = g["ms"] / 1e3
    peak = mfma_peak_tf()
    intensity = g["flops"] / g["comp"]
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    traffic, traffic_src = pmc_traffic(dom)
    if intensity >= ridge:
        ach = g["flops"] / secs / 1e12
        roof = dict(bound="mfma", achieved=round(ach, 3), peak=round(peak, 1), unit="TFLOP/s",
                    frac=round(ach / peak, 4), traffic=traffic, traffic_source=traffic_src)
    else:
        ach = g["comp"] / secs / 1e9
        roof = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src)
    roof.update(kernel=dom, launches=g["n"], avg_launch_us=round(g["ms"] / g["n"] * 1e3, 2),
                alg_bytes_per_launch=round(g["bytes"] / g["n"]), compulsory_bytes_per_launch=round(g["comp"] / g["n"]),
                alg_flops_per_launch=round(g["flops"] / g["n"]), flop_per_compulsory_byte=round(intensity, 1),
                ridge=round(ridge, 1))
    return roof


def kernel_summary(groups):
    return {k: dict(total_ms=round(v["ms"], 3), launches=v["n"],
                    compulsory_gbs=round(v["comp"] / (v["ms"] / 1e3) / 1e9, 1),
                    tflops=round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2)) for k, v in groups.items()}


def usable_cores(cap=32):
    """Host cores this process may really use: affinity mask, cgroup CPU quota, capped at `cap` (32 for the CPU legs:
    oversubscribing a quota-limited container with one OpenMP thread per visible core makes them arbitrarily slow;
    cap=None: the uncapped count, what the per-rank host budget is decided on)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n if cap is None else min(n, cap))


def thread_cpu_times():
    """{tid: (name, user + system CPU seconds)} of every thread of this process (Linux /proc)."""
    out = {}
    try:
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            with open(f"/proc/self/task/{tid}/stat") as f:
                raw = f.read()
            name = raw[raw.index("(") + 1:raw.rindex(")")]
            fields = raw[raw.rindex(")") + 2:].split()
            out[tid] = (name, (int(fields[11]) + int(fields[12])) / tick)
    except Exception:
        pass
    return out


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


T_START = time.perf_counter()


def cpu_baseline(args, model_sd, stats, layers=(1, 1, 1, 1)):
    """The oracle (torch-CPU restatement, fp32, all host cores) on a bounded sample of the same workload:
    full training steps (fwd + bwd + AdaBelief) on `--cpu-plots` synthetic 16k-point plots."""
    from oracle import sparse_ref as R
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.optim import AdaBelief
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads (os.cpu_count()={os.cpu_count()})")
    batch = synthetic.make_sparse_batch(list(range(900_000, 900_000 + args.cpu_plots)), n_points=args.points)
    sd = {k: (v.detach().clone().float().requires_grad_(v.is_floating_point() and "running" not in k))
          for k, v in model_sd.items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = AdaBelief(params, lr=0.005, weight_decay=1e-2)
    coords = torch.cat([batch.batch[:, None], batch.coords.long()], 1).numpy()
    center, scale, w = stats
    t0 = time.time()
    steps = 0
    while steps < 2 or (time.time() - t0 < 12 and steps < 40):   # a bounded sample: about 10-15 s of CPU work
        out = R.resnet_forward(sd, coords, batch.x, layers, batch_size=args.cpu_plots)
        loss = R.reg_loss(out, batch.y_reg, center, scale, w)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(params, 100)
        opt.step()
        steps += 1
        if steps <= 3 or steps % 5 == 0:
            log(f"cpu baseline: step {steps} done after {time.time() - t0:.1f}s")
    dt = time.time() - t0
    return dict(value=round(args.cpu_plots * steps / dt, 4), unit="plots/s", cores=cores, kind="port",
                sample=f"{steps} training step(s) of M{args.model} on {args.cpu_plots} synthetic {args.points}-pt plots "
                       f"(oracle/sparse_ref.py, torch-CPU fp32, {cores} threads), {dt:.1f} s")


import numpy as np  # noqa: E402


def cpu_baseline_pointnet(model_sd, stats, points, nb=4):
    """cpu_baseline leg of BASELINE config 2 (tools/bench_config.py pointnet): the oracle's MinkowskiPointNet
    (torch-CPU fp32, all usable cores), full training steps on a bounded sample of `nb` plots."""
    from oracle import sparse_ref as R
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.optim import AdaBelief
    cores = usable_cores()
    torch.set_num_threads(cores)
    cb = synthetic.make_sparse_batch(list(range(900_000, 900_000 + nb)), n_points=points)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in model_sd.items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = AdaBelief(params, lr=0.005, weight_decay=1e-2)
    feats = torch.cat([cb.pos, cb.x], 1)
    center, scale, w = stats
    t0, n = time.time(), 0
    while n < 2 or (time.time() - t0 < 12 and n < 40):
        out = R.pointnet_forward(sd, cb.batch, feats, nb)
        loss = R.reg_loss(out, cb.y_reg, center, scale, w)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(params, 100)
        opt.step()
        n += 1
    d = time.time() - t0
    return dict(value=round(nb * n / d, 3), unit="plots/s", cores=cores, kind="port",
                sample=f"{n} training steps of MPointNet on {nb} synthetic {points}-pt plots "
                       f"(oracle/sparse_ref.py, torch-CPU fp32, {cores} threads), {d:.1f} s")


def cpu_baseline_kpconv_index(batch, B, points):
    """The index path of one batch on the C++ restatement: single thread, then one plot per task on all cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import kpconv_index as O
    from dpcr_agb_amd.config import kpconv_config
    cfg = kpconv_config()
    pos = batch.pos.cpu().numpy().astype(np.float32)
    nplots = min(B, 8)                       # a bounded sample: 8 plots (the reference does the whole batch per step)
    per = [pos[i * points:(i + 1) * points] for i in range(nplots)]

    def pyramid(pts_list):
        pts = np.concatenate(pts_list)
        lens = np.array([len(p) for p in pts_list], dtype=np.int32)
        r = cfg.first_subsampling_dl * cfg.conv_radius
        for level in range(5):
            O.batch_neighbors(pts, pts, lens, lens, r)
            if level == 4:
                break
            sub, sub_l = O.batch_grid_subsampling(pts, lens, sampleDl=2 * r / cfg.conv_radius)[:2]
            O.batch_neighbors(sub, pts, sub_l, lens, r)
            pts, lens, r = sub, sub_l, r * 2

    t0 = time.time()
    pyramid(per)
    single = time.time() - t0
    cores = usable_cores()
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda p: pyramid([p]), per))
    multi = time.time() - t0
    return dict(value=round(nplots / single, 3), unit="plots/s (index path only)", cores=1, kind="port",
                all_cores=dict(value=round(nplots / multi, 3), cores=cores),
                sample=f"the KPConv index path (5 x radius neighbours + 4 x grid subsampling + 4 x pooling neighbours) of "
                       f"{nplots} synthetic {points}-pt plots on oracle/kpconv_index_ref.cpp: {single:.1f} s single thread "
                       f"(how the reference runs it), {multi:.1f} s with one plot per task on {cores} threads; the GPU "
                       f"number next to it is the WHOLE training step")


OTHER_CONFIGS = (
    ("config2", "PointNet fp32 on 1xMI355X, synthetic 16k-pt plots, batch=64",
     [os.path.join("tools", "bench_config.py"), "pointnet", "--steps", "20", "--warmup", "5"]),
    ("config3", "KPConv rigid, 1xMI355X, 16k-pt plots batch=32",
     [os.path.join("tools", "bench_config.py"), "kpconv", "--points", "16000", "--steps", "20", "--warmup", "5"]),
    ("config5_single_gpu_leg", "MSENet50 biomass+wood-volume, bf16 with fp32 index kernels (one rank's share: batch 32)",
     ["bench.py", "--model", "SENet50", "--precision", "bf16", "--bf16-rows", "--steps", "30", "--warmup", "8",
      "--no-other-configs"]),
    ("end_to_end", "config 4 from raw points: sparse-xy.yaml train transform chain on the device + MSENet14 step",
     [os.path.join("tools", "bench_config.py"), "end2end", "--steps", "150", "--warmup", "15"]),
    ("end_to_end_config3", "config 3 from raw points: xy.yaml train transform chain on the device (MaxPoints 6144) + input pyramid + "
     "KPCNN step", [os.path.join("tools", "bench_config.py"), "kpconv_e2e", "--steps", "40", "--warmup", "8"]),
)
# BASELINE.json's metric is "training plots/sec ...; val RMSE": a short fixed-seed training run (reproducible: fixed-order
# weight-gradient sums, seeded drop-path draws) on synthetic labelled plots, evaluated as eval.py does
# (trial 0 of the R2 acceptance schedule, tests/golden/make_r2_cpu_leg.py: 256 / 128 plots of 700-1600 points, batch 16, 150 epochs,
# calibrate_bn, running-statistics evaluation; ~11 s; the same numbers as row 0 of the fp32 leg of tests/golden/r2_hip_expected.json)
TRAIN_EVAL = [os.path.join("tools", "train_eval.py"), "--acceptance-trial", "0"]


def run_other_configs(timeout_s=150):
    """BASELINE configs 2, 3 and the single-GPU leg of config 5 under the same clock as the headline: each as a FRESH CHILD
    PROCESS (subprocess: a new interpreter with its own HIP context; never an exec of this GPU-initialised process), bounded
    in time; a failure is recorded as a string.  The headline metric / value / config are not touched by these."""
    import subprocess
    out = {}
    for key, name, argv in OTHER_CONFIGS:
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable] + argv, cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[key] = dict(baseline_config=name, error=f"rc {r.returncode}: " + (r.stderr or r.stdout)[-400:])
                continue
            d = json.loads(lines[-1])
            out[key] = dict(baseline_config=name, metric=d["metric"], value=d["value"], unit=d["unit"],
                            ms_per_step=d["ms_per_step"], steps=d["steps"], warmup=d["warmup"], dtype=d["dtype"],
                            workload=d["config"]["workload"], roofline=d.get("roofline"),
                            cpu_baseline=d.get("cpu_baseline"), wall_s=round(time.perf_counter() - t0, 1),
                            command="python " + " ".join(argv))
            for extra in ("ball_query_roofline", "index_path_ms_per_step", "step_ms_p50", "input_chain_device_ms_per_step",
                          "host_draws_ms_per_step_p50", "entry_points_ms_per_step", "host_enqueue_floor_ms", "step_ms_p10",
                          "step_ms_p90", "step_mfma_frac", "achieved_hbm_gbs", "kernels_time_share"):
                if extra in d:
                    out[key][extra] = d[extra]
        except subprocess.TimeoutExpired:
            out[key] = dict(baseline_config=name, error=f"timeout after {timeout_s} s")
        except Exception as e:      # noqa: BLE001 — a failed side measurement must not lose the headline line
            out[key] = dict(baseline_config=name, error=f"{type(e).__name__}: {e}")
        log(f"other config {key}: " + (f"{out[key]['value']} {out[key]['unit']}" if "value" in out[key] else out[key]["error"]))
    return out


def run_train_eval(timeout_s=150):
    """val RMSE / R2 (instance_tracker.py:85-87 definitions, metrics.py) of a short fixed-seed training of MSENet14 on synthetic
    labelled plots, as a fresh child process: {val_rmse: [biomass, volume], val_r2: [...], ...} or {error: ...}."""
    import subprocess
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable] + TRAIN_EVAL, cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"final"' in ln]
        if r.returncode != 0 or not lines:
            return dict(error=f"rc {r.returncode}: " + (r.stderr or r.stdout)[-400:])
        d = json.loads(lines[-1])
        return dict(val_rmse=d["final"]["val_rmse"], val_r2=d["final"]["val_r2"], train_loss=d["final"]["train_loss"],
                    epochs=d["final"]["epoch"] + 1, schedule=d["config"],
                    reproducible=True, wall_s=round(time.perf_counter() - t0, 1), command="python " + " ".join(TRAIN_EVAL),
                    note="synthetic labels (biomass = a * sum(height^b) of the plot's trees): the reference's NFI data is not "
                         "available offline; metric definitions = the reference's (tests/test_metrics.py golden vectors)")
    except subprocess.TimeoutExpired:
        return dict(error=f"timeout after {timeout_s} s")
    except Exception as e:      # noqa: BLE001
        return dict(error=f"{type(e).__name__}: {e}")


def launch_ranks(args):
    """`python bench.py --gpus N` as a plain command line (no launcher around it): start the N ranks as a FRESH CHILD PROCESS
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` — before this process has made any GPU call; never an
    exec of this process — relay its output (rank 0 prints the one JSON line) and exit with its return code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    have = torch.cuda.device_count()        # (counting devices does not initialise the HIP runtime)
    env = dict(os.environ)
    if have < args.gpus and "AGB_BENCH_BACKEND" not in env:
        raise SystemExit(f"--gpus {args.gpus} but this node shows {have} device(s) (AGB_BENCH_BACKEND=gloo runs the ranks on the "
                         "devices there are, as a dry run of the multi-rank code path — never a measurement)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without a launcher: starting the ranks with: {' '.join(cmd)}")
    raise SystemExit(subprocess.run(cmd, env=env, cwd=os.getcwd()).returncode)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # AGB_BENCH_BACKEND=gloo: dry run of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # gradients travel through gloo) — a correctness aid, never a measurement
    backend = os.environ.get("AGB_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # AGB_FORCE_GRAD_SYNC on one rank: the gradient exchange runs anyway, through a process group of ONE rank — RCCL's
    # init, ReduceOp.AVG and its stream hand-off against the HIP kernels execute on a single-GPU box (tests/test_dist_gpu.py;
    # arithmetically the identity, never a scaling number)
    # (AGB_FORCE_GRAD_SYNC=buckets: the bucket path alone, no process group, no collective — the comparison run)
    force_sync = bool(os.environ.get("AGB_FORCE_GRAD_SYNC"))
    force_coll = force_sync and os.environ.get("AGB_FORCE_GRAD_SYNC") != "buckets"
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import dpcr_agb_amd
    from dpcr_agb_amd import sparse_ops, synthetic
    dpcr_agb_amd.limit_host_threads()     # (the cpu_baseline leg sets its own thread count afterwards)
    global PRECISION, PMC_CONFIG
    PRECISION = args.precision
    if args.model == "SENet50" and args.precision == "bf16" and args.bf16_rows:
        PMC_CONFIG = "config5"
    elif args.model != "SENet14" or args.precision != "fp32":
        PMC_CONFIG = "none"
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.dist import GradAllReduce, broadcast_parameters, shard_seeds
    from dpcr_agb_amd.instance import MinkowskiBaselineModel

    torch.manual_seed(0)
    import random
    random.seed(20240 + rank)     # the drop-path draws (backbones/sparse.py: one python random number per plot and block)
    ds = synthetic.SyntheticDataset(feature_dimension=args.features, stat_seeds=range(10_000, 10_064))
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS[args.model]), "minkowski", ds)
    model_sd_cpu = {k: v.detach().clone() for k, v in model.model.state_dict().items()}
    model.to(dev).train()
    if args.bf16_rows and args.precision != "bf16":
        raise SystemExit("--bf16-rows needs --precision bf16")
    model.set_kernel_options(precision=args.precision, bf16_activations=args.bf16_rows,      # carried by the model
                             deterministic_wgrad=args.deterministic)
    broadcast_parameters(model)
    model.init_train_objects(TRAINING_NFI)
    sync = None
    if world > 1 or force_sync:   # the env switch exercises the bucket path AND the collective on one GPU
        sync = GradAllReduce(model.parameters(), force_collective=force_coll)
        model.grad_sync = sync

    # pre-generated, pre-voxelised, device-resident batches (disjoint seeds per rank)
    pool = []
    gb = args.batch * world
    for i in range(args.pool):
        seeds = shard_seeds(gb, rank, world, i)
        pool.append(synthetic.make_sparse_batch(seeds, n_points=args.points,
                                                extra_feature=args.features == 4).to(dev))
    voxels = sum(int(b.coords.shape[0]) for b in pool) / len(pool) / args.batch
    steps_per_epoch = 133  # 4271 train plots / 32

    host_ms, host_cpu_ms, main_cpu_ms = [], [], []

    def step(i):
        t_h, c_h, m_h = time.perf_counter(), time.process_time(), time.thread_time()
        _step(i)
        host_ms.append((time.perf_counter() - t_h) * 1e3)
        host_cpu_ms.append((time.process_time() - c_h) * 1e3)     # CPU time of ALL threads of this rank
        main_cpu_ms.append((time.thread_time() - m_h) * 1e3)      # ... of the enqueuing thread alone

    def _step(i):
        # software pipeline of the input path: this step's coordinate pyramid was built on a side stream while the
        # previous step ran; the next one is built now, behind this step's already-enqueued kernels
        model.set_input(pool[i % len(pool)], dev)
        model.optimize_parameters(epoch=i // steps_per_epoch, batch_size=args.batch, num_batches=steps_per_epoch)
        if not args.no_prefetch:
            # two batches deep: this call builds the kernel maps of batch i+1 (its coordinate levels were staged one
            # step ago) and stages the levels of batch i+2 — the host never waits for a device read-back
            model.prefetch_input(pool[(i + 2) % len(pool)], dev)

    # The side-stream input pipeline is worth 6 % of throughput on one GPU, but a second hardware queue with cross-stream
    # events keeps one thread of the HIP runtime busy for ~9 ms of CPU per step (10.2 -> 18.3 ms of CPU per step and rank,
    # profiles/r03_host_cpu.txt).  Where the ranks of a node have fewer than ~2.5 cores each (cgroup quota / world size) that
    # thread would get the whole process throttled: build the input on the compute stream there.
    # (the node's uncapped core count over the ranks of THIS node: LOCAL_WORLD_SIZE under torch.distributed.run)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world) or world)
    if not args.no_prefetch and not args.force_prefetch and usable_cores(None) / max(local_world, 1) < 2.5:
        args.no_prefetch = True
        log(f"{usable_cores(None)} usable cores for {local_world} rank(s) on this node: input pipeline on the compute stream "
            f"(--no-prefetch)")
    # allocator pools grown up front (per stream): no device allocation inside the timed region
    model.reserve_workspace(dev, main_bytes=args.reserve_gib << 30, side_bytes=(args.reserve_gib << 30) // 2)
    if not args.no_prefetch and len(pool) >= 3:
        model.prefetch_input(pool[0], dev)
        model.prefetch_input(pool[1], dev)
    # manual garbage collection, as large training loops do: a generation-2 sweep over the module / autograd object
    # graph costs ~80 ms of host time and stalls the device queue when it happens to fall inside a step
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    log(f"model + {len(pool)} batches resident ({voxels:.0f} voxels/plot); warmup")
    # HIP events around a launch cost ~6 us of queue bubbles each side.  Every sparse-conv launch is bracketed only in
    # up to three (untimed) warmup steps: that gives the per-kernel table and tells which kernel dominates.  Inside the
    # timed region only that kernel is bracketed, in every EV_EVERY-th step (a bracketed step measures ~0.6 ms longer).
    EV_EVERY = 9   # coprime with the pool of 4 batches: the bracketed steps cycle through all of them
    n_instr = min(3, args.warmup)
    prof_all = []
    for i in range(args.warmup):
        if rank == 0 and i >= args.warmup - n_instr:
            sparse_ops.PROFILE, sparse_ops.PROFILE_FILTER = prof_all, None
        step(i)
        sparse_ops.PROFILE = None
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    torch.cuda.synchronize()
    # the dominant kernel from the event times alone (no device work, no host pause): a ~100 ms idle gap here — the
    # pair-count reductions and the per-layer table — let the GPU clocks drop, and the first ~25 timed steps then ran
    # 5-8 % slower than the rest.  The table is printed after the timed region.
    ms_by_kernel = {}
    for rec in (prof_all if rank == 0 else []):
        ms_by_kernel[kernel_of(rec)] = ms_by_kernel.get(kernel_of(rec), 0.0) + rec["start"].elapsed_time(rec["end"])
    dom = max(ms_by_kernel, key=ms_by_kernel.get) if ms_by_kernel else None
    # the launches to bracket inside the timed region are chosen BEFORE they are issued (the kernel's name is only known
    # afterwards): by the call signatures that took the dominant kernel in the instrumented warmup steps
    sig = lambda r: (r["kind"], r["K3"], r["cin"], r["cout"], bool(r.get("perm")), r.get("split", 1))      # noqa: E731
    dom_sigs = {sig(r) for r in prof_all if kernel_of(r) == dom}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prof = []
    threads0 = thread_cpu_times()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    ev_start = torch.cuda.Event(enable_timing=True)
    ev_start.record()
    step_events = []
    for i in range(args.steps):
        if rank == 0 and i % EV_EVERY == 0:   # (no warmup step to pick the kernel from: bracket them all)
            sparse_ops.PROFILE = prof
            sparse_ops.PROFILE_FILTER = (lambda rec: sig(rec) in dom_sigs) if dom is not None else None
        step(args.warmup + i)
        sparse_ops.PROFILE = None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        step_events.append(ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sparse_ops.PROFILE_FILTER = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(model.loss.detach())
    # host time of a step on an EMPTY device queue (drained before every step; outside the timed region): what the enqueuing
    # thread costs by itself
    floor = []
    for i in range(8):
        torch.cuda.synchronize()
        t_f = time.perf_counter()
        _step(args.warmup + args.steps + i)
        floor.append((time.perf_counter() - t_f) * 1e3)
    torch.cuda.synchronize()
    floor.sort()
    # the shader clock while steps execute (six more steps enqueued, sampled before they finish; outside the timed region):
    # the state a 2-4 % difference between two boxes, or between two runs on one, usually comes from
    clock_mhz = None
    try:
        for i in range(6):
            _step(args.warmup + args.steps + 8 + i)
        clock_mhz = int(torch.cuda.clock_rate())
    except Exception:      # noqa: BLE001 — no SMI library in the image: the line goes out without the clock
        clock_mhz = None
    torch.cuda.synchronize()
    # what the exchange looked like, for the driver's scaling run: backend, ranks, buckets and bytes per step; a checksum
    # of every rank's parameters after the timed steps (identical on all ranks when the gradients were averaged) and each
    # rank's host CPU time per step (all threads of the process: what N ranks ask of the node's cores)
    checksum = torch.zeros(1, dtype=torch.float64, device=dev)
    for p in model.parameters():
        checksum += p.detach().double().sum()
    hc = sorted(host_cpu_ms[-args.steps:])
    slots = torch.zeros(world, 3, dtype=torch.float64, device=dev)
    slots[rank] = torch.tensor([float(checksum.item()), hc[len(hc) // 2], sorted(host_ms[-args.steps:])[len(hc) // 2]],
                               dtype=torch.float64)
    if world > 1:
        dist.all_reduce(slots, op=dist.ReduceOp.SUM)     # (a gather written as a sum: every backend has all_reduce)
    gathered = list(slots)
    comm = dict(backend=(dist.get_backend() if dist.is_initialized() else None), world=world,
                buckets=len(sync.buckets) if sync is not None else 0,
                bytes_per_step=int(sum(b["flat"].numel() * b["flat"].element_size() for b in sync.buckets)) if sync else 0,
                param_checksums=[float(g[0].item()) for g in gathered],
                host_cpu_ms_per_step_p50=[round(float(g[1].item()), 3) for g in gathered],
                host_enqueue_ms_p50=[round(float(g[2].item()), 3) for g in gathered])
    gaps = []
    if len(step_events) > 1:
        gaps = [step_events[j].elapsed_time(step_events[j + 1]) for j in range(len(step_events) - 1)]
        log("device time between step ends (ms): " + " ".join(f"{g:.2f}" for g in gaps))
        # the edges of the timed region: it starts on a drained queue (the first step runs at the pace of its enqueue) and
        # ends when every stream has drained (the input stream's work for the steps after the last one included)
        log(f"first timed step, from the start of the region to its end event: {ev_start.elapsed_time(step_events[0]):.2f} ms; "
            f"region {elapsed * 1e3:.2f} ms = {ev_start.elapsed_time(step_events[-1]):.2f} ms to the last step's end event + "
            f"{elapsed * 1e3 - ev_start.elapsed_time(step_events[-1]):.2f} ms")
        log("host enqueue per step (ms): " + " ".join(f"{h:.2f}" for h in host_ms[-args.steps:]))
    ms1 = torch.cuda.memory_stats()
    log("caching allocator over the timed region: "
        f"{ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)} device allocations, "
        f"{ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0)} device frees, "
        f"{ms1.get('num_alloc_retries', 0) - ms0.get('num_alloc_retries', 0)} retries")
    log(f"device memory: {torch.cuda.memory_allocated() / 2**30:.2f} GiB allocated now, "
        f"{torch.cuda.max_memory_allocated() / 2**30:.2f} GiB peak, {torch.cuda.memory_reserved() / 2**30:.2f} GiB reserved")
    threads1 = thread_cpu_times()
    busy = sorted(((threads1[t][1] - threads0.get(t, (None, 0.0))[1], threads1[t][0], t) for t in threads1), reverse=True)
    log("CPU time per step by thread over the timed region (ms): " +
        ", ".join(f"{name}[{tid}] {dt * 1e3 / args.steps:.2f}" for dt, name, tid in busy[:6] if dt > 0))
    mc = sorted(main_cpu_ms[-args.steps:])
    log(f"host CPU time per step: all threads median {hc[len(hc) // 2]:.2f} ms, enqueuing thread {mc[len(mc) // 2]:.2f} ms")
    hm = sorted(host_ms[-args.steps:])
    log(f"timed region: {elapsed:.3f}s for {args.steps} steps; host enqueue time per step: median "
        f"{hm[len(hm) // 2]:.2f} ms, min {hm[0]:.2f} ms (GPU-bound when well below ms_per_step)")

    if rank == 0:
        groups_all = group_profile(prof_all, table=True)
        groups_timed = group_profile(prof, table=not groups_all)
        if dom is None:
            dom, groups_all = dominant_kernel(groups_timed), groups_timed
        roof = None
        if dom in groups_timed:
            roof = roofline_of(dom, groups_timed[dom])
            roof["timed_steps_bracketed"] = len(range(0, args.steps, EV_EVERY))
        summary = kernel_summary(groups_all)
        step_bytes, step_bytes_src = pmc_step_bytes()
        line = {
            "metric": f"training plots/sec (16k-pt NFI plots) M{args.model}",
            "value": round(world * args.batch * args.steps / elapsed, 2),
            "unit": "plots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3"}[args.precision], "data": "synthetic",
            "config": {"workload": f"{args.model} sparse-voxel training step, {args.points}-pt synthetic plots, "
                                   f"voxel 0.0125 (0.375x0.375x0.5 m), batch {args.batch}/GPU, F={args.features}, "
                                   f"~{voxels:.0f} voxels/plot, fwd+bwd+AdaBelief incl. coordinate hash/kernel maps",
                       "global_batch": gb, "parallelism": f"dp{world}", "final_loss": round(loss, 5),
                       "input_pipeline": "compute stream" if args.no_prefetch else "side stream, two batches ahead",
                       "operands": {"fp32": "fp32 MFMA (exact)", "bf16": "bf16 operands, fp32 accumulate; stem, BN, SE, "
                                    "index kernels fp32", "bf16x3": "split-bf16 (3 MFMAs per product), fp32 accumulate"}[
                                        args.precision] + ("; activation / gradient rows stored in bf16" if args.bf16_rows
                                                           else "")},
            "roofline": roof,
            # the whole step against the MFMA roof: convolution FLOPs of one step (the fully bracketed warmup steps) over the
            # step time; and against HBM: PMC bytes of one step (committed pass) over the step time
            "step_mfma_frac": round(sum(g["flops"] for g in groups_all.values()) / max(n_instr, 1) / (elapsed / args.steps)
                                    / 1e12 / mfma_peak_tf(), 4) if groups_all else None,
            "achieved_hbm_gbs": (round(step_bytes / (elapsed / args.steps) / 1e9, 1) if step_bytes else None),
            "achieved_hbm_source": step_bytes_src,
            "step_ms_p10": round(sorted(gaps)[int(len(gaps) * 0.1)], 3) if gaps else None,
            "step_ms_p50": round(sorted(gaps)[len(gaps) // 2], 3) if gaps else None,
            "step_ms_p90": round(sorted(gaps)[int(len(gaps) * 0.9)], 3) if gaps else None,
            "step_ms_min": round(min(gaps), 3) if gaps else None,
            "gpu_clock_mhz_under_load": clock_mhz,
            "device_allocs_in_timed_region": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
            "host_enqueue_ms_p50": round(hm[len(hm) // 2], 3), "host_enqueue_floor_ms": round(floor[len(floor) // 2], 3),
            "host_cpu_ms_per_step_p50": round(hc[len(hc) // 2], 3),
            "host_main_thread_cpu_ms_p50": round(mc[len(mc) // 2], 3),
            "comm": comm,
            "kernels": summary, "kernels_from": f"{n_instr} fully bracketed warmup step(s), outside the timed region",
        }
        if world == 1 and not args.no_cpu_baseline:
            stats = (model.reg_center_targets.cpu(), model.reg_scale_targets.cpu(), model.reg_weights.cpu())
            line["cpu_baseline"] = cpu_baseline(args, model_sd_cpu, stats, tuple(model.model.LAYERS))
        if world == 1 and not args.no_other_configs and args.model == "SENet14" and args.precision == "fp32":
            # (the headline run only; this process's cached pools go back to the device first: 288 GB, but the children
            # reserve 12-24 GiB each)
            torch.cuda.empty_cache()
            line["other_configs"] = run_other_configs()
            te = run_train_eval()
            line["train_eval"] = te
            if "val_rmse" in te:
                line["val_rmse"], line["val_r2"] = te["val_rmse"], te["val_r2"]
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
