"""Bench lines for BASELINE.json configs 2 and 3 (bench.py stays on the headline, config 4):

  python tools/bench_config.py pointnet [--steps 20] [--warmup 5]     config 2: MPointNet fp32, B = 64 x 16k-pt plots
  python tools/bench_config.py kpconv [--points 16000|6144]           config 3: KPConv rigid, B = 32 plots

One JSON line per run in bench.py's format: training plots/s (whole step: input pyramid / coordinate maps on the device +
forward + backward + AdaBelief), `roofline` of the library entry point that takes the most device time (HIP events around
every library call during three instrumented steps OUTSIDE the timed region; algorithmic bytes / FLOPs per SURVEY.md
§8d) and `cpu_baseline`:
  * pointnet: oracle/sparse_ref.py:pointnet_forward (torch-CPU fp32, all usable cores), full training steps on a sample;
  * kpconv: the KPConv index path (5 x radius neighbours, 4 x grid subsampling, 4 x pooling neighbours) of the same
    batch on the C++ restatement oracle/kpconv_index_ref.cpp — single thread (how the reference runs it, GIL held) and all
    usable cores (one plot per task) — the part of a KPConv step the reference spends on the CPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS, MFMA_F32_PEAK_TF = 8000.0, 157.3


def log(msg):
    print(f"[bench_config] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


# entry points whose launch notes its compute kernel (agb_last_kernel): the roofline entry names that kernel
KERNEL_NOTED = ("agb_spconv_fwd_opt", "agb_spconv_fwd_lp", "agb_spconv_fwd_tiles", "agb_spconv_fwd_b16", "agb_spconv_fwd_h",
                "agb_spconv_bwd_weight_lp", "agb_spconv_bwd_weight_ws", "agb_spconv_bwd_weight_b16_ws", "agb_dense_fwd_bn")
KERNEL_OF_ENTRY = {"agb_ball_query_fill": "k_ball_query4<true>", "agb_ball_query_fill_csr": "k_ball_query4<true> (ragged rows)",
                   "agb_ball_query_fill_csr_m": "k_ball_query4<true> (ragged rows)"}


class CallTimer:
    """Brackets every library call (dpcr_agb_amd._lib.call) with HIP events on torch's current stream while active."""

    def __init__(self):
        from dpcr_agb_amd import _lib
        self._lib, self._orig, self.records = _lib, _lib.call, []

    def __enter__(self):
        def timed(name, *args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = self._orig(name, *args)
            e1.record()
            kern = self._lib.last_kernel() if name in KERNEL_NOTED else None
            note = getattr(self._lib, "CALL_NOTE", None)     # (what a wrapper knows and the argument list does not: pair counts)
            self._lib.CALL_NOTE = None
            self.records.append((name, args, e0, e1, kern, note))
            return rc
        self._lib.call = timed
        import dpcr_agb_amd.sparse_ops as so, dpcr_agb_amd.norm_ops as no, dpcr_agb_amd.kpconv_ops as ko  # noqa: E401
        import dpcr_agb_amd.kp_index as ki
        self._mods = [m for m in (so, no, ko, ki) if hasattr(m, "_lib")]
        return self

    def __exit__(self, *exc):
        self._lib.call = self._orig

    def by_name(self):
        torch.cuda.synchronize()
        out = {}
        for name, args, e0, e1, kern, note in self.records:
            g = out.setdefault(name, dict(ms=0.0, n=0, calls=[], kernels={}, call_kernels=[], call_notes=[]))
            ms = e0.elapsed_time(e1)
            g["ms"] += ms
            g["n"] += 1
            g["calls"].append((args, ms))
            g["call_kernels"].append(kern)
            g["call_notes"].append(note)
            if kern:
                g["kernels"][kern] = g["kernels"].get(kern, 0.0) + ms
        return out


def conv_call_cost(args):
    """agb_spconv_fwd_opt / _lp(X, ldx, W, nbr, nbr_stride, kflip, bias, Y, ldy, n_out, K3, Cin, Cout, ...) with the
    identity map: a dense [n, Cin] x [Cin, Cout] product."""
    n, K3, cin, cout = args[9], args[10], args[11], args[12]
    return 2.0 * n * cin * cout, (n * (cin + cout) + K3 * cin * cout) * 4.0


def wgrad_call_cost(args):
    """agb_spconv_bwd_weight_lp(X, ldx, dY, ldy, nbr, nbr_stride, dW, n_out, K3, Cin, Cout, precision)"""
    n, K3, cin, cout = args[7], args[8], args[9], args[10]
    return 2.0 * n * cin * cout, (n * (cin + cout) + K3 * cin * cout) * 4.0


def dense_bn_call_cost(args):
    """agb_dense_fwd_bn(X, ldx, W, bias, Y, ldy, n, Cin, Cout, bn_part, stream)"""
    n, cin, cout = args[6], args[7], args[8]
    return 2.0 * n * cin * cout, (n * (cin + cout) + cin * cout) * 4.0


def kp_gather_fwd_cost(args, note=None):
    """agb_kpconv_gather_fwd_csr(q, s, row_ptr, indices, limit, Ns, x, ldx, kp, K, extent, wf, N, Cin, stream): SURVEY.md 8(d)
    KPConv gather: sum(valid) * (12 + Cin * 4) read + the gathered wf [N, K, Cin] written (the fused layer would write N * Cout
    instead) ; FLOPs = sum(valid) * K * (10 + 2 Cin).  sum(valid) = neighbour entries of the rows (the wrapper's note)."""
    K, N, cin = args[9], args[12], args[13]
    valid = (note or {}).get("valid", 0)
    return valid * K * (10.0 + 2.0 * cin), valid * (12.0 + cin * 4.0) + N * K * cin * 4.0 + N * 16.0


def kp_gather_bwd_cost(args, note=None):
    """agb_kpconv_gather_bwd_csr(q, s, row_ptr, indices, limit, Ns, dwf, kp, K, extent, dx, ldx, N, Cin, stream)"""
    K, N, cin = args[8], args[12], args[13]
    valid = (note or {}).get("valid", 0)
    return valid * K * (10.0 + 2.0 * cin), valid * (12.0 + cin * 4.0) + N * K * cin * 4.0 + N * 16.0


def kp_fused_fwd_cost(args, note=None):
    """agb_kpconv_fused_fwd(pts, row_ptr, indices, limit, N, x, ldx, kp, K, extent, W, out, ldo, Cin, Cout, stream): SURVEY.md
    8(d) fused KPConv: sum(valid) * (12 + Cin * 4) + N * Cout * 4 + K * Cin * Cout * 4 bytes (+ the index rows: 4 per entry, 4
    per row); FLOPs = gather sum(valid) * K * (10 + 2 Cin) + contraction 2 N K Cin Cout."""
    N, K, cin, cout = args[4], args[8], args[13], args[14]
    valid = (note or {}).get("valid", 0)
    return (valid * K * (10.0 + 2.0 * cin) + 2.0 * N * K * cin * cout,
            valid * (12.0 + cin * 4.0 + 4.0) + N * (cout * 4.0 + 16.0) + K * cin * cout * 4.0)


def kp_fused_bwd_cost(args, note=None):
    """agb_kpconv_fused_bwd(pts, row_ptr, indices, limit, N, dy, lddy, kp, K, extent, W, x, ldx, dx, lddx, dW, accumulate,
    workspace, workspace_bytes, Cin, Cout, stream): the gather on dy, then dx = wfd W^T and dW = x^T wfd from the same tile."""
    N, K, cin, cout = args[4], args[8], args[19], args[20]
    note = note or {}
    valid, both = note.get("valid", 0), int(bool(note.get("dx", True))) + int(bool(note.get("dw", True)))
    return (valid * K * (10.0 + 2.0 * cout) + both * 2.0 * N * K * cin * cout,
            valid * (12.0 + cout * 4.0 + 4.0) + N * (both * cin * 4.0 + 16.0) + 2.0 * K * cin * cout * 4.0)


def pn_pool_fwd_cost(args):
    """agb_pointnet_pool_fwd_aux(Z, ldz, n, C, ...): the [n, C] pre-activation read once (SURVEY 8(d) PointNet)"""
    return 0.0, args[2] * args[3] * 4.0


def pn_pool_bwd_cost(args):
    """agb_pointnet_pool_bwd_aux(Z, ldz, n, C, ...): z read, dz written"""
    return 0.0, 2.0 * args[2] * args[3] * 4.0


def rows_cost(n_idx, c_idx, passes):
    return lambda args: (0.0, passes * args[n_idx] * args[c_idx] * 4.0)


# algorithmic (FLOPs, bytes) of one call of every entry point that can dominate a config's step
ALL_COSTS = {
    "agb_spconv_fwd_opt": conv_call_cost, "agb_spconv_fwd_lp": conv_call_cost, "agb_spconv_fwd_tiles": conv_call_cost,
    "agb_spconv_bwd_weight_lp": wgrad_call_cost, "agb_spconv_bwd_weight_ws": wgrad_call_cost,
    "agb_dense_fwd_bn": dense_bn_call_cost,
    "agb_kpconv_gather_fwd_csr": kp_gather_fwd_cost, "agb_kpconv_gather_bwd_csr": kp_gather_bwd_cost,
    "agb_kpconv_fused_fwd": kp_fused_fwd_cost, "agb_kpconv_fused_bwd": kp_fused_bwd_cost,
    "agb_pointnet_pool_fwd_aux": pn_pool_fwd_cost, "agb_pointnet_pool_bwd_aux": pn_pool_bwd_cost,
    "agb_bn_stats_tracked": rows_cost(2, 3, 1), "agb_bn_act_fwd": rows_cost(2, 3, 2), "agb_bn_act_bwd_colsum": rows_cost(4, 5, 3),
}
KERNEL_OF_ENTRY.update({"agb_kpconv_gather_fwd_csr": "k_kpconv_gather_mm_fwd", "agb_kpconv_gather_bwd_csr": "k_kpconv_gather_mm_bwd",
                        "agb_kpconv_fused_fwd": "k_kpconv_fused", "agb_kpconv_fused_bwd": "k_kpconv_fused",
                        "agb_pointnet_pool_fwd_aux": "k_pn_pool_fwd", "agb_pointnet_pool_bwd_aux": "k_pn_bwd"})


NOTE_COSTS = (kp_gather_fwd_cost, kp_gather_bwd_cost, kp_fused_fwd_cost, kp_fused_bwd_cost)    # priced with the wrapper's note


def kernel_rooflines(groups, config=None, costs=None):
    """The step's launches regrouped by KERNEL (the name the library noted, else the entry point's kernel): name ->
    dict(ms, n, flops, bytes, entry).  The roofline entry of a config's line is the kernel with the largest share of the
    step's bracketed device time — always (round-5 review: a cost function must not decide which kernel is reported)."""
    costs = dict(ALL_COSTS, **(costs or {}))
    per = {}
    for name, g in groups.items():
        cost = costs.get(name)
        for (args, ms), kern, note in zip(g["calls"], g["call_kernels"], g["call_notes"]):
            k = kern or KERNEL_OF_ENTRY.get(name, name)
            e = per.setdefault(k, dict(ms=0.0, n=0, flops=0.0, bytes=0.0, entry=name, priced=cost is not None))
            e["ms"] += ms
            e["n"] += 1
            if cost is not None:
                try:
                    fl, by = cost(args, note) if cost in NOTE_COSTS else cost(args)
                except Exception:
                    fl, by = 0.0, 0.0
                e["flops"] += fl
                e["bytes"] += by
    return per


def dominant_roofline(groups, config=None, costs=None):
    per = kernel_rooflines(groups, config, costs)
    total = sum(e["ms"] for e in per.values()) or 1.0
    shares = {k: round(e["ms"] / total, 3) for k, e in sorted(per.items(), key=lambda kv: -kv[1]["ms"])[:8]}
    dom = max(per, key=lambda k: per[k]["ms"])
    e = per[dom]
    secs = e["ms"] / 1e3
    ridge = MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
    if not e["priced"] or e["bytes"] <= 0:
        r = dict(bound=None, achieved=None, peak=None, unit=None, frac=None,
                 note="no algorithmic cost function for this entry point: named, not priced")
    elif e["flops"] / e["bytes"] >= ridge:
        ach = e["flops"] / secs / 1e12
        r = dict(bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(ach / MFMA_F32_PEAK_TF, 4))
    else:
        ach = e["bytes"] / secs / 1e9
        r = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
    traffic, src = pmc_traffic(dom, config) if config else (None, None)
    r.update(traffic=traffic, traffic_source=src, kernel=dom, entry_point=e["entry"], kernels_time_share=shares,
             kernels_time_share_of="the step's bracketed device time (three instrumented steps)", launches=e["n"],
             avg_launch_us=round(e["ms"] / e["n"] * 1e3, 2), alg_bytes_per_launch=round(e["bytes"] / e["n"]),
             alg_flops_per_launch=round(e["flops"] / e["n"]))
    return r, per


def step_level(per, ms_per_step, config, instrumented_steps=3):
    """step_mfma_frac: FLOPs of the priced MFMA products of one step over the step time and the fp32 MFMA peak;
    achieved_hbm_gbs: PMC bytes of one step (the config's committed pass) over the step time."""
    flops = sum(e["flops"] for e in per.values()) / instrumented_steps
    out = dict(step_mfma_frac=round(flops / (ms_per_step / 1e3) / 1e12 / MFMA_F32_PEAK_TF, 4))
    for tag in ("r06", "r05", "r04"):
        rel = os.path.join("profiles", f"{tag}_pmc_traffic_{config}.json")
        try:
            v = json.load(open(os.path.join(ROOT, rel))).get("hbm_bytes_per_step")
        except Exception:
            continue
        if v:
            out.update(achieved_hbm_gbs=round(float(v) / (ms_per_step / 1e3) / 1e9, 1), achieved_hbm_source=rel)
            break
    else:
        out.update(achieved_hbm_gbs=None, achieved_hbm_source=None)
    return out


def shape_table(groups):
    """Per distinct (rows, Cin, Cout) of the dense products and weight gradients: time, TFLOP/s and GB/s of 3 steps."""
    for name, cost, sig in (("agb_spconv_fwd_opt", conv_call_cost, lambda x: (x[9], (x[10], x[11], x[12]))),
                            ("agb_spconv_bwd_weight_lp", wgrad_call_cost, lambda x: (x[7], (x[8], x[9], x[10]))),
                            ("agb_spconv_bwd_weight_ws", wgrad_call_cost, lambda x: (x[7], (x[8], x[9], x[10]))),
                            ("agb_bn_stats_tracked", lambda x: (0.0, x[2] * x[3] * 4.0), lambda x: (x[2], (x[3],))),
                            ("agb_bn_act_fwd", lambda x: (0.0, 2.0 * x[2] * x[3] * 4.0), lambda x: (x[2], (x[3],))),
                            ("agb_bn_act_bwd_colsum", lambda x: (0.0, 3.0 * x[4] * x[5] * 4.0), lambda x: (x[4], (x[5],))),
                            ("agb_kpconv_gather_fwd", None, lambda x: (x[11], (x[12],))),
                            ("agb_kpconv_gather_bwd", None, lambda x: (x[11], (x[12],)))):
        if name not in groups:
            continue
        acc = {}
        for args, ms in groups[name]["calls"]:
            rows, key = sig(args)
            key = (len(str(rows)),) + key              # (row counts differ a little from batch to batch: group by magnitude)
            e = acc.setdefault(key, [0.0, 0, 0.0, 0.0, 0])
            e[0] += ms; e[1] += 1; e[4] += rows
            if cost:
                fl, by = cost(args)
                e[2] += fl; e[3] += by
        log(f"--- {name}: mean rows, (K3, Cin, Cout | Cin) -> ms per step, calls per step, TFLOP/s, GB/s")
        for k, (ms, n, fl, by, rows) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            log(f"    {rows // n:8d} {str(k[1:]):18s}: {ms / 3:8.3f} ms {n / 3:5.1f} calls  {fl / ms / 1e9 if fl else 0:7.1f} TF"
                f"  {by / ms / 1e6 if by else 0:7.0f} GB/s")


def pmc_traffic(kernel, config):
    """(HBM bytes per launch of `kernel`, file) from the committed PMC pass of this config (tools/collect_pmc_configs.sh ->
    profiles/<tag>_pmc_traffic_<config>.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections), or (None, None)."""
    for tag in ("r06", "r05", "r04"):
        rel = os.path.join("profiles", f"{tag}_pmc_traffic_{config}.json")
        try:
            data = json.load(open(os.path.join(ROOT, rel)))["kernels"]
        except Exception:
            continue
        want = kernel.split(" (")[0]
        for k, v in data.items():
            if k.replace("void ", "").strip() == want:
                return round(v["hbm_bytes_per_launch"]), rel
        # a kernel FAMILY (the name an entry point is booked under when it launches one of several instantiations:
        # k_kpconv_fused<1, 1, false, 4, 1>, <2, 2, true, 8, 1>, ...): the launch-weighted mean of the family
        fam = [v for k, v in data.items() if k.replace("void ", "").strip().startswith(want + "<")]
        n = sum(v["launches"] for v in fam)
        if n > 0:
            return round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / n), rel
    return None, None


def pmc_traffic_prefix(prefixes, config):
    """(HBM bytes per STEP of all kernels whose names start with one of `prefixes`, file) from the config's committed PMC pass —
    for an entry point that launches several kernels (the voxeliser) — or (None, None)."""
    for tag in ("r06", "r05", "r04"):
        rel = os.path.join("profiles", f"{tag}_pmc_traffic_{config}.json")
        try:
            doc = json.load(open(os.path.join(ROOT, rel)))
        except Exception:
            continue
        steps = doc.get("steps") or 1
        tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in doc["kernels"].items()
                  if k.replace("void ", "").strip().startswith(tuple(prefixes)))
        if tot > 0:
            return round(tot / steps), rel
    return None, None


def step_shares(groups, top=8):
    """Share of the instrumented steps' bracketed device time per KERNEL (the name the library noted when it launched it, else
    the entry point): what decides which kernel dominates a config's step (round-4 review: the share inside one entry point
    says nothing about the step)."""
    per = {}
    for name, g in groups.items():
        noted = sum(g.get("kernels", {}).values())
        for k, ms in g.get("kernels", {}).items():
            per[k] = per.get(k, 0.0) + ms
        if g["ms"] - noted > 1e-6:
            key = KERNEL_OF_ENTRY.get(name, name)
            per[key] = per.get(key, 0.0) + g["ms"] - noted
    total = sum(per.values()) or 1.0
    return {k: round(v / total, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:top]}


def roofline_entry(name, g, cost, config=None, groups=None):
    flops = sum(cost(a)[0] for a, _ in g["calls"])
    byts = sum(cost(a)[1] for a, _ in g["calls"])
    secs = g["ms"] / 1e3
    ridge = MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
    if flops / max(byts, 1.0) >= ridge:
        ach = flops / secs / 1e12
        r = dict(bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(ach / MFMA_F32_PEAK_TF, 4))
    else:
        ach = byts / secs / 1e9
        r = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
    # the kernel(s) behind the entry point, by device time (the library notes what it launches: agb_last_kernel)
    kernels = {k: round(ms / g["ms"], 3) for k, ms in sorted(g.get("kernels", {}).items(), key=lambda kv: -kv[1])}
    kernel = next(iter(kernels), KERNEL_OF_ENTRY.get(name, name))
    traffic, src = pmc_traffic(kernel, config) if config else (None, None)
    r.update(traffic=traffic, traffic_source=src, kernel=kernel, entry_point=name,
             kernels_time_share=step_shares(groups) if groups is not None else (kernels or None),
             kernels_time_share_of="the step's bracketed device time (three instrumented steps)" if groups is not None
             else "this entry point", launches=g["n"],
             avg_launch_us=round(g["ms"] / g["n"] * 1e3, 2), alg_bytes_per_launch=round(byts / g["n"]),
             alg_flops_per_launch=round(flops / g["n"]))
    return r


def timed_loop(step, steps, warmup):
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    evs = []
    for i in range(steps):
        step(warmup + i)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    ms1 = torch.cuda.memory_stats()
    log("caching allocator over the timed region: "
        f"{ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)} device allocations, "
        f"{ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0)} device frees, "
        f"{ms1.get('num_alloc_retries', 0) - ms0.get('num_alloc_retries', 0)} retries; "
        f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB reserved, {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB peak")
    gaps = sorted(evs[j].elapsed_time(evs[j + 1]) for j in range(len(evs) - 1))
    return dt, gaps


def host_floor(step, first, n=10):
    """Host time of a step on an EMPTY device queue (the device is drained before every step: no launch ever waits for a queue
    slot): what the enqueuing thread costs by itself — median ms over n steps, outside any timed region."""
    ts = []
    for i in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(first + i)
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    ts.sort()
    return round(ts[len(ts) // 2], 3)


def top_table(groups, k=8):
    for name, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:k]:
        log(f"  {g['ms']:9.3f} ms  {g['n']:5d} calls  {name}")


# ------------------------------------------------------------------------------------------------ config 2
def run_pointnet(a):
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B = a.batch or 64
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_064))
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["MPointNet"]), "minkowski", ds)
    sd_cpu = {k: v.detach().clone() for k, v in model.model.state_dict().items()}
    pool = [synthetic.make_sparse_batch(list(range(i * B, (i + 1) * B)), n_points=a.points).to(dev) for i in range(3)]
    model.to(dev).train()
    model.init_train_objects(TRAINING_NFI)
    model.reserve_workspace(dev, main_bytes=16 << 30, side_bytes=2 << 30)
    voxels = sum(int(b.coords.shape[0]) for b in pool) / len(pool) / B

    def step(i):
        model.set_input(pool[i % 3], dev)
        model.optimize_parameters(epoch=0, batch_size=B, num_batches=133)
        model.prefetch_input(pool[(i + 1) % 3], dev)

    dt, gaps = timed_loop(step, a.steps, a.warmup)
    with CallTimer() as ct:
        for i in range(3):
            step(i)
    groups = ct.by_name()
    top_table(groups)
    roof, per = dominant_roofline(groups, "config2")
    lvl = step_level(per, dt / a.steps * 1e3, "config2")
    line = dict(metric="training plots/sec (16k-pt NFI plots) MPointNet", value=round(B * a.steps / dt, 2), unit="plots/s",
                n_gpus=1, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3), higher_is_better=True,
                scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload=f"BASELINE config 2: MinkowskiPointNet (shared MLP 6-64-128-1024 + BN + GELU, per-plot sum "
                                     f"pooling, head) training step, {a.points}-pt synthetic plots, batch {B}, ~{voxels:.0f} "
                                     "voxels/plot", global_batch=B, parallelism="dp1",
                            final_loss=round(float(model.loss.detach()), 5)),
                roofline=roof, step_ms_p10=round(gaps[int(len(gaps) * 0.1)], 3), step_ms_p50=round(gaps[len(gaps) // 2], 3),
                step_ms_p90=round(gaps[int(len(gaps) * 0.9)], 3), **lvl,
                entry_points_ms_per_step={n: round(g["ms"] / 3, 3) for n, g in
                                          sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:8]})
    if not a.no_cpu_baseline:
        import bench      # the oracle is only ever imported by bench.py's cpu_baseline legs
        stats = (model.reg_center_targets.cpu(), model.reg_scale_targets.cpu(), model.reg_weights.cpu())
        line["cpu_baseline"] = bench.cpu_baseline_pointnet(sd_cpu, stats, a.points)
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------------------ config 3
def ballquery_cost(args):
    """agb_ball_query_fill(queries, nq, q_elem, origin_cs, dims, cell_start, sorted, radius, ns, width, out, status):
    SURVEY.md §8d: Ns*12 (supports) + Nq*12 (queries) + Nq*width*4 (the padded matrix the API returns)."""
    nq, ns, width = args[1], args[8], args[9]
    return 0.0, ns * 16.0 + nq * 12.0 + nq * width * 4.0


def ballquery_cost_csr(args):
    """agb_ball_query_fill_csr(queries, nq, q_elem, origin_cs, dims, cell_start, sorted, radius, ns, row_ptr, indices,
    capacity, status): SURVEY.md §8d ragged form: Ns*16 (sorted supports) + Nq*12 (queries) + Nq*4 (row_ptr) + sum(counts)*4."""
    nq, ns, total = args[1], args[8], args[11]
    return 0.0, ns * 16.0 + nq * 16.0 + total * 4.0


def run_kpconv(a):
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import KPConvModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    B = a.batch or 32
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_032))
    model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds)
    pool = [synthetic.make_point_batch(list(range(i * B, (i + 1) * B)), n_points=a.points) for i in range(2)]
    for b in pool:
        b.pos, b.x = b.pos.to(dev), b.x.to(dev)
    model.to(dev).train()
    model.init_train_objects(TRAINING_NFI)
    model.reserve_workspace(dev, main_bytes=12 << 30, side_bytes=6 << 30)

    host = dict(set_input=[], optimize=[], prefetch=[])

    def step(i):
        # the next batch's input pyramid is STARTED before this step is enqueued: its count read-backs (one per level) land
        # while the host enqueues the forward / backward pass and are picked up between library calls (instance/kpconv.py)
        t0 = time.perf_counter()
        model.set_input(pool[i % 2], dev)
        t1 = time.perf_counter()
        model.prefetch_input(pool[(i + 1) % 2], dev)
        t2 = time.perf_counter()
        model.optimize_parameters(epoch=0, batch_size=B, num_batches=133)
        t3 = time.perf_counter()
        host["set_input"].append(t1 - t0); host["prefetch"].append(t2 - t1); host["optimize"].append(t3 - t2)

    model.prefetch_input(pool[0], dev)
    dt, gaps = timed_loop(step, a.steps, a.warmup)
    for k, v in host.items():
        v = sorted(v[-a.steps:])
        log(f"host time in {k}: median {v[len(v) // 2] * 1e3:.2f} ms, max {v[-1] * 1e3:.2f} ms")
    # the instrumented steps run the input pyramid on the compute stream (the events are recorded there)
    with CallTimer() as ct:
        for i in range(3):
            model.set_input(pool[i % 2], dev)
            model.optimize_parameters(epoch=0, batch_size=B, num_batches=133)
    groups = ct.by_name()
    top_table(groups, 12)
    if a.shapes:
        shape_table(groups)
    # (agb_ball_query_fill_csr_m: the same argument positions up to `capacity`)
    costs = {"agb_ball_query_fill": ballquery_cost, "agb_ball_query_fill_csr": ballquery_cost_csr,
             "agb_ball_query_fill_csr_m": ballquery_cost_csr,
             "agb_spconv_fwd_opt": conv_call_cost, "agb_spconv_bwd_weight_ws": wgrad_call_cost,
             "agb_spconv_bwd_weight_lp": wgrad_call_cost}
    roof, per = dominant_roofline(groups, "config3", costs)
    lvl = step_level(per, dt / a.steps * 1e3, "config3")
    index_names = ("agb_ball_query_fill", "agb_ball_query_fill_csr", "agb_ball_query_fill_csr_m", "agb_ball_query_offsets",
                   "agb_ball_query_count",
                   "agb_ball_grid_build", "agb_grid_subsample_ws", "agb_elem_bbox", "agb_elem_of_row", "agb_rotate_points")
    index_ms = sum(groups[n]["ms"] for n in index_names if n in groups) / 3
    fill = next(n for n in ("agb_ball_query_fill_csr_m", "agb_ball_query_fill_csr", "agb_ball_query_fill") if n in groups)
    bq = roofline_entry(fill, groups[fill], costs[fill], "config3")
    line = dict(metric="training plots/sec KPConv rigid", value=round(B * a.steps / dt, 2), unit="plots/s", n_gpus=1,
                steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3), higher_is_better=True,
                scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload=f"BASELINE config 3: KPConv rigid (KPCNN, 15 blocks, K = 15) training step incl. the "
                                     f"5-level input pyramid (radius neighbours, grid subsampling) on the device, "
                                     f"{a.points}-pt synthetic plots, batch {B}", global_batch=B, parallelism="dp1",
                            final_loss=round(float(model.loss.detach()), 5)),
                roofline=roof, ball_query_roofline=bq, index_path_ms_per_step=round(index_ms, 3),
                step_ms_p10=round(gaps[int(len(gaps) * 0.1)], 3), step_ms_p50=round(gaps[len(gaps) // 2], 3),
                step_ms_p90=round(gaps[int(len(gaps) * 0.9)], 3), **lvl,
                entry_points_ms_per_step={n: round(g["ms"] / 3, 3) for n, g in
                                          sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:10]})
    if not a.no_cpu_baseline:
        import bench      # the oracle is only ever imported by bench.py's cpu_baseline legs
        line["cpu_baseline"] = bench.cpu_baseline_kpconv_index(pool[0], B, a.points)
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------------------ end to end
INPUT_ENTRY_PREFIXES = ("agb_plot_", "agb_voxelize", "agb_coords_augment", "agb_plot")


def run_end2end(a):
    """The training step FROM RAW POINTS (the reference runs the sparse-xy.yaml chain per sample in its
    DataLoader workers before set_input: conf/data/instance/NFI/transforms/sparse-xy.yaml:4-104, core/data_transform/transforms.py,
    grid_transform.py:112-128): raw 16 000-point plots resident on the device -> SparseTrainPipeline (ground removal, dropout,
    noise, rotation, shift, added / copied points, polygon crop, features, GridSampling3D(last), flip / shift; the random draws
    per sample in DataLoader workers with the reference's generators, applied on the device) -> MSENet14 training step.  The
    chain and the model's input staging run on the input stream two batches ahead of the step that consumes them."""
    import random
    from collections import deque
    from functools import partial
    from torch.utils.data import DataLoader
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import MinkowskiBaselineModel
    from dpcr_agb_amd.train_transforms import NFITrainConfig, SampleDraws, SparseTrainPipeline, collate_draws
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    B = a.batch or 32
    cfg = NFITrainConfig()
    host_pool = []
    for i in range(3):
        raws, ys = [], []
        for seed in range(i * B, (i + 1) * B):
            pos, _, y = synthetic.make_plot(seed, a.points)
            # the raw frame of the reference: metres, centred on the plot centre, heights above an arbitrary datum
            raws.append(np.stack([(pos[:, 0] - 0.5) * 30.0, (pos[:, 1] - 0.5) * 30.0, pos[:, 2] * 40.0 + 3.25], 1).astype(np.float32))
            ys.append(y)
        host_pool.append((raws, np.stack(ys)))
    # The per-sample draws run in DataLoader worker processes (where the reference runs its transforms).  The workers are
    # forked HERE, before this process touches the GPU (a forked copy of an initialised HIP runtime is nothing to rely on).
    workers = 0 if a.inline_draws else max(1, min(6, usable_cores() - 2))
    loader = DataLoader(SampleDraws([r for raws, _ in host_pool for r in raws], cfg, length=1 << 30), batch_size=B, shuffle=False,
                        num_workers=workers, collate_fn=partial(collate_draws, cfg=cfg), pin_memory=workers > 0,
                        worker_init_fn=SampleDraws.seed_worker, persistent_workers=workers > 0,
                        prefetch_factor=4 if workers else None)
    draws_it = iter(loader)
    dev = torch.device("cuda:0")
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_064))
    model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds)
    model.to(dev).train()
    model.init_train_objects(TRAINING_NFI)
    model.reserve_workspace(dev, main_bytes=16 << 30, side_bytes=2 << 30)
    pipe = SparseTrainPipeline(cfg, device_shuffle=not a.host_shuffle)
    pool = [(raws, [torch.from_numpy(r).to(dev) for r in raws], y) for raws, y in host_pool]
    host_draw_ms, voxels, points_in = [], [], []
    # the device part of the chain and the coordinate maps of a batch are enqueued on the model's input stream two steps ahead
    # of its training step, so that the chain's two count read-backs wait for an idle side stream, not for the queued steps
    side = model.input_stream(dev)
    queue = deque()

    from dpcr_agb_amd.train_transforms import StagedBatches
    flight = StagedBatches(pipe, dev, side)

    def stage(i):
        """One turn of the input pipeline: the chain of batch i starts, the batches in flight move one stage on (their count
        read-backs were started a step ago: no wait), finished ones go to the model's input staging."""
        _, raws_d, y = pool[i % 3]
        t0 = time.perf_counter()
        draws = next(draws_it)
        host_draw_ms.append((time.perf_counter() - t0) * 1e3)
        points_in.append(int(draws["n1s"].sum() + draws["n_add"].sum() + draws["n_cj"].sum()))
        if a.sync_chain:        # (the round-5 form: the chain waits for its two read-backs where they occur)
            with torch.cuda.stream(side):
                done = [pipe(raws_d, dev, y_reg=y, draws=draws)]
        else:
            done = flight.advance()
            flight.submit(raws_d, y_reg=y, draws=draws)
        for batch in done:
            voxels.append(int(batch.coords.shape[0]))
            model.prefetch_input(batch, dev)
            queue.append(batch)

    n_stage = 0
    while len(queue) < 2:       # fill the pipeline: two batches staged for the model
        stage(n_stage)
        n_stage += 1
    staged = [n_stage]

    def step(i):
        batch = queue.popleft()
        model.set_input(batch, dev)
        model.optimize_parameters(epoch=0, batch_size=B, num_batches=133)
        stage(staged[0])
        staged[0] += 1

    dt, gaps = timed_loop(step, a.steps, a.warmup)
    floor_ms = host_floor(step, a.warmup + a.steps)
    with CallTimer() as ct:
        for i in range(3):
            step(i)
    groups = ct.by_name()
    top_table(groups, 12)
    inp = {n: g for n, g in groups.items() if n.startswith(INPUT_ENTRY_PREFIXES)}
    pipe_ms = sum(g["ms"] for g in inp.values()) / 3
    dom = max(inp, key=lambda n: inp[n]["ms"])
    g = inp[dom]
    # N (12 + 4 F) + M (12 + 4 F) + 8 N bytes for N points in, M voxels out, F = 3 features
    N, M, F = float(np.mean(points_in[-3:])), float(np.mean(voxels[-3:])), 3
    byts = N * (12 + 4 * F) + M * (12 + 4 * F) + 8 * N
    avg_us = g["ms"] / g["n"] * 1e3
    # PMC bytes of the voxeliser's kernels per batch (k_vox_*: the entry point launches several), committed pass of this line
    vox_traffic, vox_src = pmc_traffic_prefix(("k_vox", "k_voxel"), "end2end") if dom.startswith("agb_voxelize") else (None, None)
    roof = dict(bound="hbm", achieved=round(byts / (avg_us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(byts / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=vox_traffic, traffic_source=vox_src,
                kernel=KERNEL_OF_ENTRY.get(dom, dom), entry_point=dom, launches=g["n"], avg_launch_us=round(avg_us, 2),
                alg_bytes_per_launch=round(byts),
                note="the dominant entry point of the input chain; bytes = the whole chain's per-batch figure of SURVEY 8(d): the "
                     "chain is a dozen launches of 5-60 us each, bound by launch count and two host read-backs, not by bytes")
    hd = sorted(host_draw_ms[-a.steps:])
    line = dict(metric="training plots/sec (16k-pt NFI plots) MSENet14, from raw points", value=round(B * a.steps / dt, 2),
                unit="plots/s", n_gpus=1, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3),
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload=f"raw {a.points}-pt synthetic plots resident on the device -> sparse-xy.yaml train chain on "
                                     f"the device (crop, jitter, rotation, features, GridSampling3D(last) at 0.0125, flip / shift; "
                                     f"per-sample draws on the host, in DataLoader workers) -> SENet14 training step, batch {B}, ~{M / B:.0f} voxels/plot "
                                     "after augmentation", global_batch=B, parallelism="dp1",
                            input_pipeline="side stream, two batches ahead; "
                                           f"per-sample draws in {workers} DataLoader worker process(es)", final_loss=round(float(model.loss.detach()), 5)),
                roofline=roof, step_ms_p10=round(gaps[int(len(gaps) * 0.1)], 3), step_ms_p50=round(gaps[len(gaps) // 2], 3),
                step_ms_p90=round(gaps[int(len(gaps) * 0.9)], 3),
                input_chain_device_ms_per_step=round(pipe_ms, 3), host_draws_ms_per_step_p50=round(hd[len(hd) // 2], 3),
                host_enqueue_floor_ms=floor_ms,
                entry_points_ms_per_step={n: round(gg["ms"] / 3, 3) for n, gg in
                                          sorted(inp.items(), key=lambda kv: -kv[1]["ms"])[:8]})
    print(json.dumps(line), flush=True)


def run_kpconv_e2e(a):
    """The KPConv training step FROM RAW POINTS (SURVEY.md section 8(f)1 for the point models; the reference runs the xy.yaml chain
    per sample in its DataLoader workers: conf/data/instance/NFI/transforms/xy.yaml:4-75 — the sparse chain without the voxel
    tail, MaxPoints 6144): raw plots resident on the device -> PointTrainPipeline (draws per sample in DataLoader workers,
    applied on the device) -> the 5-level input pyramid on the model's side stream -> KPCNN training step."""
    import random
    from collections import deque
    from functools import partial
    from torch.utils.data import DataLoader
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
    from dpcr_agb_amd.instance import KPConvModel
    from dpcr_agb_amd.train_transforms import NFITrainConfig, PointTrainPipeline, SampleDraws, StagedBatches, collate_draws
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    B = a.batch or 32
    cfg = NFITrainConfig(voxel=None, max_points=6144)
    host_pool = []
    for i in range(3):
        raws, ys = [], []
        for seed in range(i * B, (i + 1) * B):
            pos, _, y = synthetic.make_plot(seed, a.points)
            raws.append(np.stack([(pos[:, 0] - 0.5) * 30.0, (pos[:, 1] - 0.5) * 30.0, pos[:, 2] * 40.0 + 3.25], 1).astype(np.float32))
            ys.append(y)
        host_pool.append((raws, np.stack(ys)))
    workers = 0 if a.inline_draws else max(1, min(6, usable_cores() - 2))
    loader = DataLoader(SampleDraws([r for raws, _ in host_pool for r in raws], cfg, length=1 << 30), batch_size=B, shuffle=False,
                        num_workers=workers, collate_fn=partial(collate_draws, cfg=cfg), pin_memory=workers > 0,
                        worker_init_fn=SampleDraws.seed_worker, persistent_workers=workers > 0,
                        prefetch_factor=4 if workers else None)
    draws_it = iter(loader)
    dev = torch.device("cuda:0")
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_032))
    model = KPConvModel(Opt(MODEL_OPTIONS["KPConv"]), "kpconv", ds)
    model.to(dev).train()
    model.init_train_objects(TRAINING_NFI)
    model.reserve_workspace(dev, main_bytes=8 << 30, side_bytes=4 << 30)
    pipe = PointTrainPipeline(cfg)
    pool = [(raws, [torch.from_numpy(r).to(dev) for r in raws], y) for raws, y in host_pool]
    side = torch.cuda.Stream(device=dev)
    queue, points = deque(), []
    flight = StagedBatches(pipe, dev, side)

    def stage(i):
        _, raws_d, y = pool[i % 3]
        draws = next(draws_it)
        done = flight.advance()
        flight.submit(raws_d, y_reg=y, draws=draws)
        for batch in done:
            points.append(int(batch.pos.shape[0]))
            # the pyramid's side stream continues where the chain's stream left the batch
            ev = torch.cuda.Event()
            ev.record(side)
            batch.ready = ev
            queue.append(batch)

    n_stage = 0
    while len(queue) < 2:
        stage(n_stage)
        n_stage += 1
    staged = [n_stage]

    def start_pyramid(batch):
        if not hasattr(model, "_side_stream"):
            model._side_stream = torch.cuda.Stream(device=dev)
        model._side_stream.wait_event(batch.ready)
        model.prefetch_input(batch, dev)

    start_pyramid(queue[0])

    def step(i):
        batch = queue.popleft()
        model.set_input(batch, dev)
        stage(staged[0])
        staged[0] += 1
        start_pyramid(queue[0])           # the next batch's pyramid starts before this step is enqueued
        model.optimize_parameters(epoch=0, batch_size=B, num_batches=133)

    dt, gaps = timed_loop(step, a.steps, a.warmup)
    floor_ms = host_floor(step, a.warmup + a.steps)
    line = dict(metric="training plots/sec KPConv rigid, from raw points", value=round(B * a.steps / dt, 2), unit="plots/s", n_gpus=1,
                steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload=f"raw {a.points}-pt synthetic plots resident on the device -> xy.yaml train chain on the device "
                                     f"(ground removal, dropout, jitter, rotation, added / copied points, polygon crop, MaxPoints 6144, "
                                     f"features; per-sample draws on the host, in DataLoader workers) -> 5-level input pyramid -> KPCNN "
                                     f"training step, batch {B}, {np.mean(points[-3:]) / B:.0f} points/plot", global_batch=B,
                            parallelism="dp1", final_loss=round(float(model.loss.detach()), 5)),
                step_ms_p10=round(gaps[int(len(gaps) * 0.1)], 3), step_ms_p50=round(gaps[len(gaps) // 2], 3),
                step_ms_p90=round(gaps[int(len(gaps) * 0.9)], 3), host_enqueue_floor_ms=floor_ms)
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["pointnet", "kpconv", "end2end", "kpconv_e2e"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inline-draws", action="store_true", help="end2end: per-sample draws in the training process, no workers")
    ap.add_argument("--host-shuffle", action="store_true", help="end2end: GridSampling3D's shuffle with torch.randperm on the host")
    ap.add_argument("--sync-chain", action="store_true", help="end2end: the transform chain waits for its two count read-backs "
                    "where they occur (round 5) instead of picking them up a step later (train_transforms.StagedBatches)")
    ap.add_argument("--shapes", action="store_true", help="per-shape table of the dense products / gathers")
    a = ap.parse_args()
    # (device_count does not initialise the HIP runtime, is_available does: the end2end line forks its loader workers first)
    if torch.cuda.device_count() < 1:
        raise SystemExit("needs a HIP device")
    # The host side of a step only issues launches and shuffles a few small CPU tensors: keep torch's intra-op pool small.
    # With one OpenMP thread per VISIBLE core (256 on the GPU box, 16 allowed by the cgroup quota) the spinning workers
    # exhaust the CPU quota and the kernel throttles the whole process for the rest of the 100 ms period (the 50-150 ms
    # hiccups of the KPConv loop: host stalls with zero device allocations and no pageable copy in flight).
    import dpcr_agb_amd
    dpcr_agb_amd.limit_host_threads()
    {"pointnet": run_pointnet, "kpconv": run_kpconv, "end2end": run_end2end, "kpconv_e2e": run_kpconv_e2e}[a.which](a)


if __name__ == "__main__":
    main()
