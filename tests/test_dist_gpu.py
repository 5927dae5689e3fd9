"""The data-parallel path of bench.py as TWO PROCESSES on the GPU box (one MI355X there, so the ranks share it and the
gradients travel through gloo — AGB_BENCH_BACKEND=gloo; RCCL needs one GPU per rank and runs only in the driver's
scaling bench).  What this covers that tests/test_dist_cpu.py cannot: the bucket path fed by the HIP backward, the fused
AdaBelief reading bucket slices, the side-stream input pipeline, per-rank seeds — under torch.distributed.run, as the
driver launches it.  Not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--precision", "bf16", "--bf16-rows"]], ids=["fp32", "bf16rows"])
def test_two_rank_bench_dry_run_keeps_ranks_identical(mode):
    """mode bf16rows: BASELINE config 5's operand / storage mode (bf16 rows; gradients, buckets and optimiser state fp32)."""
    env = dict(os.environ, AGB_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None)
    # a fresh child process (fork + exec of a NEW interpreter, never an exec of this GPU-initialised one)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--reserve-gib", "8"] + mode
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["scaling"] == "weak"
    comm = line["comm"]
    assert comm["backend"] == "gloo" and comm["world"] == 2 and comm["buckets"] >= 2
    assert comm["bytes_per_step"] >= 14_000_000 * 4      # SENet14: 14.45 M parameters exchanged every step
    a, b = comm["param_checksums"]
    assert a == b and a != 0.0, comm                     # averaged gradients -> bit-identical replicas after 8 steps
    assert all(v > 0 for v in comm["host_cpu_ms_per_step_p50"])
    assert line["value"] > 0 and line["roofline"] is not None


def _bench_line(extra_env, extra_args=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-other-configs",
           "--deterministic", "--reserve-gib", "8"] + list(extra_args)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_rccl_runs_in_a_group_of_one_rank():
    """RCCL on the one GPU of the test box: bench.py as a fresh child process with AGB_FORCE_GRAD_SYNC=1 initialises
    ``init_process_group("nccl", device_id=...)`` with ONE rank and GradAllReduce issues the ``ReduceOp.AVG`` all-reduce of
    every gradient bucket from the backward hooks (dist.py, force_collective) — the collective, its stream hand-off against the
    HIP kernels' stream and the fused optimiser reading the reduced bucket slices all execute.  The average over one rank is
    the identity: with fixed-order weight-gradient sums (--deterministic) the parameters after eight steps equal those of
    the run on the same buckets without any exchange (AGB_FORCE_GRAD_SYNC=buckets) BIT FOR BIT.  Not a scaling statement: no
    multi-GPU box was available."""
    plain = _bench_line({"AGB_FORCE_GRAD_SYNC": "buckets"})
    # (the second run also takes the allocator's way of keeping side-stream-built inputs alive — Tensor.record_stream — instead
    # of the default hold-until-the-step-is-done of InstanceBase._hold_input: a lifetime bug in either shows up as a checksum
    # that differs or is not finite)
    forced = _bench_line({"AGB_FORCE_GRAD_SYNC": "1", "AGB_INPUT_RECORD_STREAM": "1"})
    assert plain["comm"]["backend"] is None and plain["comm"]["buckets"] == forced["comm"]["buckets"]
    c = forced["comm"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["buckets"] >= 2 and c["bytes_per_step"] >= 14_000_000 * 4
    assert c["param_checksums"][0] == plain["comm"]["param_checksums"][0] != 0.0, (c, plain["comm"])
    assert forced["n_gpus"] == 1 and forced["value"] > 0


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as a plain command line (no torch.distributed.run around it, no WORLD_SIZE in the
    environment): bench.py starts its ranks itself as a fresh child process and relays rank 0's ONE JSON line.  On the
    one-GPU test box the two ranks share the device (AGB_BENCH_BACKEND=gloo); where the node has a device per rank the
    backend must be RCCL."""
    import torch
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    have_two = torch.cuda.device_count() >= 2
    if not have_two:
        env["AGB_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--deterministic",
           "--reserve-gib", "8"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64
    comm = line["comm"]
    assert comm["world"] == 2 and comm["backend"] == ("nccl" if have_two else "gloo")
    a, b = comm["param_checksums"]
    assert a == b and a != 0.0, comm


def test_bench_refuses_more_ranks_than_devices_without_the_dry_run_switch():
    """No GPU call, no child process: the plain command line on a node with too few devices says what to do."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AGB_BENCH_BACKEND"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "64"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode != 0 and "AGB_BENCH_BACKEND=gloo" in res.stderr
