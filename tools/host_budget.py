"""Host-CPU budget of an N-rank data-parallel run, MEASURED on a one-GPU box (not a scaling number: the ranks share the one
GPU and exchange gradients through gloo; what is measured is what N ranks ask of the node's CORES per training step).

  python tools/host_budget.py [--ranks 8] [--steps 12] [--warmup 4] [--prefetch]

Starts `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` as a fresh child process with
AGB_BENCH_BACKEND=gloo (bench.py's dry-run mode: ranks map to device rank % device_count) and a small allocator reserve so
that N ranks fit one device, then prints per rank the CPU time of all threads per step (comm.host_cpu_ms_per_step_p50), the
enqueue time per step, the cores N ranks need at the single-GPU step time of the headline, and the usable cores of this box.
"""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--prefetch", action="store_true", help="keep the side-stream input pipeline whatever the cores per rank")
    ap.add_argument("--single-gpu-step-ms", type=float, default=8.35, help="step time of the one-GPU headline run")
    a = ap.parse_args()
    import bench
    cores = bench.usable_cores(None)
    env = dict(os.environ, AGB_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(a.ranks), "--steps", str(a.steps),
           "--warmup", str(a.warmup), "--no-cpu-baseline", "--reserve-gib", "4"] + (["--force-prefetch"] if a.prefetch else [])
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    if res.returncode != 0:
        print(res.stderr[-3000:])
        raise SystemExit(res.returncode)
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    comm = line["comm"]
    cpu = comm["host_cpu_ms_per_step_p50"]
    enq = comm["host_enqueue_ms_p50"]
    need = sum(cpu) / a.single_gpu_step_ms
    out = dict(ranks=a.ranks, usable_cores=cores, input_pipeline=line["config"]["input_pipeline"],
               host_cpu_ms_per_step_p50_per_rank=cpu, host_enqueue_ms_p50_per_rank=enq,
               shared_gpu_ms_per_step=line["ms_per_step"], single_gpu_step_ms=a.single_gpu_step_ms,
               cores_needed_at_single_gpu_step=round(need, 2), fits=bool(need <= cores),
               param_checksums_identical=len(set(comm["param_checksums"])) == 1,
               note="ranks share ONE GPU over gloo: a measurement of host CPU demand per step, not of scaling")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
