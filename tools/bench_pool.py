"""Micro-benchmark of the strided max pooling (3^3, stride 2, 64 channels after the stem) on a full synthetic batch.

  python tools/bench_pool.py [--reps 20] [--batch 32]
Prints us/launch and the effective bandwidth on the algorithmic bytes (input rows once, output + argmax rows, maps)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--channels", type=int, default=64)
    args = ap.parse_args()
    from dpcr_agb_amd import _lib, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    dev = torch.device("cuda", 0)
    b = synthetic.make_sparse_batch(list(range(args.batch)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=args.batch, bounds=b.coord_bounds)
    nbr = cm.kernel_map(1, 3, 2)
    nbrT = cm.transposed_map(1, 3, 2)
    n_in, n_out = cm.level(1).n, cm.level(2).n
    C = args.channels
    x = torch.randn(n_in, C, device=dev)
    y = torch.empty(n_out, C, device=dev)
    arg = torch.empty(n_out, C, dtype=torch.int32, device=dev)
    arg8 = torch.empty(n_out, C, dtype=torch.uint8, device=dev)
    dy = torch.randn(n_out, C, device=dev)
    dx = torch.empty(n_in, C, device=dev)
    P = lambda t: t.data_ptr()   # noqa: E731
    us = timed(lambda: _lib.call("agb_maxpool_fwd", P(x), C, P(nbr), nbr.stride(0), P(y), C, P(arg), n_out, 27, C,
                                 _lib.stream()), args.reps)
    byts = (n_in * C + 2 * n_out * C) * 4 + 27 * n_out * 4
    print(f"rows {n_in} -> {n_out}; fwd {us:7.1f} us  {byts / us / 1e3:6.0f} GB/s on {byts / 1e6:.0f} MB")
    us = timed(lambda: _lib.call("agb_maxpool_bwd", P(dy), C, P(arg), P(nbrT), nbrT.stride(0), P(dx), C, n_in, 27, C,
                                 _lib.stream()), args.reps)
    byts = (n_in * C + 2 * n_out * C) * 4 + 27 * n_in * 4
    print(f"bwd {us:7.1f} us  {byts / us / 1e3:6.0f} GB/s on {byts / 1e6:.0f} MB")
    us = timed(lambda: _lib.call("agb_maxpool_fwd_k", P(x), C, P(nbr), nbr.stride(0), P(y), C, P(arg8), n_out, 27, C,
                                 _lib.stream()), args.reps)
    print(f"byte-argmax variant: fwd {us:7.1f} us", end="")
    us = timed(lambda: _lib.call("agb_maxpool_bwd_k", P(dy), C, P(arg8), P(nbrT), nbrT.stride(0), P(dx), C, n_in, 27, C,
                                 _lib.stream()), args.reps)
    print(f"  bwd {us:7.1f} us")


if __name__ == "__main__":
    main()
