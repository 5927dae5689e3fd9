"""Micro-benchmark of the 3 -> 64 channel 7^3 stem convolution on a full synthetic batch (tuning aid).

  python tools/bench_stem.py [--reps 20] [--batch 32]
Times the forward and the weight gradient through the pair-sparse vector kernels (csrc/stem.hip) and through the dense
MFMA kernels they replace, and prints us/launch, algorithmic TFLOP/s (2*pairs*3*64) and the difference between the two."""
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
    ap.add_argument("--kernel", type=int, default=7)
    args = ap.parse_args()
    from dpcr_agb_amd import _lib, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    from dpcr_agb_amd.sparse_ops import spconv_forward_raw
    dev = torch.device("cuda", 0)
    b = synthetic.make_sparse_batch(list(range(args.batch)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=args.batch, bounds=b.coord_bounds)
    n = cm.level(1).n
    K3 = args.kernel ** 3
    nbr = cm.kernel_map(1, args.kernel, 1)
    pairs = int((nbr[:, :n] >= 0).sum())
    print(f"rows {n}, offsets {K3}, pairs {pairs} ({pairs / n:.1f} per row, density {pairs / (K3 * n):.3f})")
    x = torch.zeros(n, 4, device=dev)
    x[:, :3] = torch.randn(n, 3, device=dev)
    w = torch.randn(K3, 3, 64, device=dev) * 0.05
    dy = torch.randn(n, 64, device=dev)
    flops = 2.0 * pairs * 3 * 64
    outs = {}
    for mode in (1, 0):
        _lib.call("agb_spconv_set_stem_mode", mode)
        y = spconv_forward_raw(x, w.view(K3 * 3, 64), nbr, 0, None, n, K3, 3, 64)
        outs[mode] = y
        us = timed(lambda: spconv_forward_raw(x, w.view(K3 * 3, 64), nbr, 0, None, n, K3, 3, 64), args.reps)
        print(f"fwd   {'sparse' if mode else 'dense '}: {us:8.1f} us  {flops / us / 1e6:6.1f} TF", flush=True)
    err = float((outs[1] - outs[0]).abs().max() / outs[0].abs().max())
    print(f"fwd   max rel diff sparse vs dense: {err:.2e}")
    lib = _lib.load()
    scratch = torch.empty(lib.agb_spconv_bwd_weight3_scratch(n, K3), device=dev)
    dk = torch.empty(K3, 3, 64, device=dev)

    def wg_sparse():
        _lib.call("agb_spconv_bwd_weight3", x.data_ptr(), dy.data_ptr(), 64, nbr.data_ptr(), nbr.stride(0),
                  dk.data_ptr(), scratch.data_ptr(), n, K3, 64, _lib.stream())
    dwp = torch.zeros(K3, 4, 64, device=dev)

    def wg_dense():
        _lib.call("agb_spconv_bwd_weight", x.data_ptr(), 4, dy.data_ptr(), 64, nbr.data_ptr(), nbr.stride(0),
                  dwp.data_ptr(), n, K3, 4, 64, _lib.stream())
    us = timed(wg_sparse, args.reps)
    print(f"wgrad sparse: {us:8.1f} us  {flops / us / 1e6:6.1f} TF")
    us = timed(wg_dense, args.reps)
    print(f"wgrad dense : {us:8.1f} us  {flops / us / 1e6:6.1f} TF")
    dwp.zero_()
    wg_dense()
    wg_sparse()
    torch.cuda.synchronize()
    err = float((dk - dwp[:, :3]).abs().max() / dwp.abs().max())
    print(f"wgrad max rel diff sparse vs dense: {err:.2e}")
    _lib.call("agb_spconv_set_stem_mode", 1)


if __name__ == "__main__":
    main()
