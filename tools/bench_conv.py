"""Micro-benchmark of the sparse-conv kernels on the real SENet14 pyramid of a synthetic batch (tuning aid).

  python tools/bench_conv.py [--modes 0,64,128] [--reps 20]
For every (level, Cin, Cout) of the stride-1 3^3 layers it times agb_spconv_fwd_ex per kernel-selection mode
(sparse_ops.KernelOptions.cmp_mode) and the weight gradient, and prints us/launch and algorithmic TFLOP/s (2*pairs*Cin*Cout)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="0,64,128")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--b16", action="store_true", help="also time the forward pass on bf16 rows (agb_spconv_fwd_h)")
    ap.add_argument("--il", default="-1", help="interleave block shifts to sweep for the pair-compacted kernel, e.g. 0,2,3 (-1: by level size, the library default)")
    args = ap.parse_args()
    from dpcr_agb_amd import _lib, sparse_ops, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    from dpcr_agb_amd.sparse_ops import spconv_forward_raw
    dev = torch.device("cuda", 0)
    b = synthetic.make_sparse_batch(list(range(args.batch)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=args.batch, bounds=b.coord_bounds)
    ts = 1
    cases = []
    for ts_in, c in ((2, 64), (4, 128), (8, 256), (16, 512)):
        while ts < ts_in:
            cm.stride(ts, 2)
            ts *= 2
        n = cm.level(ts_in).n
        nbr = cm.kernel_map(ts_in, 3, 1)
        pairs = int((nbr[:, :n] >= 0).sum())
        cases.append((ts_in, c, c, n, nbr, pairs))
    for ts_in, cin, cout, n, nbr, pairs in cases:
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27 * cin, cout, device=dev) * 0.05
        ref = None
        for mode, il in [(int(m), int(i)) for m in args.modes.split(",") for i in args.il.split(",")]:
            sparse_ops.DEFAULTS.cmp_mode, sparse_ops.DEFAULTS.cmp_interleave = mode, il
            y = spconv_forward_raw(x, w, nbr, 0, None, n, 27, cin, cout)
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            err = float((y - ref).abs().max() / ref.abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                spconv_forward_raw(x, w, nbr, 0, None, n, 27, cin, cout)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.reps * 1e3
            print(f"ts{ts_in:2d} {cin:4d}->{cout:4d} rows {n:7d} density {pairs / (27 * n):.2f} mode {mode:3d} il {il}: "
                  f"{us:8.1f} us  {2.0 * pairs * cin * cout / us / 1e6:6.1f} TF  (max rel diff vs first mode {err:.1e})",
                  flush=True)
        if args.b16:
            x16 = x.to(torch.bfloat16)
            wkm = w.view(27, cin, cout).transpose(1, 2).contiguous()
            opts = sparse_ops.KernelOptions(precision="bf16")
            spconv_forward_raw(x16, None, nbr, 0, None, n, 27, cin, cout, w_kmajor=wkm, opts=opts)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                spconv_forward_raw(x16, None, nbr, 0, None, n, 27, cin, cout, w_kmajor=wkm, opts=opts)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.reps * 1e3
            print(f"ts{ts_in:2d} {cin:4d}->{cout:4d} fwd bf16 rows (incl. the weight conversion launch): {us:8.1f} us  "
                  f"{2.0 * pairs * cin * cout / us / 1e6:6.1f} TF", flush=True)
        if args.wgrad:
            dy = torch.randn(n, cout, device=dev)
            ref_dw = None
            # register-operand kernel with the fixed-order fold / with atomic accumulation / the LDS-staged kernel
            x16, dy16 = x.to(torch.bfloat16), dy.to(torch.bfloat16)
            for tag, kw in (("reg+fold", dict(deterministic_wgrad=True)), ("reg+atomic", dict(dw_variant=2)), ("staged", {}),
                            ("bf16 rows", dict(precision="bf16"))):
                opts = sparse_ops.KernelOptions(**kw)
                dw = torch.zeros(27, cin, cout, device=dev)
                if tag == "bf16 rows":
                    x, dy = x16, dy16
                sparse_ops.weight_grad_raw(x, dy, nbr, dw, n, 27, cin, cout, opts)
                torch.cuda.synchronize()
                if ref_dw is None:
                    ref_dw = dw.clone()
                err = float((dw - ref_dw).abs().max() / ref_dw.abs().max())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    sparse_ops.weight_grad_raw(x, dy, nbr, dw, n, 27, cin, cout, opts)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / args.reps * 1e3
                print(f"ts{ts_in:2d} {cin:4d}->{cout:4d} wgrad {tag:10s}: {us:8.1f} us  "
                      f"{2.0 * pairs * cin * cout / us / 1e6:6.1f} TF  (max rel diff vs first {err:.1e})", flush=True)
    sparse_ops.DEFAULTS.cmp_mode, sparse_ops.DEFAULTS.cmp_interleave = 1, -1


if __name__ == "__main__":
    main()
