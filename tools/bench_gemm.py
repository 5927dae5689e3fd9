"""Micro-benchmark of the dense products of the path (identity-map convolution kernels): forward / data gradient
(agb_spconv_fwd_opt / _lp with nbr == NULL) and weight gradient (agb_spconv_bwd_weight_lp with nbr == NULL) on the shapes of
MPointNet, MSENet50 and KPConv.  Usage: python tools/bench_gemm.py [--precision fp32|bf16|bf16x3] [--reps 20] [--only fwd|wgrad]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = [  # (rows, Cin, Cout, what)
    (823_000, 128, 1024, "PointNet L3"), (823_000, 1024, 128, "PointNet L3 dgrad"), (823_000, 64, 128, "PointNet L2"),
    (211_000, 64, 256, "SENet50 s1 conv3"), (211_000, 256, 64, "SENet50 s1 conv1"), (61_000, 128, 512, "SENet50 s2 conv3"),
    (61_000, 512, 128, "SENet50 s2 conv1"), (14_000, 256, 1024, "SENet50 s3 conv3"), (14_000, 1024, 256, "SENet50 s3 conv1"),
    (2_900, 512, 2048, "SENet50 s4 conv3"), (2_900, 2048, 512, "SENet50 s4 conv1"),
    (506_000, 240, 16, "KPConv L0"), (213_000, 480, 32, "KPConv L1"), (64_000, 960, 64, "KPConv L2"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--b16", action="store_true", help="bf16 precision on bf16 ROWS (the bf16-activation mode's kernels)")
    a = ap.parse_args()
    from dpcr_agb_amd import _lib, sparse_ops
    from dpcr_agb_amd.sparse_ops import spconv_forward_raw
    sparse_ops.set_conv_precision(a.precision)
    prec = {"fp32": 0, "bf16": 1, "bf16x3": 2}[a.precision]
    dev = torch.device("cuda", 0)
    for n, cin, cout, what in SHAPES:
        if a.b16 and "SENet50" not in what:
            continue
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(cin, cout, device=dev) * 0.05
        wkm = w.t().contiguous() if prec else None
        dy = torch.randn(n, cout, device=dev)
        dw = torch.zeros(cin, cout, device=dev)

        x16, dy16 = x.to(torch.bfloat16), dy.to(torch.bfloat16)

        def fwd():
            if a.b16:
                return spconv_forward_raw(x16, None, None, 0, None, n, 1, cin, cout, "fwd1x1", None, None, wkm)
            return spconv_forward_raw(x, None if prec else w, None, 0, None, n, 1, cin, cout, "fwd1x1", None, None, wkm)

        def wgrad():
            if a.b16:
                _lib.call("agb_spconv_bwd_weight_b16", x16.data_ptr(), cin, dy16.data_ptr(), cout, None, 0, dw.data_ptr(), n, 1,
                          cin, cout, _lib.stream())
                return
            _lib.call("agb_spconv_bwd_weight_lp", x.data_ptr(), cin, dy.data_ptr(), cout, None, 0, dw.data_ptr(), n, 1, cin,
                      cout, prec, _lib.stream())

        for name, fn in (("fwd", fwd), ("wgrad", wgrad)):
            if a.only and a.only != name:
                continue
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            byts = (n * (cin + cout) + cin * cout) * 4
            print(f"{what:20s} {name:5s} [{n:7d} x {cin:4d}] x [{cin:4d} x {cout:4d}] {a.precision}: {us:8.1f} us  "
                  f"{2.0 * n * cin * cout / us / 1e6:6.1f} TF  {byts / us / 1e3:6.0f} GB/s compulsory", flush=True)


if __name__ == "__main__":
    main()
