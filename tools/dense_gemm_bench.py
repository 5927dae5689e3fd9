"""Device time of the dense products of the point MLP's wide layer (config 2: [N, 128] x [128, 1024], its data gradient
[N, 1024] x [1024, 128] and its weight gradient) on this library's kernels, with torch.mm (rocBLAS / hipBLASLt fp32) beside them.
`python tools/dense_gemm_bench.py [--rows 870000] [--reps 10]`."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=870000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shapes", default="128x1024,1024x128,64x128,128x64,256x64,64x256")
    args = ap.parse_args()
    import dpcr_agb_amd  # noqa: F401
    from dpcr_agb_amd import _lib, sparse_ops as so
    dev = torch.device("cuda", 0)
    n = args.rows
    for shp in args.shapes.split(","):
        cin, cout = (int(t) for t in shp.split("x"))
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(cin, cout, device=dev) / cin ** 0.5
        dy = torch.randn(n, cout, device=dev)
        flop = 2.0 * n * cin * cout
        byts = 4.0 * (n * cin + n * cout + cin * cout)
        t_own = timed(lambda: so.spconv_forward_raw(x, w, None, 0, None, n, 1, cin, cout, "fwd1x1"), args.reps)
        kern = _lib.last_kernel()
        t_bn = timed(lambda: so.spconv_forward_raw(x, w, None, 0, None, n, 1, cin, cout, "fwd1x1", bn_stats=True), args.reps)
        print(f"    with the BatchNorm-partials epilogue (agb_dense_fwd_bn): {t_bn * 1e3:8.1f} us")
        t_mm = timed(lambda: torch.mm(x, w), args.reps)
        print(f"[{n} x {cin}] @ [{cin} x {cout}]: own {t_own * 1e3:8.1f} us ({flop / t_own / 1e9:6.1f} TF, {byts / t_own / 1e9:5.2f} TB/s; "
              f"{kern})   torch.mm {t_mm * 1e3:8.1f} us ({flop / t_mm / 1e9:6.1f} TF)")
        dw = torch.zeros(cin, cout, device=dev)
        t_wg = timed(lambda: so.weight_grad_raw(x, dy, None, dw, n, 1, cin, cout, so.current()), args.reps)
        kern = _lib.last_kernel()
        t_wmm = timed(lambda: torch.mm(x.t(), dy), args.reps)
        print(f"    weight gradient [{cin} x {n}] @ [{n} x {cout}]: own {t_wg * 1e3:8.1f} us ({flop / t_wg / 1e9:6.1f} TF; {kern})   "
              f"torch.mm {t_wmm * 1e3:8.1f} us ({flop / t_wmm / 1e9:6.1f} TF)")
        del x, w, dy


if __name__ == "__main__":
    main()
