"""A/B of the fused KPConv layer (csrc/kpfused.hip, kpconv_ops.KPConvFusedFunction) against the two-kernel form
(KPConvSymmetricFunction: gather -> wf in HBM -> dense product) on the real input pyramid of B synthetic plots:
numerical agreement (output, dx, dW) and interleaved timings of forward and forward + backward.

    python tools/kpfused_ab.py --plots 32 --points 16000 --reps 20
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plots", type=int, default=32)
    ap.add_argument("--points", type=int, default=16000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--levels", type=str, default="0,1")
    ap.add_argument("--channels", type=str, default="16,32")
    ap.add_argument("--sort", type=float, default=0.0, help="cell size (m) of a spatial sort of every plot's points first (0: input order)")
    args = ap.parse_args()
    import dpcr_agb_amd  # noqa: F401
    from dpcr_agb_amd import synthetic
    from dpcr_agb_amd.config import MODEL_OPTIONS, Opt
    from dpcr_agb_amd.instance import KPConvModel
    from dpcr_agb_amd.kpconv_ops import KPConvFusedFunction, KPConvSymmetricFunction
    from dpcr_agb_amd.sparse_ops import current
    import dpcr_agb_amd.backbones.kpconv as KB
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_016))
    opt = Opt(MODEL_OPTIONS["KPConv"])
    model = KPConvModel(opt, "kpconv", ds).to(dev).train()
    b = synthetic.make_point_batch(list(range(args.plots)), n_points=args.points)
    lens = np.bincount(b.batch.numpy()).astype(np.int64)
    if args.sort > 0:
        cell = torch.floor((b.pos - b.pos.min(0).values) / args.sort).long()
        key = ((b.batch.long() * 4096 + cell[:, 2]) * 4096 + cell[:, 1]) * 4096 + cell[:, 0]
        perm = torch.argsort(key, stable=True)
        b.pos, b.x, b.batch = b.pos[perm].contiguous(), b.x[perm].contiguous(), b.batch[perm].contiguous()
    inp = model.prepare_inputs(b.pos, b.x, lens, dev)
    cfg = opt.config
    for lvl in [int(v) for v in args.levels.split(",")]:
        pts, nb = inp["points"][lvl], inp["neighbors"][lvl]
        N = pts.shape[0]
        radius = cfg.first_subsampling_dl * cfg.conv_radius * 2 ** lvl
        extent = radius * cfg.KP_extent / cfg.conv_radius
        valid = int(nb.indices.shape[0])
        for C in [int(v) for v in args.channels.split(",")]:
            conv = KB.KPConv(15, 3, C, C, extent, radius).to(dev)
            x0 = torch.randn(N, C, device=dev)
            gy = torch.randn(N, C, device=dev)
            assert KPConvFusedFunction.supported(15, C, C, nb, current()), "fused kernel does not cover this layer"

            def run(fn, backward=True):
                x = x0.clone().requires_grad_(True)
                conv.weights.grad = None
                y = fn.apply(x, pts, nb, conv.kernel_points, conv.KP_extent, conv.weights)
                if backward:
                    y.backward(gy)
                    return y.detach(), x.grad, conv.weights.grad
                return y.detach(), None, None

            ref = run(KPConvSymmetricFunction)
            got = run(KPConvFusedFunction)
            got2 = run(KPConvFusedFunction)
            torch.cuda.synchronize()
            rel = lambda a, r: float((a - r).abs().max() / r.abs().max())  # noqa: E731
            print(f"[kpfused_ab] level {lvl}: N {N}, {valid} pairs ({valid / N:.1f} per row), C {C}: rel. difference "
                  f"out {rel(got[0], ref[0]):.2e}, dx {rel(got[1], ref[1]):.2e}, dW {rel(got[2], ref[2]):.2e}; repeat bitwise "
                  f"{all(torch.equal(a, c) for a, c in zip(got, got2))}", flush=True)
            times = {}
            for name, fn in (("two-kernel", KPConvSymmetricFunction), ("fused", KPConvFusedFunction)):
                for bw in (False, True):
                    times[(name, bw)] = []
            for rep in range(args.reps + 3):
                for name, fn in (("two-kernel", KPConvSymmetricFunction), ("fused", KPConvFusedFunction)):
                    for bw in (False, True):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        x = x0.clone().requires_grad_(True)
                        conv.weights.grad = None
                        e0.record()
                        if bw:
                            y = fn.apply(x, pts, nb, conv.kernel_points, conv.KP_extent, conv.weights)
                            y.backward(gy)
                        else:
                            with torch.no_grad():
                                y = fn.apply(x, pts, nb, conv.kernel_points, conv.KP_extent, conv.weights)
                        e1.record()
                        torch.cuda.synchronize()
                        if rep >= 3:
                            times[(name, bw)].append(e0.elapsed_time(e1))
            for bw in (False, True):
                a, f = np.median(times[("two-kernel", bw)]), np.median(times[("fused", bw)])
                print(f"[kpfused_ab]     {'fwd + bwd' if bw else 'fwd      '}: two-kernel {a:.3f} ms, fused {f:.3f} ms  ({a / f:.2f}x)",
                      flush=True)


if __name__ == "__main__":
    main()
