"""A/B of the fp32 forward kernels in ONE process, interleaved repetitions per shape (the four 3^3 stride-1 levels of the
SENet14 pyramid of a synthetic batch of 32 plots).
   python tools/cma_ab.py 128 129 0     (KernelOptions.cmp_mode: 128 = k_spconv_cma, the hand-scheduled product kernel;
                                         129 = k_spconv_cmpt, its C++ twin; 0 = k_spconv_pipe, register accumulators)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    modes = sys.argv[1:] or ["128", "129"]      # "128:3" = cmp_mode 128 with cmp_interleave 3; "128:-1:0" = ... without balanced tiles
    from dpcr_agb_amd import sparse_ops, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    from dpcr_agb_amd.sparse_ops import spconv_forward_raw
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    b = synthetic.make_sparse_batch(list(range(32)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=32, bounds=b.coord_bounds)
    ts = 1
    for ts_in, c in ((2, 64), (4, 128), (8, 256), (16, 512)):
        while ts < ts_in:
            cm.stride(ts, 2)
            ts *= 2
        n = cm.level(ts_in).n
        nbr = cm.kernel_map(ts_in, 3, 1)
        x = torch.randn(n, c, device=dev)
        w = torch.randn(27 * c, c, device=dev) * 0.05
        tot = {m: 0.0 for m in modes}
        ref = None
        res = {}
        for rep in range(6):
            for m in modes:
                parts = m.split(":")
                sparse_ops.DEFAULTS.cmp_mode = int(parts[0])
                sparse_ops.DEFAULTS.cmp_interleave = int(parts[1]) if len(parts) > 1 else -1
                sparse_ops.DEFAULTS.balanced_tiles = bool(int(parts[2])) if len(parts) > 2 else True
                y = spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
                torch.cuda.synchronize()
                if rep == 0:
                    if ref is None:
                        ref = y.clone()
                    res[m] = (bool(torch.equal(y, ref)), float((y - ref).abs().max() / ref.abs().max()))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    tot[m] += e0.elapsed_time(e1) / 5 * 1e3
        print(f"ts{ts_in:2d} {c:3d}->{c:3d}: " + "  ".join(f"[{m}] {tot[m] / 5:6.1f} us{'' if res[m][0] else f' (diff {res[m][1]:.1e})'}" for m in modes), flush=True)


if __name__ == "__main__":
    main()
