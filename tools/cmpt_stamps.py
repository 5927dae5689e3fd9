"""In-kernel stamp shares of k_spconv_cmpt (diagnostic build: make timeline; the stamps' fences forbid overlaps the real kernel has:
read the SHARES).  AGB_CMPT=257 python tools/cmpt_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AGB_LIBRARY", os.path.join(ROOT, "dpcr-agb_amd", "libagbhip_timeline.so"))
os.environ.setdefault("AGB_CMPT", "257")
import torch  # noqa: E402


def main():
    from dpcr_agb_amd import _lib, sparse_ops, synthetic
    from dpcr_agb_amd.coords import CoordinateManager
    from dpcr_agb_amd.sparse_ops import spconv_forward_raw
    L = _lib.load()
    dev = torch.device("cuda", 0)
    b = synthetic.make_sparse_batch(list(range(32)))
    coords = torch.cat([b.batch[:, None].int(), b.coords.int()], 1).to(dev)
    cm = CoordinateManager(coords, device=dev, batch_size=32, bounds=b.coord_bounds)
    ts = 1
    sparse_ops.DEFAULTS.cmp_mode = 128
    for ts_in, c in ((2, 64), (4, 128), (8, 256), (16, 512)):
        while ts < ts_in:
            cm.stride(ts, 2)
            ts *= 2
        n = cm.level(ts_in).n
        nbr = cm.kernel_map(ts_in, 3, 1)
        x = torch.randn(n, c, device=dev)
        w = torch.randn(27 * c, c, device=dev) * 0.05
        for _ in range(3):
            spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
        torch.cuda.synchronize()
        L.agb_debug_cmpt_stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        spconv_forward_raw(x, w, nbr, 0, None, n, 27, c, c)
        e1.record()
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 16)()
        L.agb_debug_cmpt_stamps(out, 0)
        pro, steps, ng, s0, s1, s2, s3, s4, wave = [float(v) for v in out[:9]]
        print(f"ts {ts_in:2d} {c}->{c} rows {n}: {e0.elapsed_time(e1) * 1e3:.0f} us (stamped build); steps {steps:.0f}, groups/step {ng / steps:.2f}; "
              f"per step: prologue {pro / steps:.0f} clk; per group: cb0 {s0 / ng:.0f}  cb1 {s1 / ng:.0f}  cb2 {s2 / ng:.0f}  cb3 {s3 / ng:.0f}  "
              f"tail {s4 / ng:.0f}  (sum {(s0 + s1 + s2 + s3 + s4) / ng:.0f}); wave clocks accounted "
              f"{(pro + s0 + s1 + s2 + s3 + s4) / wave:.3f}", flush=True)


if __name__ == "__main__":
    main()
