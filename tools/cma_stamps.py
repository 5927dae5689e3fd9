"""Stamp shares of the hand-scheduled k_spconv_cma (diagnostic build: cd dpcr-agb_amd/csrc && make timeline; read the SHARES)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AGB_LIBRARY", os.path.join(ROOT, "dpcr-agb_amd", "libagbhip_timeline.so"))
os.environ.setdefault("AGB_CMPT", "3")
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
        top, tf, tp, nf, np_, waves, wvm, wlg = [float(v) for v in out[:8]]
        steps = waves * 27 * (c // 64)
        print(f"ts {ts_in:2d} {c}->{c} rows {n}: {e0.elapsed_time(e1) * 1e3:.0f} us (stamped build); waves {waves:.0f}, steps {steps:.0f}, "
              f"groups/step {(nf + np_) / steps:.2f}; per step: top+end {top / steps:.0f} clk; first group {tf / max(nf, 1):.0f} clk; "
              f"plain group {tp / max(np_, 1):.0f} clk; total per step {(top + tf + tp) / steps:.0f}, pure MFMA {(nf + np_) * 2048 / steps:.0f} "
              f"({(nf + np_) * 2048 / (top + tf + tp):.3f}); per group: in vmcnt waits {wvm / (nf + np_):.0f} clk, in the list-entry wait "
              f"{wlg / (nf + np_):.0f} clk (each bracketed wait includes ~2 clock reads)", flush=True)


if __name__ == "__main__":
    main()
